"""GPU parity tests (MI355X): the HIP path, called through the C ABI, against the CPU oracle and the golden
vectors captured from the reference.  fp32 tolerances (north_star: loss within 1e-4 relative):
    forward tensors  <= 2e-5 absolute (values are O(1));   losses <= 2e-5 relative;
    gradients        <= 2e-4 of the tensor's max |g| (different fp32 summation order over the batch).
"""
import numpy as np
import pytest
import torch

import vpc_amd as vpc
from conftest import golden_params
from oracle import vae_oracle as O

pytestmark = pytest.mark.gpu
L = 10
TP = {"batch_size": 64, "patience": 100}
DEV = "cuda"


@pytest.fixture(autouse=True, params=["auto", "tile64", "tile128"])
def tile_shape(request, monkeypatch):
    """Every test of this module runs three times: with the library's own choice (for these batch sizes the fused fp32 step
    runs the 16-row N-split kernel csrc/vpc_small.hip; the API path the 64-row small-batch shape), with the 64-row small-batch
    shape forced (VPC_TILE=64: 4 waves, passes spread over blockIdx.y) and with the throughput shape forced (VPC_TILE=128:
    128-row tiles, 8 waves, passes looped) - csrc/vpc_abi_internal.h `tile_shape`, csrc/vpc_small.hip."""
    if request.param == "tile128":
        monkeypatch.setenv("VPC_TILE", "128")
    elif request.param == "tile64":
        monkeypatch.setenv("VPC_TILE", "64")
    else:
        monkeypatch.delenv("VPC_TILE", raising=False)
    return request.param


def _t(a, dev=DEV):
    return torch.from_numpy(np.array(a)).to(dev)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


def make_model(cls, d, params, **kw):
    if cls is vpc.Reg_VAE:
        m = cls(d, 500, 10, L, TP, "exp", kw.get("reg_type", "kl_reg"))
    else:
        m = cls(d, 500, 10, L, TP, "exp")
    sd = m.state_dict()
    for k, v in params.items():
        sd[k] = v.clone()
    m.load_state_dict(sd)
    return m.to(DEV)


def synth(B, d, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, d, generator=g)
    mask = torch.rand(B, d, generator=g) < 0.7
    mask_p = mask & (torch.rand(B, d, generator=g) < 0.7)
    eq = torch.randn(B, L, generator=g)
    ep = torch.randn(B, L, generator=g)
    return x, mask, mask_p, eq, ep


# ----------------------------------------------------------------------------------------------- raw kernels
@pytest.mark.parametrize("d,B", [(14, 64), (128, 64), (128, 200), (40, 33), (13, 5), (128, 1), (64, 257)])
def test_encoder_decoder_forward_vs_oracle(d, B):
    params = O.init_params(d, L, seed=d + B)
    x, mask, _, eq, _ = synth(B, d, seed=B)
    port = O.TorchPort(params, L)
    z_ref, mean_ref, lv_ref = port.encoder(x, mask, eq)
    xh_ref, _ = port.decoder(z_ref)
    m = make_model(vpc.Reg_VAE, d, params)
    lay = m._lay()
    xd, md, ed = x.to(DEV), vpc.ops.as_mask_u8(mask.to(DEV)), eq.to(DEV)
    h1 = torch.empty(B, 112, device=DEV); h2 = torch.empty(B, 64, device=DEV)
    mean = torch.empty(B, L, device=DEV); lv = torch.empty(B, L, device=DEV); z = torch.empty(B, L, device=DEV)
    vpc.ops.encoder_fwd(xd, m._enc_img(), [md], [ed], [h1], [h2], [mean], [lv], [z], d, L)
    assert torch.allclose(mean.cpu(), mean_ref, atol=2e-5)
    assert torch.allclose(lv.cpu(), lv_ref, atol=2e-5)
    assert torch.allclose(z.cpu(), z_ref, atol=3e-5)
    # hidden activations incl. the constant-1 units of the bias chain
    P = O._np(params)
    c = O.closed_form_pass(P, x.numpy().astype(np.float64), mask.numpy().astype(np.float64),
                           eq.numpy().astype(np.float64), L)
    from vpc_amd._lib import HIDDEN_POS1 as P1, HIDDEN_POS2 as P2  # unit -> position in the padded workspace
    h1n, h2n = h1.cpu().numpy(), h2.cpu().numpy()
    assert np.allclose(h1n[:, P1[:100]], c.h1, atol=2e-5)
    assert np.allclose(h1n[:, P1[100]], 1.0) and np.all(np.delete(h1n, P1, axis=1) == 0)
    assert np.allclose(h2n[:, P2[:50]], c.h2, atol=2e-5) and np.allclose(h2n[:, P2[50]], 1.0)
    xh = torch.empty(B, d, device=DEV)
    vpc.ops.decoder_fwd(z, m._dec_img(), xh, d, L)
    assert torch.allclose(xh.cpu(), xh_ref, atol=2e-5)
    # sample=False branch: z = mean
    z2 = torch.empty(B, L, device=DEV)
    vpc.ops.encoder_fwd(xd, m._enc_img(), [md], [None], [h1], [h2], [mean], [lv], [z2], d, L)
    assert torch.equal(z2, mean)


# ----------------------------------------------------------------------------------------------- API path
@pytest.mark.parametrize("d", [14, 128])
def test_api_forward_matches_golden(d):
    g = vpc_golden(f"reg_d{d}.npz")
    m = make_model(vpc.Reg_VAE, d, golden_params(g))
    x, mk, mp = _t(g["x"]), _t(g["mask"]), _t(g["mask_p"])
    # inject the recorded eps through the autograd Function (forward() itself draws on the device)
    zq, mq, lq = vpc.ops.EncoderFn.apply(m, x, vpc.ops.as_mask_u8(mk), _t(g["eps_q"]), *m.trainable()[:6])
    xq, xlv = m.decoder(zq)
    zp, mpn, lp = vpc.ops.EncoderFn.apply(m, x, vpc.ops.as_mask_u8(mp), _t(g["eps_p"]), *m.trainable()[:6])
    xp, _ = m.decoder(zp)
    for got, name in ((mq, "mean_q"), (lq, "logvar_q"), (xq, "x_mean_q"), (mpn, "mean_p"), (lp, "logvar_p"),
                      (xp, "x_mean_p")):
        assert np.allclose(got.detach().cpu().numpy(), g[name], atol=2e-5), name
    assert xlv.shape == (1,) and abs(xlv.item() - g["x_logvar"][0]) < 1e-6
    # forward() return order / shapes (VAE.py:507) and device-side eps statistics
    out = m.forward(x, mk, mp, "train")
    assert len(out) == 8 and out[0].shape == (64, L) and out[2].shape == (64, d) and out[3].shape == (1,)
    assert np.allclose(out[4].detach().cpu().numpy(), g["mean_q"], atol=2e-5)  # mean_q does not depend on eps


_gold = {}


def vpc_golden(name):
    from conftest import load_golden
    if name not in _gold:
        _gold[name] = load_golden(name)
    return _gold[name]


def _api_outputs(m, g):
    x, mk, mp = _t(g["x"]), _t(g["mask"]), _t(g["mask_p"])
    zq, mq, lq = vpc.ops.EncoderFn.apply(m, x, vpc.ops.as_mask_u8(mk), _t(g["eps_q"]), *m.trainable()[:6])
    xq, xlv = m.decoder(zq)
    zp, mpn, lp = vpc.ops.EncoderFn.apply(m, x, vpc.ops.as_mask_u8(mp), _t(g["eps_p"]), *m.trainable()[:6])
    xp, _ = m.decoder(zp)
    return x, mk, mp, (mpn, lp, xp, xlv, mq, lq, xq, xlv)


@pytest.mark.parametrize("d", [14, 128])
def test_api_loss_grid_matches_golden(d):
    g = vpc_golden(f"reg_d{d}.npz")
    m = make_model(vpc.Reg_VAE, d, golden_params(g))
    with torch.no_grad():
        x, mk, mp, o = _api_outputs(m, g)
        for cfg, want in zip(g["klreg_grid_cfg"], g["klreg_grid_loss"]):
            alpha, beta, ann, epoch = cfg
            pl, tl = m.loss(x, o[2], o[3], o[0], o[1], o[6], o[7], o[4], o[5], mk, mp, int(epoch),
                            beta_annealing=bool(ann), beta=float(beta), alpha=float(alpha), stage="train")
            assert abs(tl.item() - want) <= 2e-5 * abs(want), cfg
        r = m.loss(x, o[2], o[3], o[0], o[1], o[6], o[7], o[4], o[5], mk, mp, 1, llh_eval=True, beta=1.0, alpha=1.0,
                   stage="evaluate")
        assert rel([r[1].item(), r[2].item(), r[3].item()], g["eval_llh"]) < 2e-5
        r = m.loss(x, o[2], o[3], o[0], o[1], o[6], o[7], o[4], o[5], mk, mp, 1, MI=True, beta=1.0, alpha=1.0,
                   stage="evaluate")
        assert rel([r[1].item(), r[2].item(), r[3].item()], g["eval_MI"]) < 1e-4
        r = m.loss(x, o[2], o[3], o[0], o[1], o[6], o[7], o[4], o[5], mk, mp, 7, llh_eval=True, beta=0.7, alpha=0.5,
                   stage="train", beta_annealing=True)
        assert rel([r[1].item(), r[2].item(), float(r[3])], g["train_llh"]) < 2e-5


@pytest.mark.parametrize("d", [14, 128])
@pytest.mark.parametrize("tag,kw", [
    ("a1", dict(alpha=1.0, beta=1.0, beta_annealing=False, epoch=1)),
    ("a05", dict(alpha=0.5, beta=0.7, beta_annealing=True, epoch=1400)),
])
def test_api_backward_matches_golden(d, tag, kw):
    g = vpc_golden(f"reg_d{d}.npz")
    m = make_model(vpc.Reg_VAE, d, golden_params(g))
    x, mk, mp, o = _api_outputs(m, g)
    kw = dict(kw)
    epoch = kw.pop("epoch")
    _, tl = m.loss(x, o[2], o[3], o[0], o[1], o[6], o[7], o[4], o[5], mk, mp, epoch, stage="train", **kw)
    tl.backward()
    want = float(g["loss_" + tag])
    assert abs(tl.item() - want) <= 2e-5 * abs(want)
    for k, p in zip(O.PARAM_KEYS, m.trainable()):
        assert rel(p.grad.cpu().numpy(), g[f"{tag}.grad.{k}"]) < 2e-4, k


@pytest.mark.parametrize("d", [14, 128])
def test_api_ml_reg_vs_oracle(d):
    """ml_reg draws its extra eps on the device, so compare with the oracle through the injected-eps LossFn."""
    g = vpc_golden(f"reg_d{d}.npz")
    m = make_model(vpc.Reg_VAE, d, golden_params(g), reg_type="ml_reg")
    x, mk, mp, o = _api_outputs(m, g)
    cfg = dict(d=d, L=L, x_logvar=m._x_logvar_value, cr=0.0, bp=0.0, bq=1.0, wml=(1400 / 2800) * 0.8,
               maskA=[vpc.ops.as_mask_u8(mk), vpc.ops.as_mask_u8(mp)], maskB=[None, None], cA=[1.0, 0.0], cE=[0.0, 0.0])
    tl, _ = vpc.ops.LossFn.apply(cfg, x, o[6], o[2], o[4], o[5], o[0], o[1], _t(g["eps_ml"]))
    tl.backward()
    want = float(g["loss_ml"])
    assert abs(tl.item() - want) <= 2e-5 * abs(want)
    for k, p in zip(O.PARAM_KEYS, m.trainable()):
        assert rel(p.grad.cpu().numpy(), g[f"ml.grad.{k}"]) < 2e-4, k
    # the public loss() runs too (device-side eps) and is finite
    x, mk, mp, o = _api_outputs(m, g)
    _, tl2 = m.loss(x, o[2], o[3], o[0], o[1], o[6], o[7], o[4], o[5], mk, mp, 1400, beta=1.0, alpha=0.8)
    assert torch.isfinite(tl2)


@pytest.mark.parametrize("d", [14, 128])
def test_vanilla_api_matches_golden(d):
    g = vpc_golden(f"vanilla_d{d}.npz")
    m = make_model(vpc.vanilla_VAE, d, golden_params(g))
    x, mk = _t(g["x"]), _t(g["mask"])
    mf = mk * torch.ones(x.shape, device=DEV)  # train.py:58,97 float mask
    zq, mq, lq = vpc.ops.EncoderFn.apply(m, x, vpc.ops.as_mask_u8(mf), _t(g["eps_q"]), *m.trainable()[:6])
    xq, xlv = m.decoder(zq)
    assert np.allclose(xq.detach().cpu().numpy(), g["x_mean_q"], atol=2e-5)
    with torch.no_grad():
        for cfg, want in zip(g["grid_cfg"], g["grid_loss"]):
            beta, ann, epoch = cfg
            _, tl = m.loss(x, xq, xlv, mq, lq, int(epoch), mf, beta_annealing=bool(ann), beta=float(beta))
            assert abs(tl.item() - want) <= 2e-5 * abs(want)
        r = m.loss(x, xq, xlv, mq, lq, 1, mf, llh_eval=True, stage="evaluate")
        assert rel([r[1].item(), r[2].item(), r[3].item()], g["eval_llh"]) < 2e-5
    _, tl = m.loss(x, xq, xlv, mq, lq, 1, mf, stage="train")
    tl.backward()
    for k, p in zip(O.PARAM_KEYS, m.trainable()):
        assert rel(p.grad.cpu().numpy(), g[f"b1.grad.{k}"]) < 2e-4, k
    out = m.forward(x, mf)
    assert len(out) == 4 and out[2].shape == (64, d)


# ----------------------------------------------------------------------------------------------- fused step
@pytest.mark.parametrize("d", [14, 128])
@pytest.mark.parametrize("tag,kw", [
    ("a1", dict(alpha=1.0, beta=1.0, beta_annealing=False, epoch=1)),
    ("a05", dict(alpha=0.5, beta=0.7, beta_annealing=True, epoch=1400)),
    ("ml", dict(alpha=0.8, beta=1.0, beta_annealing=False, epoch=1400)),
])
def test_fused_step_grads_match_golden(d, tag, kw):
    g = vpc_golden(f"reg_d{d}.npz")
    m = make_model(vpc.Reg_VAE, d, golden_params(g), reg_type="ml_reg" if tag == "ml" else "kl_reg")
    tr = vpc.FusedTrainer(m)
    tr.step(_t(g["x"]), _t(g["mask"]), _t(g["mask_p"]), _t(g["eps_q"]), _t(g["eps_p"]),
            _t(g["eps_ml"]) if tag == "ml" else None, update=False, **kw)
    want = float(g["loss_" + tag])
    assert abs(tr.loss_value() - want) <= 2e-5 * abs(want)
    flat = tr.grad.cpu().numpy()
    off = 0
    for k, p in zip(O.PARAM_KEYS, m.trainable()):
        got = flat[off:off + p.numel()].reshape(p.shape)
        assert rel(got, g[f"{tag}.grad.{k}"]) < 2e-4, k
        off += p.numel()


@pytest.mark.parametrize("d", [14, 128])
def test_fused_vanilla_step_grads_match_golden(d):
    """vanilla_VAE on the FUSED step against the reference's own vectors (loss_b1 + all 12 gradients).  d = 128 is
    the shape that runs dec8_kernel / enc_*_kernel<8> with npass == 1 (one pass, no second mask, no KL coupling);
    d = 14 runs the 4-wave decoder.  Reference: src/models/VAE.py:1171-1208 + autograd."""
    g = vpc_golden(f"vanilla_d{d}.npz")
    m = make_model(vpc.vanilla_VAE, d, golden_params(g))
    tr = vpc.FusedTrainer(m)
    tr.step(_t(g["x"]), _t(g["mask"]), eps_q=_t(g["eps_q"]), beta=1.0, epoch=1, update=False)
    want = float(g["loss_b1"])
    assert abs(tr.loss_value() - want) <= 2e-5 * abs(want)
    flat = tr.grad.cpu().numpy()
    off = 0
    for k, p in zip(O.PARAM_KEYS, m.trainable()):
        got = flat[off:off + p.numel()].reshape(p.shape)
        assert rel(got, g[f"b1.grad.{k}"]) < 2e-4, k
        off += p.numel()
    # and at a batch that fills several workgroup tiles, against the oracle's port (same shape class)
    B = 700
    params = O.init_params(d, L, seed=21)
    x, mask, _, eq, _ = synth(B, d, seed=5)
    loss_ref, grads_ref, _ = O.torch_vanilla_step(params, L, x, mask, eq)
    m2 = make_model(vpc.vanilla_VAE, d, params)
    tr2 = vpc.FusedTrainer(m2)
    tr2.step(x.to(DEV), mask.to(DEV), eps_q=eq.to(DEV), update=False)
    assert abs(tr2.loss_value() - loss_ref.item()) <= 2e-5 * abs(loss_ref.item())
    flat, off = tr2.grad.cpu().numpy(), 0
    for k, p in zip(O.PARAM_KEYS, m2.trainable()):
        assert rel(flat[off:off + p.numel()].reshape(p.shape), grads_ref[k].numpy()) < 2e-4, k
        off += p.numel()


@pytest.mark.parametrize("kind", ["reg", "vanilla"])
def test_fused_adam_trajectory_matches_golden(kind):
    g = vpc_golden(f"traj_{kind}_d14.npz")
    cls = vpc.Reg_VAE if kind == "reg" else vpc.vanilla_VAE
    m = make_model(cls, 14, golden_params(g, "param0."))
    tr = vpc.FusedTrainer(m, lr=1e-3)
    x, mk = _t(g["x"]), _t(g["mask"])
    for i in range(len(g["loss"])):
        if kind == "reg":
            tr.step(x, mk, _t(g["mask_p"][i]), _t(g["eps_q"][i]), _t(g["eps_p"][i]), epoch=i + 1)
        else:
            tr.step(x, mk, eps_q=_t(g["eps_q"][i]), epoch=i + 1)
        assert abs(tr.loss_value() - g["loss"][i]) <= 3e-5 * abs(g["loss"][i]), i
    assert abs(tr.epoch_total() - g["loss"].sum()) <= 3e-5 * g["loss"].sum()
    pT = golden_params(g, "paramT.")
    for k, p in zip(O.PARAM_KEYS, m.trainable()):
        assert rel(p.detach().cpu().numpy(), pT[k].numpy()) < 1e-4, k
    # the API path sees the updated weights (Adam re-packs the image)
    with torch.no_grad():
        _, mean, _ = m.encoder(x, mk, sample=False)
    port = O.TorchPort(pT, L)
    _, mean_ref, _ = port.encoder(_t(g["x"], "cpu"), _t(g["mask"], "cpu"), sample=False)
    assert torch.allclose(mean.cpu(), mean_ref, atol=1e-4)


def test_api_path_training_with_torch_adam_matches_golden():
    """The reference's own sequence (model.forward/loss, backward, optim.Adam) on the API path."""
    g = vpc_golden("traj_vanilla_d14.npz")
    m = make_model(vpc.vanilla_VAE, 14, golden_params(g, "param0."))
    m.flatten_parameters()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    x, mk = _t(g["x"]), _t(g["mask"])
    mf = mk * torch.ones(x.shape, device=DEV)
    for i in range(len(g["loss"])):
        zq, mq, lq = vpc.ops.EncoderFn.apply(m, x, vpc.ops.as_mask_u8(mf), _t(g["eps_q"][i]), *m.trainable()[:6])
        xq, xlv = m.decoder(zq)
        _, tl = m.loss(x, xq, xlv, mq, lq, i + 1, mf, beta_annealing=False, beta=1.0, stage="train")
        opt.zero_grad()
        tl.backward()
        opt.step()
        assert abs(tl.item() - g["loss"][i]) <= 3e-5 * abs(g["loss"][i]), i
    pT = golden_params(g, "paramT.")
    for k, p in zip(O.PARAM_KEYS, m.trainable()):
        assert rel(p.detach().cpu().numpy(), pT[k].numpy()) < 1e-4, k


# ----------------------------------------------------------------------------------------------- sizes / edges
@pytest.mark.parametrize("d,B", [(13, 5), (40, 129), (128, 1), (100, 300), (128, 1000), (125, 70), (72, 130), (66, 257)])
def test_fused_step_ragged_shapes_vs_oracle(d, B):
    params = O.init_params(d, L, seed=7)
    x, mask, mask_p, eq, ep = synth(B, d, seed=B + d)
    loss_ref, grads_ref, _ = O.torch_reg_step(params, L, x, mask, mask_p, eq, ep, alpha=0.8, beta=0.9)
    m = make_model(vpc.Reg_VAE, d, params)
    tr = vpc.FusedTrainer(m)
    tr.step(x.to(DEV), mask.to(DEV), mask_p.to(DEV), eq.to(DEV), ep.to(DEV), alpha=0.8, beta=0.9, update=False)
    assert abs(tr.loss_value() - loss_ref.item()) <= 2e-5 * abs(loss_ref.item())
    off = 0
    flat = tr.grad.cpu().numpy()
    for k, p in zip(O.PARAM_KEYS, m.trainable()):
        assert rel(flat[off:off + p.numel()].reshape(p.shape), grads_ref[k].numpy()) < 2e-4, k
        off += p.numel()


@pytest.mark.parametrize("d,B,kind", [(128, 8192, "reg"), (128, 16384, "reg"), (128, 8192 + 37, "vanilla"),
                                      (40, 4000, "reg"), (128, 64, "reg")])
def test_workgroup_shapes_agree(d, B, kind, monkeypatch):
    """Small-batch shape (VPC_TILE=64) vs throughput shape (VPC_TILE=128) on the same inputs: losses and gradients
    agree up to fp32 summation order, and each is bit-reproducible run to run.  B = 8 192 is one GPU's shard of the
    headline batch under 8-way strong scaling; 16 384 is the largest batch the library gives the small shape on its own."""
    params = O.init_params(d, L, seed=3)
    x, mask, mask_p, eq, ep = synth(B, d, seed=B)
    res = {}
    small_ok = B <= 8192  # the N-split kernel takes one 16-row tile per workgroup, up to 2 x CUs of them
    for tile in ("64", "128", "64") + (("16", "16") if small_ok else ()):
        monkeypatch.setenv("VPC_TILE", tile)
        m = make_model(vpc.Reg_VAE if kind == "reg" else vpc.vanilla_VAE, d, params)
        tr = vpc.FusedTrainer(m)
        if kind == "reg":
            tr.step(x.to(DEV), mask.to(DEV), mask_p.to(DEV), eq.to(DEV), ep.to(DEV), alpha=0.8, beta=0.9, update=False)
        else:
            tr.step(x.to(DEV), mask.to(DEV), eps_q=eq.to(DEV), update=False)
        cur = (tr.loss_value(), tr.grad.clone(), tr.last_blocks)
        if tile in res:
            assert res[tile][0] == cur[0] and torch.equal(res[tile][1], cur[1])  # deterministic
        res[tile] = cur
    npass = 2 if kind == "reg" else 1
    assert res["64"][2] == (min((B + 63) // 64, 256) * npass,) * 2
    assert abs(res["64"][0] - res["128"][0]) <= 2e-6 * abs(res["128"][0])
    g64, g128 = res["64"][1].cpu().numpy(), res["128"][1].cpu().numpy()
    assert rel(g64, g128) < 2e-5
    # the 16-row N-split kernel (csrc/vpc_small.hip): one launch for the whole step, the same partial-block layout
    if small_ok:
        assert abs(res["16"][0] - res["128"][0]) <= 2e-6 * abs(res["128"][0])
        assert rel(res["16"][1].cpu().numpy(), g128) < 2e-5


@pytest.mark.parametrize("d,B", [(128, 300), (100, 130), (125, 70), (128, 40000), (128, 65536)])
def test_decoder_kernel_variants_agree(d, B, monkeypatch):
    """d in (64, 128] has two fused decoder kernels (8 waves x 1 row tile - the default - and 4 waves x 2 row tiles,
    VPC_DEC8=0): same arguments, same partial-block layout, results equal up to fp32 summation order."""
    params = O.init_params(d, L, seed=11)
    x, mask, mask_p, eq, ep = synth(B, d, seed=B * 3 + d)
    res = []
    for v in ("1", "0"):
        monkeypatch.setenv("VPC_DEC8", v)
        tr = vpc.FusedTrainer(make_model(vpc.Reg_VAE, d, params))
        tr.step(x.to(DEV), mask.to(DEV), mask_p.to(DEV), eq.to(DEV), ep.to(DEV), alpha=0.6, beta=0.9, update=False)
        res.append((tr.loss_value(), tr.grad.cpu().numpy().copy()))
    assert abs(res[0][0] - res[1][0]) <= 2e-6 * abs(res[1][0])
    assert rel(res[0][1], res[1][1]) < 2e-5


def test_full_size_step_vs_oracle_and_api_path():
    """BASELINE.json headline shape: B = 65 536, d = 128.  Loss within 1e-4 relative (target) of the CPU path."""
    B, d = 65536, 128
    params = O.init_params(d, L, seed=0)
    x, mask, mask_p, eq, ep = synth(B, d, seed=1)
    loss_ref, grads_ref, _ = O.torch_reg_step(params, L, x, mask, mask_p, eq, ep, alpha=1.0, beta=1.0)
    m = make_model(vpc.Reg_VAE, d, params)
    tr = vpc.FusedTrainer(m)
    xd, md, mpd, eqd, epd = x.to(DEV), mask.to(DEV), mask_p.to(DEV), eq.to(DEV), ep.to(DEV)
    tr.step(xd, md, mpd, eqd, epd, update=False)
    assert abs(tr.loss_value() - loss_ref.item()) <= 1e-5 * abs(loss_ref.item())
    fused = tr.grad.clone()
    off = 0
    for k, p in zip(O.PARAM_KEYS, m.trainable()):
        assert rel(fused[off:off + p.numel()].cpu().numpy().reshape(p.shape), grads_ref[k].numpy()) < 5e-4, k
        off += p.numel()
    # determinism: the same step twice gives bit-identical gradients (fixed summation order, no atomics)
    tr.step(xd, md, mpd, eqd, epd, update=False)
    assert torch.equal(tr.grad, fused)
    # API path on the same inputs
    for p in m.trainable():
        p.grad = None
    zq, mq, lq = vpc.ops.EncoderFn.apply(m, xd, vpc.ops.as_mask_u8(md), eqd, *m.trainable()[:6])
    xq, xlv = m.decoder(zq)
    zp, mpn, lp = vpc.ops.EncoderFn.apply(m, xd, vpc.ops.as_mask_u8(mpd), epd, *m.trainable()[:6])
    xp, _ = m.decoder(zp)
    _, tl = m.loss(xd, xp, xlv, mpn, lp, xq, xlv, mq, lq, md, mpd, 1, beta=1.0, alpha=1.0)
    tl.backward()
    assert abs(tl.item() - loss_ref.item()) <= 1e-5 * abs(loss_ref.item())
    api = torch.cat([p.grad.reshape(-1) for p in m.trainable()])
    assert rel(api.cpu().numpy(), fused.cpu().numpy()) < 2e-4


# ----------------------------------------------------------------------------------------------- random draws
def test_device_mask_and_normal_draws():
    B, d = 4096, 128
    mask = (torch.rand(B, d, device=DEV) < 0.7)
    keep = vpc.create_missing_uci((B, d), 30, device=DEV)
    assert keep.dtype == torch.bool and abs(keep.float().mean().item() - 0.7) < 5e-3
    mp = torch.empty(B, d, dtype=torch.uint8, device=DEV)
    vpc.ops.draw_mask(vpc.ops.as_mask_u8(mask), mp, 0.7, 123, 0)
    mpb = mp.bool()
    assert not (mpb & ~mask).any()                                    # mask_p is a subset of mask (train.py:55)
    assert abs((mpb.sum() / mask.sum()).item() - 0.7) < 5e-3
    mp2 = torch.empty_like(mp)
    vpc.ops.draw_mask(vpc.ops.as_mask_u8(mask), mp2, 0.7, 123, 0)
    assert torch.equal(mp, mp2)                                       # counter-based: reproducible
    vpc.ops.draw_mask(vpc.ops.as_mask_u8(mask), mp2, 0.7, 123, 1 << 20)
    assert not torch.equal(mp, mp2)
    e = torch.empty(1 << 20, device=DEV)
    vpc.ops.fill_normal(e, 5, 0)
    assert abs(e.mean().item()) < 5e-3 and abs(e.std().item() - 1) < 5e-3
    assert abs((e ** 4).mean().item() - 3.0) < 0.05


def test_device_mask_draw_ragged_and_unaligned():
    """The keep-mask kernel serves 8 bytes per Philox call: lengths that are not multiples of 8 and byte-offset
    (unaligned) views take its scalar tail path and must give the same bits as the vector path for the same counters."""
    n = 8 * 5000
    base_in = (torch.rand(n + 16, device=DEV) < 0.8).to(torch.uint8)
    ref = torch.empty(n, dtype=torch.uint8, device=DEV)
    vpc.ops.draw_mask(base_in[:n], ref, 0.6, 77, 5)                      # aligned, vector path
    assert abs(ref.float().sum().item() / base_in[:n].float().sum().item() - 0.6) < 2e-2
    assert not (ref.bool() & ~base_in[:n].bool()).any()
    short = torch.empty(n - 3, dtype=torch.uint8, device=DEV)
    vpc.ops.draw_mask(base_in[:n - 3], short, 0.6, 77, 5)                # ragged tail: last group is scalar
    assert torch.equal(short, ref[:n - 3])
    buf_in = torch.zeros(n + 16, dtype=torch.uint8, device=DEV)
    buf_in[1:n + 1] = base_in[:n]
    buf_out = torch.zeros(n + 16, dtype=torch.uint8, device=DEV)
    vpc.ops.draw_mask(buf_in[1:n + 1], buf_out[1:n + 1], 0.6, 77, 5)     # both pointers off by one byte: scalar path
    assert torch.equal(buf_out[1:n + 1], ref) and buf_out[0] == 0 and buf_out[n + 1] == 0
    ones = torch.empty(n, dtype=torch.uint8, device=DEV)
    vpc.ops.draw_mask(None, ones, 1.0, 1, 0)                             # keep_prob 1 keeps everything, NULL = all ones
    assert bool(ones.all())
    vpc.ops.draw_mask(None, ones, 0.0, 1, 0)
    assert not bool(ones.any())


def test_gradient_reduction_forms_agree():
    """vpc_reduce_step reads the partial blocks in layout order (16 B / lane) when the caller passes the inverse maps
    of vpc_build_inverse_maps and the blocks are 16-byte aligned, and gathers through grad_idx otherwise (no map, or
    unaligned blocks): same gradients and loss terms up to summation order."""
    d, B = 128, 1000
    params = O.init_params(d, L, seed=5)
    x, mask, mask_p, eq, ep = synth(B, d, seed=99)
    tr = vpc.FusedTrainer(make_model(vpc.Reg_VAE, d, params))
    tr.step(x.to(DEV), mask.to(DEV), mask_p.to(DEV), eq.to(DEV), ep.to(DEV), alpha=0.7, beta=0.9, update=False)
    lay, (nbE, nb) = tr.lay, tr.last_blocks
    assert nbE == nb
    co = tr.coefficients(1, 0.7, 0.9, False)
    args = (lay.n_enc, tr.loss_part, nb, co["cA"][0], co["cE"][0], co["cA"][1], co["bq"], co["bp"], co["cr"], co["wml"],
            B, B, d)
    res = []
    for shift in (0, 1, 2):  # 1: views that start 4 bytes into a buffer -> not 16-byte aligned -> gather form; 2: no map
        sh = shift & 1
        pe = torch.zeros(tr.partE.numel() + 4, device=DEV)[sh:sh + tr.partE.numel()]
        pd = torch.zeros(tr.partD.numel() + 4, device=DEV)[sh:sh + tr.partD.numel()]
        pe.copy_(tr.partE); pd.copy_(tr.partD)
        assert (pe.data_ptr() % 16 == 0) == (sh == 0)
        g, o9 = torch.zeros_like(tr.grad), torch.zeros(9, device=DEV)
        vpc.ops.reduce_step(pe, nb, lay.enc_part, pd, nb, lay.dec_part, tr.gidx, g, *args, o9,
                            inv_maps=None if shift == 2 else tr.inv)
        res.append((g.cpu().numpy(), o9.cpu().numpy()))
    assert np.array_equal(res[0][0], tr.grad.cpu().numpy())       # the trainer used the layout-order form
    assert np.array_equal(res[1][0], res[2][0])                    # both gather runs
    assert rel(res[0][0], res[1][0]) < 1e-6 and np.allclose(res[0][1], res[1][1], rtol=1e-6)


def test_train_harness_fused_and_api(tmp_path, monkeypatch):
    """train() restated (train.py:13-133): loss decreases, checkpoint lands where model_loader('test') reads it."""
    monkeypatch.chdir(tmp_path)
    g = torch.Generator().manual_seed(0)
    N, d = 512, 14
    x = torch.rand(N, d, generator=g)
    mask = torch.rand(N, d, generator=g) < 0.7
    loader = [(x[i:i + 128], mask[i:i + 128]) for i in range(0, N, 128)]
    for fused in (True, False):
        torch.manual_seed(1)
        m = vpc.train((loader, None), 30, d, 500, 10, 1, L, "synth", TP, "exp", "reg_vae1", 20, 10, max_epochs=3,
                      device=torch.device(DEV), alpha=1.0, p_missingness=30, reg_type="kl_reg", fused=fused, verbose=False)
        m2 = vpc.model_loader("test", d, 500, 10, L, 30, "synth", TP, 3, 20, 10, "exp", "kl_reg", "reg_vae1", alpha=1.0,
                              p_missingness=30)
        for a, b in zip(m.trainable(), m2.trainable()):
            assert torch.equal(a.detach().cpu(), b.detach().cpu())


@pytest.mark.parametrize("kind", ["reg", "vanilla"])
def test_eval_vae_matches_oracle(kind, tmp_path, monkeypatch):
    """eval_vae restated (evaluate.py:136-297): RMSE on the unobserved entries, ELBO, NLL observed / imputed, M
    Monte-Carlo passes.  The device eps stream is replayed (torch.manual_seed) to feed the CPU oracle."""
    monkeypatch.chdir(tmp_path)
    d, N, Bs, M = 14, 96, 32, 3
    params = O.init_params(d, L, seed=21)
    g = torch.Generator().manual_seed(3)
    x = torch.rand(N, d, generator=g)
    mask = torch.rand(N, d, generator=g) < 0.7
    loader = [(x[i:i + Bs], mask[i:i + Bs]) for i in range(0, N, Bs)]
    cls = vpc.Reg_VAE if kind == "reg" else vpc.vanilla_VAE
    m = make_model(cls, d, params)
    vt = "reg_vae1" if kind == "reg" else "vanilla_vae1"
    torch.manual_seed(77)
    out = vpc.eval_vae([(loader, "test")], 30, d, 500, 10, M, L, "synth", TP, "exp", vt, 100, 5, 1, device=torch.device(DEV),
                       alpha=1.0, stage="evaluate", p_missingness=30, reg_type="kl_reg", beta=1.0, model=m)["test"]
    # replay
    torch.manual_seed(77)
    port = O.TorchPort(params, L)
    rm, el, nl, ni = [], [], [], []
    for _ in range(M):
        r_, e_, n_, i_ = [], [], [], []
        for xb, mb in loader:
            eq = torch.randn(xb.shape[0], L, device=DEV).cpu()
            if kind == "reg":
                torch.randn(xb.shape[0], L, device=DEV)  # eps_p, drawn by forward() but unused at stage=evaluate
            zq, mq, lq = port.encoder(xb, mb, eq)
            xq, xlv = port.decoder(zq)
            if kind == "reg":
                r = port.reg_loss(xb, xq, xlv, mq, lq, xq, xlv, mq, lq, mb, mb, 100, llh_eval=True, beta=1.0, alpha=1.0,
                                  stage="evaluate")
            else:
                r = port.vanilla_loss(xb, xq, xlv, mq, lq, 100, mb * 1.0, llh_eval=True, beta=1.0, stage="evaluate")
            inv = ~mb
            r_.append(torch.sqrt(torch.sum(torch.square(xq * inv - xb * inv)) / torch.sum(inv)))
            e_.append(r[1]); n_.append(r[2]); i_.append(r[3])
        rm.append(torch.stack(r_).mean()); el.append(torch.stack(e_).mean())
        nl.append(torch.stack(n_).mean()); ni.append(torch.stack(i_).mean())
    want = dict(rmse=torch.stack(rm).mean(), elbo=torch.stack(el).mean(), negll=torch.stack(nl).mean(),
                negll_imp=torch.stack(ni).mean())
    for k in want:
        assert abs(out[k].item() - want[k].item()) <= 3e-5 * abs(want[k].item()), (k, out[k].item(), want[k].item())
    import os
    for pth in vpc.result_paths("exp", "synth", vt, "test", 30, 1.0, 30, "kl_reg").values():
        assert os.path.exists(pth)


@pytest.mark.parametrize("d,B", [(128, 300), (14, 64)])
def test_mask_dtypes_give_one_result(d, B):
    """The kernels convert mask BYTES with v_cvt_f32_ubyte, so the C ABI takes 0 / 1 bytes (include/vpc.h); the host side
    accepts what the reference passes around - bool masks, float masks, and uint8 masks with any non-zero value for
    "observed" (normalised by ops.as_mask_u8).  All of them must give the bool-mask result bit for bit."""
    params = O.init_params(d, L, seed=5)
    x, mask, mask_p, eq, ep = (t.to(DEV) for t in synth(B, d, seed=11))
    forms = {"bool": (mask, mask_p), "float": (mask.float(), mask_p.float()),
             "u8_255": (mask.to(torch.uint8) * 255, mask_p.to(torch.uint8) * 7),
             "u8_01": (mask.to(torch.uint8), mask_p.to(torch.uint8))}
    res = {}
    for name, (mk, mp) in forms.items():
        tr = vpc.FusedTrainer(make_model(vpc.Reg_VAE, d, params))
        tr.step(x, mk, mp, eq, ep, alpha=0.8, beta=0.9, update=False)
        res[name] = (tr.loss_value(), tr.grad.clone())
    for name in ("float", "u8_255", "u8_01"):
        assert res[name][0] == res["bool"][0], name
        assert torch.equal(res[name][1], res["bool"][1]), name
    # the API path (encoder with a uint8 mask of 255s)
    m = make_model(vpc.Reg_VAE, d, params)
    za, _, _ = m.encoder(x, mask, sample=False)
    zb, _, _ = m.encoder(x, mask.to(torch.uint8) * 255, sample=False)
    assert torch.equal(za, zb)


@pytest.mark.parametrize("cls", ["Reg_VAE", "vanilla_VAE"])
def test_graph_replay_matches_eager_steps(cls):
    """FusedTrainer.step_graph (captured HIP graph, device-side step / RNG counters) == the same steps run eagerly:
    identical Philox streams, identical Adam bias corrections.  vanilla_VAE draws its eps through vpc_fill_normal,
    which reads the counter offset from the device `state` too: a replay must not repeat the captured noise."""
    d, B = 14, 64
    params = O.init_params(d, L, seed=9)
    x, mask, _, _, _ = synth(B, d, seed=4)
    xd, md = x.to(DEV), mask.to(DEV)
    C = getattr(vpc, cls)
    m1, m2 = make_model(C, d, params), make_model(C, d, params)
    t1, t2 = vpc.FusedTrainer(m1, seed=3), vpc.FusedTrainer(m2, seed=3)
    seen = []
    for i in range(6):
        t1.step(xd, md, alpha=0.9, beta=0.8)
        t2.step_graph(xd, md, alpha=0.9, beta=0.8)
        assert abs(t1.loss_value() - t2.loss_value()) <= 1e-6 * abs(t1.loss_value()), i
        assert torch.equal(t1.eps_buf[0], t2.eps_buf[0]), i
        seen.append(t2.eps_buf[0].clone())
    assert all(not torch.equal(a, b) for a, b in zip(seen, seen[1:]))  # fresh noise on every replay
    assert torch.equal(m1._flat, m2._flat)
    assert abs(t1.epoch_total() - t2.epoch_total()) < 1e-3
    # new inputs are copied into the captured buffers
    x2 = torch.rand(B, d, device=DEV)
    t1.step(x2, md, alpha=0.9, beta=0.8); t2.step_graph(x2, md, alpha=0.9, beta=0.8)
    assert torch.equal(m1._flat, m2._flat)


@pytest.mark.parametrize("prec,tile,B", [("f32", None, 64), ("f32", "64", 200), ("bf16", "128", 300), ("bf16", None, 64)])
def test_graph_replay_matches_eager_steps_d128(prec, tile, B, monkeypatch):
    """The same at obs_dim 128 in every form of the step: the N-split kernel (B = 64, fp32), the three small-shape kernels, the
    whole-step bf16 kernel (its compact image is re-packed inside the graph) and the small-shape bf16 kernels - bitwise equal
    weights after six steps, fresh noise on every replay."""
    if tile:
        monkeypatch.setenv("VPC_TILE", tile)
    d = 128
    params = O.init_params(d, L, seed=9)
    x, mask, _, _, _ = synth(B, d, seed=4)
    xd, md = x.to(DEV), mask.to(DEV)
    m1, m2 = make_model(vpc.Reg_VAE, d, params), make_model(vpc.Reg_VAE, d, params)
    t1, t2 = vpc.FusedTrainer(m1, seed=3, precision=prec), vpc.FusedTrainer(m2, seed=3, precision=prec)
    seen = []
    for i in range(6):
        t1.step(xd, md, alpha=0.9, beta=0.8)
        t2.step_graph(xd, md, alpha=0.9, beta=0.8)
        assert t1.loss_value() == t2.loss_value(), i
        seen.append(t2.eps_buf[0].clone())
    assert t1.dominant_launch() == t2.dominant_launch()
    assert all(not torch.equal(a, b) for a, b in zip(seen, seen[1:]))
    assert torch.equal(m1._flat, m2._flat)
    assert int(t2.state[0]) == 6  # steps done: one eager warm-up + five replays
