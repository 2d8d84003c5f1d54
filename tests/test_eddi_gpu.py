"""GPU parity of the PNP / EDDI family through the C ABI, against vectors captured from
the reference itself (tests/golden/eddi_*.npz) and the float64 folded closed form of the oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ed():
    import vpc_amd
    from vpc_amd import eddi
    return eddi


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _close(a, b, tol, what):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * scale, f"{what}: max abs err {err:.3e} (scale {scale:.3e})"


def _load(g, cls, *extra, prefix="param."):
    d = g["x"].shape[1]
    model = cls(d, 500, int(g["K"]), int(g["L"]), {"batch_size": 64, "patience": 1}, "exp", *extra)
    model.load_state_dict({k[len(prefix):]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith(prefix)})
    return model.cuda()


def _grad_check(model, g, tag, tol=2e-4):
    for k, p in model.named_parameters():
        ref = g.get(f"grad.{tag}.{k}")
        if ref is None:
            continue
        assert p.grad is not None, k
        _close(p.grad, torch.from_numpy(ref), tol, f"grad {k}")


@pytest.mark.parametrize("B,d,K", [(1, 14, 10), (37, 40, 20), (300, 128, 32), (513, 100, 7)])
def test_front_end_vs_closed_form(ed, B, d, K):
    from oracle import eddi_oracle as O
    rng = np.random.default_rng(B + d)
    x = rng.random((B, d), dtype=np.float32)
    m = rng.random((B, d)) < 0.6
    E, tb = rng.normal(size=(d, K)).astype(np.float32), rng.normal(size=(d, 1)).astype(np.float32)
    Wp, cp = rng.normal(size=(K, 2 + K)).astype(np.float32) * 0.3, rng.normal(size=K).astype(np.float32) * 0.3
    dagg = rng.normal(size=(B, K)).astype(np.float32)
    agg_ref, gr = O.front_closed_form(x, m, E, tb, Wp, cp, dagg)
    AC = torch.empty(2, K, d, device="cuda")
    ed.eddi_fold(_dev(E), _dev(tb), _dev(Wp), _dev(cp), AC, d, K)
    agg = torch.full((B, K), float("nan"), device="cuda")
    mu8 = _dev(m.astype(np.uint8))
    ed.eddi_front_fwd(_dev(x), mu8, AC, agg, B, d, K)
    _close(agg, torch.from_numpy(agg_ref), 1e-5, "agg")
    gE, gtb, gWp, gcp = (torch.full(s, float("nan"), device="cuda") for s in ((d, K), (d, 1), (K, 2 + K), (K,)))
    ed.eddi_front_bwd(_dev(x), mu8, AC, _dev(dagg), _dev(E), _dev(tb), _dev(Wp), gE, gtb, gWp, gcp, B, d, K)
    for got, key in ((gE, "type_pars1"), (gtb, "type_bias1"), (gWp, "pnp_encoder1.0.weight"), (gcp, "pnp_encoder1.0.bias")):
        _close(got, torch.from_numpy(gr[key]), 2e-5, key)


@pytest.mark.parametrize("d", [14, 40])
def test_reg_eddi_against_reference(ed, d):
    g = load_golden(f"eddi_reg_d{d}.npz")
    x, m, mp = _dev(g["x"]), _dev(g["mask"]), _dev(g["mask_p"])
    names = ["mean_p", "logvar_p", "x_mean_p", "x_logvar_p", "mean_q", "logvar_q", "x_mean_q", "x_logvar_q"]
    for tag, rt, alpha in (("kl0.5", "kl_reg", 0.5), ("kl1.0", "kl_reg", 1.0), ("ml0.8", "ml_reg", 0.8)):
        model = _load(g, ed.Reg_EDDI, rt)
        torch.manual_seed(0)
        # inject the reference's eps through torch.randn's stream: the three draws happen in forward (q, p) and loss
        draws = iter([_dev(g["eps_q"]), _dev(g["eps_p"]), _dev(g["eps_ml"])])
        orig = torch.randn
        torch.randn = lambda *a, **k: next(draws)
        try:
            o = model.forward(x, m, mp, "train")
            _, tl = model.loss(x, o[2], o[3], o[0], o[1], o[6], o[7], o[4], o[5], m, mp, 1400, beta=0.9, alpha=alpha,
                               beta_annealing=(tag == "kl1.0"))
        finally:
            torch.randn = orig
        for n, t in zip(names, o):
            _close(t.reshape(g["fwd." + n].shape), torch.from_numpy(g["fwd." + n]), 2e-5, n)
        ref = float(g[f"loss.{tag}"])
        assert abs(tl.item() - ref) <= 1e-4 * abs(ref), (tag, tl.item(), ref)
        tl.backward()
        _grad_check(model, g, tag)
    with torch.no_grad():
        r = model.loss(x, o[2], o[3], o[0], o[1], o[6], o[7], o[4], o[5], m, mp, 7, llh_eval=True, stage="evaluate")
    for got, key in zip(r[1:], ("eval_loss", "eval_re", "eval_re_imp")):
        assert abs(float(got.detach() if torch.is_tensor(got) else got) - float(g[key])) <= 1e-4 * abs(float(g[key])), key


@pytest.mark.parametrize("d", [14, 40])
def test_vanilla_eddi_against_reference(ed, d):
    g = load_golden(f"eddi_van_d{d}.npz")
    model = _load(g, ed.vanilla_EDDI)
    x, m = _dev(g["x"]), _dev(g["mask"]).float()
    orig = torch.randn
    torch.randn = lambda *a, **k: _dev(g["eps_q"])
    try:
        o = model.forward(x, m)
    finally:
        torch.randn = orig
    for n, t in zip(["mean", "logvar", "x_mean", "x_logvar"], o):
        _close(t.reshape(g["fwd." + n].shape), torch.from_numpy(g["fwd." + n]), 2e-5, n)
    r = model.loss(x, o[2], o[3], o[0], o[1], 3, m, beta=0.8, llh_eval=True)
    for got, key in zip(r[1:], ("loss", "re", "re_imp")):
        assert abs(float(got.detach() if torch.is_tensor(got) else got) - float(g[key])) <= 1e-4 * abs(float(g[key])), key
    r[1].backward()
    _grad_check(model, g, "v")


@pytest.mark.parametrize("kind", ["reg", "van"])
def test_adam_trajectory(ed, kind):
    g = load_golden(f"eddi_traj_{kind}_d14.npz")
    model = _load(g, ed.Reg_EDDI, "kl_reg", prefix="param0.") if kind == "reg" else _load(g, ed.vanilla_EDDI,
                                                                                          prefix="param0.")
    model.flatten_parameters()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    x, m = _dev(g["x"]), _dev(g["mask"])
    orig = torch.randn
    for s in range(len(g["losses"])):
        draws = iter([_dev(e) for e in g["eps"][s]])
        torch.randn = lambda *a, **k: next(draws)
        try:
            if kind == "reg":
                mp = _dev(g["mask_p"][s])
                o = model.forward(x, m, mp, stage="train")
                _, tl = model.loss(x, o[2], o[3], o[0], o[1], o[6], o[7], o[4], o[5], m, mp, s + 1, alpha=0.5)
            else:
                mf = m.float()
                o = model.forward(x, mf)
                _, tl = model.loss(x, o[2], o[3], o[0], o[1], s + 1, mf)
        finally:
            torch.randn = orig
        opt.zero_grad()
        tl.backward()
        opt.step()
        assert abs(tl.item() - g["losses"][s]) <= 1e-4 * abs(g["losses"][s]), (s, tl.item(), g["losses"][s])
    sd = model.state_dict()
    for k, v in g.items():
        if k.startswith("param5."):
            _close(sd[k[7:]], torch.from_numpy(v), 5e-5, k)


def test_harness_train_and_eval_eddi(ed, tmp_path, monkeypatch):
    """train() (train.py:13-133) and eval_vae (evaluate.py:136-297) for the EDDI families, incl. the 'with_drop'
    keep-mask of create_missing_uci_drop_eddi (utils.py:42-45); checkpoints keep the reference's names."""
    import os
    import vpc_amd
    from torch.utils.data import DataLoader, TensorDataset
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(0)
    x = torch.rand(96, 14)
    m = torch.rand(96, 14) < 0.7
    loader = DataLoader(TensorDataset(x, m), batch_size=32, shuffle=False)
    tp = {"batch_size": 32, "patience": 1}
    for vae_type in ("reg_EDDI1", "vanilla_EDDI1_with_drop"):
        torch.manual_seed(1)
        model = vpc_amd.train((loader, None), 30, 14, 500, 10, 1, 10, "toy", tp, "exp", vae_type, 1, 1, max_epochs=3,
                              alpha=0.5, p_missingness=30, reg_type="kl_reg", verbose=False)
        ck = vpc_amd.checkpoint_path("exp", "toy", vae_type, 30, alpha=0.5, p_missingness=30, reg_type="kl_reg")
        assert os.path.exists(ck) and os.path.basename(os.path.dirname(ck)) == vae_type.split("1")[0]
        res = vpc_amd.eval_vae([(loader, "test")], 30, 14, 500, 10, 2, 10, "toy", tp, "exp", vae_type, 3, 1, 1,
                               alpha=0.5, p_missingness=30, reg_type="kl_reg")
        r = res["test"]
        assert all(torch.isfinite(v) for v in r.values()) and 0.05 < float(r["rmse"]) < 0.6


@pytest.mark.parametrize("kind", ["reg", "van"])
def test_fused_trainer_trajectory(ed, kind):
    """EDDITrainer (front-end + stacked trunk GEMMs + the VAE step's fused decoder kernel + flat Adam) reproduces the
    reference's 5-step trajectory; afterwards the API path sees the updated weights (packed image in step)."""
    g = load_golden(f"eddi_traj_{kind}_d14.npz")
    model = _load(g, ed.Reg_EDDI, "kl_reg", prefix="param0.") if kind == "reg" else _load(g, ed.vanilla_EDDI,
                                                                                          prefix="param0.")
    tr = ed.EDDITrainer(model, lr=1e-3)
    x, m = _dev(g["x"]), _dev(g["mask"])
    for s in range(len(g["losses"])):
        eps = _dev(g["eps"][s])
        if kind == "reg":
            tr.step(x, m, mask_p=_dev(g["mask_p"][s]), eps_q=eps[0], eps_p=eps[1], epoch=s + 1, alpha=0.5)
        else:
            tr.step(x, m, eps_q=eps[0], epoch=s + 1)
        assert abs(tr.loss_value() - g["losses"][s]) <= 1e-4 * abs(g["losses"][s]), (s, tr.loss_value())
    sd = model.state_dict()
    for k, v in g.items():
        if k.startswith("param5."):
            _close(sd[k[7:]], torch.from_numpy(v), 5e-5, k)
    # the decoder image used by the API path follows the in-place Adam updates
    with torch.no_grad():
        z = torch.zeros(4, int(g["L"]), device="cuda")
        xhat, _ = model.decoder(z)
        ref = torch.sigmoid(torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(torch.relu(
            torch.nn.functional.linear(z, sd["seq_decoder.0.weight"], sd["seq_decoder.0.bias"])),
            sd["seq_decoder.2.weight"], sd["seq_decoder.2.bias"])), sd["seq_decoder.4.weight"], sd["seq_decoder.4.bias"]))
    _close(xhat, ref, 2e-5, "decoder after fused steps")


@pytest.mark.parametrize("kind", ["reg", "van"])
def test_fused_trainer_d128_vs_oracle(ed, kind):
    """EDDITrainer at d = 128 (UCI gas width; the shape that goes through dec8_kernel<8>), K = 20, B = 200 against the
    EDDI oracle's torch port (pinned to the reference by eddi_*.npz): loss 1e-4 relative, all 16 gradients 2e-4 of max.
    Reference: src/models/VAE.py:719-741 (front-end), :403-467 / :935-950 (loss)."""
    from oracle import eddi_oracle as EO
    d, K, Ld, B = 128, 20, 10, 200
    torch.manual_seed(31)
    if kind == "reg":
        model = ed.Reg_EDDI(d, 500, K, Ld, {"batch_size": B, "patience": 1}, "exp", "kl_reg")
    else:
        model = ed.vanilla_EDDI(d, 500, K, Ld, {"batch_size": B, "patience": 1}, "exp")
    p = {k: v.detach().clone().float().requires_grad_(True) for k, v in model.state_dict().items() if k in EO.EDDI_KEYS}
    model = model.cuda()
    g = torch.Generator().manual_seed(4)
    x = torch.rand(B, d, generator=g)
    m = torch.rand(B, d, generator=g) < 0.7
    mp = m & (torch.rand(B, d, generator=g) < 0.7)
    eq, ep = torch.randn(B, Ld, generator=g), torch.randn(B, Ld, generator=g)
    port = EO.EDDIPort(p, Ld, "kl_reg")
    if kind == "reg":
        o = port.reg_forward(x, m, mp, eq, ep)
        _, ref = port.reg_loss(x, o[2], o[3], o[0], o[1], o[6], o[7], o[4], o[5], m, mp, 1, alpha=0.5, beta=1.0)
    else:
        o = port.vanilla_forward(x, m, eq)
        _, ref = port.vanilla_loss(x, o[2], o[3], o[0], o[1], 1, m)
    ref.backward()
    tr = ed.EDDITrainer(model, lr=1e-3)
    if kind == "reg":
        tr.step(x.cuda(), m.cuda(), mask_p=mp.cuda(), eps_q=eq.cuda(), eps_p=ep.cuda(), epoch=1, alpha=0.5)
    else:
        tr.step(x.cuda(), m.cuda(), eps_q=eq.cuda(), epoch=1)
    assert abs(tr.loss_value() - ref.item()) <= 1e-4 * abs(ref.item()), (tr.loss_value(), ref.item())
    for k, prm in model.named_parameters():
        if k in p:
            _close(prm.grad, p[k].grad, 2e-4, f"grad {k}")


def test_fused_trainer_device_draws(ed):
    torch.manual_seed(2)
    B, d, K = 256, 100, 20
    x = torch.rand(B, d, device="cuda")
    m = torch.rand(B, d, device="cuda") < 0.7
    finals = []
    for rep in range(2):
        torch.manual_seed(3)
        model = ed.Reg_EDDI(d, 500, K, 10, {"batch_size": B, "patience": 1}, "exp", "kl_reg").cuda()
        tr = ed.EDDITrainer(model, seed=9)
        losses = []
        for s in range(8):
            tr.step(x, m, epoch=s + 1, alpha=0.5, p_missingness=30)
            losses.append(tr.loss_value())
        assert losses[-1] < losses[0]
        finals.append((losses, model._flat.clone()))
    assert finals[0][0] == finals[1][0] and torch.equal(finals[0][1], finals[1][1])


def test_front_end_stacked_passes(ed):
    """Two passes over the same x stacked in one launch (mask, mask_p) == the two single-pass calls: forward bitwise,
    gradients equal to the sum of the per-pass gradients."""
    rng = np.random.default_rng(3)
    B, d, K = 130, 100, 20
    x = _dev(rng.random((B, d), dtype=np.float32))
    m0 = _dev((rng.random((B, d)) < 0.7).astype(np.uint8))
    m1 = m0 * _dev((rng.random((B, d)) < 0.7).astype(np.uint8))
    E, tb = _dev(rng.normal(size=(d, K)).astype(np.float32)), _dev(rng.normal(size=(d, 1)).astype(np.float32))
    Wp, cp = _dev(rng.normal(size=(K, 2 + K)).astype(np.float32) * 0.3), _dev(rng.normal(size=K).astype(np.float32))
    dagg = _dev(rng.normal(size=(2 * B, K)).astype(np.float32))
    AC = torch.empty(2, K, d, device="cuda")
    ed.eddi_fold(E, tb, Wp, cp, AC, d, K)
    agg2 = torch.empty(2 * B, K, device="cuda")
    ed.eddi_front_fwd(x, m0, AC, agg2, B, d, K, mask2_u8=m1)
    a0, a1 = torch.empty(B, K, device="cuda"), torch.empty(B, K, device="cuda")
    ed.eddi_front_fwd(x, m0, AC, a0, B, d, K)
    ed.eddi_front_fwd(x, m1, AC, a1, B, d, K)
    assert torch.equal(agg2[:B], a0) and torch.equal(agg2[B:], a1)
    mk = lambda: [torch.zeros(s, device="cuda") for s in ((d, K), (d, 1), (K, 2 + K), (K,))]
    g2, gs = mk(), mk()
    ed.eddi_front_bwd(x, m0, AC, dagg, E, tb, Wp, *g2, B, d, K, mask2_u8=m1)
    ed.eddi_front_bwd(x, m0, AC, dagg[:B].contiguous(), E, tb, Wp, *gs, B, d, K)
    ed.eddi_front_bwd(x, m1, AC, dagg[B:].contiguous(), E, tb, Wp, *gs, B, d, K, accumulate=True)
    for a, b in zip(g2, gs):
        _close(a, b, 2e-6, "stacked vs per-pass gradients")


def test_with_drop_variant_runs_the_fused_step(ed, tmp_path, monkeypatch):
    """'vanilla_EDDI*_with_drop' (train.py:32-37, 50-51, 95-101; utils.py:42-45): the model is trained on mask * mask_drop.  train()
    now runs that variant as the fused vanilla step on the thinned mask (r02: API path only).  With the same drop masks injected
    the fused form and the reference's own forward -> loss -> backward -> Adam sequence on the API path follow the same
    trajectory, and the keep-mask has the reference's distribution."""
    import vpc_amd
    from vpc_amd import harness
    from torch.utils.data import DataLoader, TensorDataset
    monkeypatch.chdir(tmp_path)
    md = vpc_amd.create_missing_uci_drop_eddi((4000, 50))
    assert md.dtype == torch.float32 and set(md.unique().tolist()) <= {0.0, 1.0}
    assert abs(float(md.mean()) - 0.50005) < 0.01  # 1 - E[min(U, .99)]
    torch.manual_seed(0)
    x = torch.rand(96, 14)
    m = torch.rand(96, 14) < 0.7
    loader = DataLoader(TensorDataset(x, m), batch_size=32, shuffle=False)
    tp = {"batch_size": 32, "patience": 1}
    drops = [(torch.rand(32, 14, generator=torch.Generator().manual_seed(100 + i)) < 0.5).float().cuda() for i in range(9)]
    finals = {}
    for fused in (True, False):
        it = iter(drops)
        monkeypatch.setattr(harness, "create_missing_uci_drop_eddi", lambda shape, device="cuda", generator=None: next(it))
        torch.manual_seed(1)
        model = vpc_amd.train((loader, None), 30, 14, 500, 10, 1, 10, "toy", tp, "exp", "vanilla_EDDI1_with_drop", 1, 1,
                              max_epochs=3, verbose=False, save=False, fused=fused, seed=3)
        finals[fused] = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    # (eps differs between the two forms - device Philox vs torch.randn - so the weights agree statistically, not bitwise: both
    # moved away from the common initialisation by the same nine Adam steps on the same thinned masks)
    for k in finals[True]:
        a, b = finals[True][k].float(), finals[False][k].float()
        assert torch.isfinite(a).all() and float((a - b).abs().max()) <= 2 * 9 * 1e-3 + 1e-6, k
