"""The bf16 engine against a bf16-EMULATING oracle.

tests/test_bf16.py compares the bf16 / bf16x3 kernels with fp32 results under the precision's own (loose) tolerance:
0.15 of max |g| cannot see a mis-permuted k-slot in one tile.  Here the float64 oracle rounds every matrix-product
operand exactly where the kernels round it (oracle/vae_oracle.py GemmModel, oracle/notmiwae_oracle.py rounded_linear), so
what is left between the two is fp32 accumulation order: loss <= 1e-5 relative, every gradient tensor <= 1e-3 of its max
(a ReLU gate can still flip where a pre-activation sits within fp32 rounding of zero; with both sides rounding alike that
is rare and moves an entry by O(1/B)).  The emulation itself is pinned on the CPU: bf16_round is bit-equal to torch's
bfloat16 cast, "f64" mode is the unmodified closed form (itself checked against the reference's vectors in
tests/test_oracle_golden.py), and the split form agrees with exact products to ~2^-16.
"""
import os
import subprocess

import numpy as np
import pytest
import torch

import vpc_amd as vpc
from conftest import ROOT, golden_params, load_golden
from oracle import notmiwae_oracle as NO
from oracle import vae_oracle as O

L = 10
TP = {"batch_size": 64, "patience": 100}
PRECS = ["bf16x3", "bf16"]
LOSS_TOL, GRAD_TOL = 1e-5, 1e-3


@pytest.fixture(autouse=True)
def _requested_engine(monkeypatch):
    """These tests are about the bf16 engine's kernels.  FusedTrainer routes small batches of EVERY precision to the fp32 N-split
    kernel (csrc/vpc_small.hip: more accurate and faster there); VPC_STEP_SMALL=0 keeps the engine that was asked for."""
    monkeypatch.setenv("VPC_STEP_SMALL", "0")


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))


# ------------------------------------------------------------------------------------------------ CPU: the emulation
def test_bf16_round_is_torch_bfloat16():
    g = torch.Generator().manual_seed(0)
    v = torch.cat([torch.randn(20000, generator=g) * 10 ** torch.randint(-6, 6, (20000,), generator=g).float(),
                   torch.tensor([0.0, -0.0, 1.0, 1.00390625, 1.01171875, 3.0e38, 1e-40])])
    want = v.to(torch.bfloat16).to(torch.float64).numpy()
    assert np.array_equal(O.bf16_round(v.numpy()), want)
    hi, lo = O.bf16_split(v.numpy())
    assert np.array_equal(hi, want)
    ok = np.abs(v.numpy()) > 1e-30
    assert np.max(np.abs(hi + lo - v.double().numpy())[ok] / np.abs(v.double().numpy())[ok]) < 2.0 ** -15


def test_emulated_step_brackets_the_exact_one():
    g = load_golden("reg_d128.npz")
    P = golden_params(g)
    args = (P, L, g["x"], g["mask"], g["mask_p"], g["eps_q"], g["eps_p"])
    l64, g64, _, _ = O.closed_form_reg_step(*args, alpha=1.0)
    assert abs(l64 - float(g["loss_a1"])) <= 2e-6 * abs(l64)  # "f64" is the closed form the goldens pin
    l3, g3, _, _ = O.closed_form_reg_step(*args, alpha=1.0, gemm="bf16x3")
    l1, g1, _, _ = O.closed_form_reg_step(*args, alpha=1.0, gemm="bf16")
    assert abs(l3 - l64) <= 2e-5 * abs(l64) and 1e-7 * abs(l64) < abs(l1 - l64) <= 5e-3 * abs(l64)
    k = "seq_decoder.4.weight"
    assert rel(g3[k], g64[k]) < 5e-3 and 1e-4 < rel(g1[k], g64[k]) < 0.25


def test_rounded_linear_autograd():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(7, 3, 20, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(11, 20, generator=g, dtype=torch.float64, requires_grad=True)
    b = torch.randn(11, generator=g, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(7, 3, 11, generator=g, dtype=torch.float64)
    y = NO.rounded_linear("bf16")(x, w, b)
    y.backward(dy)
    r = NO._bf16_t
    assert torch.allclose(y, r(x) @ r(w).t() + b, rtol=0, atol=1e-12)
    assert torch.allclose(x.grad, r(dy) @ r(w), atol=1e-12)
    assert torch.allclose(w.grad, r(dy).reshape(-1, 11).t() @ r(x).reshape(-1, 20), atol=1e-12)
    assert torch.allclose(b.grad, dy.reshape(-1, 11).sum(0), atol=1e-12)
    x.grad = w.grad = b.grad = None
    y3 = NO.rounded_linear("bf16x3")(x, w, b)
    assert float((y3 - (x @ w.t() + b)).abs().max()) < 1e-3 * float(y3.abs().max()) * 2.0 ** -6


# ------------------------------------------------------------------------------------------------ GPU: the kernels
def _model(cls, d, params, reg_type="kl_reg"):
    m = cls(d, 500, 10, L, TP, "exp", *([reg_type] if cls is vpc.Reg_VAE else []))
    sd = m.state_dict()
    sd.update({k: v.clone() for k, v in params.items()})
    m.load_state_dict(sd)
    return m.cuda()


def _check(tr, model, loss_ref, grads_ref):
    assert abs(tr.loss_value() - loss_ref) <= LOSS_TOL * abs(loss_ref), (tr.loss_value(), loss_ref)
    flat, off = tr.grad.cpu().numpy(), 0
    for k, p in zip(O.PARAM_KEYS, model.trainable()):
        e = rel(flat[off:off + p.numel()].reshape(p.shape), grads_ref[k])
        assert e < GRAD_TOL, (k, e)
        off += p.numel()


def _t(a):
    return torch.from_numpy(np.array(a)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("tag,kw", [
    ("a1", dict(alpha=1.0, beta=1.0, beta_annealing=False, epoch=1)),
    ("a05", dict(alpha=0.5, beta=0.7, beta_annealing=True, epoch=1400)),
    ("ml", dict(alpha=0.8, beta=1.0, beta_annealing=False, epoch=1400)),
])
@pytest.mark.parametrize("tile", ["auto", "64", "128"])
def test_reg_step_on_reference_vectors(prec, tag, kw, tile, monkeypatch):
    """reg_d128.npz inputs (the reference's own), both workgroup shapes: the 64-row fixture runs the small-batch kernels
    with VPC_TILE=64 and the throughput kernels with VPC_TILE=128; plain bf16 runs the whole-step kernel unless VPC_TILE=64
    (VPC_STEP_SMALL=0 is set for this module: otherwise 64 rows would go to the fp32 N-split kernel)."""
    if tile != "auto":
        monkeypatch.setenv("VPC_TILE", tile)
    g = load_golden("reg_d128.npz")
    rt = "ml_reg" if tag == "ml" else "kl_reg"
    P = golden_params(g)
    m = _model(vpc.Reg_VAE, 128, P, rt)
    tr = vpc.FusedTrainer(m, precision=prec)
    tr.step(_t(g["x"]), _t(g["mask"]), _t(g["mask_p"]), _t(g["eps_q"]), _t(g["eps_p"]),
            _t(g["eps_ml"]) if tag == "ml" else None, update=False, **kw)
    loss, grads, _, _ = O.closed_form_reg_step(P, L, g["x"], g["mask"], g["mask_p"], g["eps_q"], g["eps_p"], reg_type=rt,
                                               eps_ml=g["eps_ml"] if tag == "ml" else None, gemm=prec,
                                               db1_rounded=tr._used_step_fused, **kw)
    assert tr._used_step_fused == (prec == "bf16" and tile != "64")  # the whole-step kernel is what ran there
    _check(tr, m, loss, grads)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("tile", ["auto", "64", "128"])
def test_vanilla_step_on_reference_vectors(prec, tile, monkeypatch):
    if tile != "auto":
        monkeypatch.setenv("VPC_TILE", tile)
    g = load_golden("vanilla_d128.npz")
    P = golden_params(g)
    m = _model(vpc.vanilla_VAE, 128, P)
    tr = vpc.FusedTrainer(m, precision=prec)
    tr.step(_t(g["x"]), _t(g["mask"]), eps_q=_t(g["eps_q"]), update=False)
    loss, grads, _ = O.closed_form_vanilla_step(P, L, g["x"], g["mask"], g["eps_q"], gemm=prec,
                                                db1_rounded=tr._used_step_fused)
    assert tr._used_step_fused == (prec == "bf16" and tile != "64")
    _check(tr, m, loss, grads)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("d,B,tile", [(128, 1000, "auto"), (128, 1000, "64"), (128, 1000, "128"), (100, 300, "128"), (100, 300, "64"),
                                      (72, 129, "128"), (40, 300, "auto"), (16, 100, "128"), (128, 20000, "auto")])
def test_reg_step_ragged_shapes(prec, d, B, tile, monkeypatch):
    """B = 1 000 and ragged shapes in both workgroup shapes (d <= 64: the 4-wave decoder; d in (64, 128]: the fused bf16 step kernel
    unless VPC_TILE=64 keeps the three small-shape kernels; bf16x3: the 8-wave kernels with VPC_TILE=128 or B > 16 384)."""
    if tile != "auto":
        monkeypatch.setenv("VPC_TILE", tile)
    P = O.init_params(d, L, seed=7)
    g = torch.Generator().manual_seed(B + d)
    x = torch.rand(B, d, generator=g)
    mask = torch.rand(B, d, generator=g) < 0.7
    mask_p = mask & (torch.rand(B, d, generator=g) < 0.7)
    eq, ep = torch.randn(B, L, generator=g), torch.randn(B, L, generator=g)
    m = _model(vpc.Reg_VAE, d, P)
    tr = vpc.FusedTrainer(m, precision=prec)
    tr.step(x.cuda(), mask.cuda(), mask_p.cuda(), eq.cuda(), ep.cuda(), alpha=0.8, beta=0.9, update=False)
    loss, grads, _, _ = O.closed_form_reg_step(P, L, x.numpy(), mask.numpy(), mask_p.numpy(), eq.numpy(), ep.numpy(),
                                               alpha=0.8, beta=0.9, gemm=prec, db1_rounded=tr._used_step_fused)
    assert tr._used_step_fused == (prec == "bf16" and d > 64 and tile != "64")
    _check(tr, m, loss, grads)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("kind", ["reg", "van"])
def test_mnar_trainer_vs_emulating_oracle(prec, kind):
    """NMTrainer (config 3: d = 128, K = 20, batch 128, p_missingness 50) with bf16 / bf16x3 GEMMs against the oracle's
    port in float64 with the same operand rounding in every forward, dgrad and wgrad product."""
    from vpc_amd import notmiwae as nm
    d, K, B = 128, 20, 128
    torch.manual_seed(5)
    cls = nm.REG_notMIWAE_v2 if kind == "reg" else nm.notMIWAE_myversion
    model = cls(d, 128, 10, L, {"batch_size": B, "patience": 1}, K, 1)
    p = {k: v.detach().clone().double().requires_grad_(True) for k, v in model.state_dict().items() if k in NO.NM_KEYS}
    model = model.cuda()
    g = torch.Generator().manual_seed(2)
    x = torch.rand(B, d, generator=g)
    m = (torch.rand(B, d, generator=g) < 0.7).float()
    mp = m * (torch.rand(B, d, generator=g) < 0.5).float()
    eps = torch.randn(2, B, K, L, generator=g)
    tr = nm.NMTrainer(model, precision=prec)
    tr.step(x.cuda(), m.cuda(), mask_p=mp.cuda() if kind == "reg" else None, eps=eps.cuda(), alpha=0.5, p_missingness=50)
    # (plain bf16 runs the layer-fused decoder kernel at this shape: tests/test_nmdec.py - the port then models that kernel's
    # rounding points on the decoder side)
    assert tr.use_nmdec == (prec == "bf16")
    port = NO.NMTorchPort(p, L, K, kind == "reg", linear=NO.rounded_linear(prec), fused_decoder=tr.use_nmdec)
    xd, md, mpd, ed = x.double(), m.double(), mp.double(), eps.double()
    if kind == "reg":
        ref = port.reg_loss(xd, port.reg_forward(xd, md, mpd, ed[0], ed[1]), md, mpd, alpha=0.5)
    else:
        ref = port.van_loss(xd, port.van_forward(xd, md, ed[0]), md, ed[1])
    ref.backward()
    assert abs(tr.loss_value() - ref.item()) <= LOSS_TOL * abs(ref.item()), (tr.loss_value(), ref.item())
    for k, prm in model.named_parameters():
        if k in p and p[k].grad is not None:
            e = rel(prm.grad.cpu().numpy(), p[k].grad.numpy())
            # 2e-3 here: Hardtanh(-10, 0) on the log-variance head gates its gradient like a ReLU, and at the initial weights
            # many pre-activations sit within fp32 accumulation error of the clamp at 0
            assert e < 2 * GRAD_TOL, (k, e)


@pytest.mark.gpu
def test_bf16_engine_microtest(tmp_path):
    """tools/microbench/bf16_engine_test.hip (image layout, K permutation, swizzle, forward and ds_read_b64_tr_b16 fragments
    of csrc/vpc_bf16.h against a float64 reference, one wave) - built here with hipcc and run as a child process."""
    src = os.path.join(ROOT, "tools", "microbench", "bf16_engine_test.hip")
    exe = str(tmp_path / "bf16_engine_test")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    b = subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "vae-posterior-consistency_amd", "csrc"),
                        src, "-o", exe], capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("prec", PRECS)
def test_small_batches_run_the_f32_kernel_in_every_precision(prec, monkeypatch):
    """Without VPC_STEP_SMALL=0 a FusedTrainer(precision=bf16 | bf16x3) step on the reference's 64-row vectors runs the fp32
    N-split kernel: bit-equal to the fp32 trainer's step, i.e. inside the fp32 tolerances of tests/test_gpu_parity.py."""
    monkeypatch.delenv("VPC_STEP_SMALL", raising=False)
    g = load_golden("reg_d128.npz")
    P = golden_params(g)
    outs = []
    for p_ in ("f32", prec):
        m = _model(vpc.Reg_VAE, 128, P)
        tr = vpc.FusedTrainer(m, precision=p_)
        tr.step(_t(g["x"]), _t(g["mask"]), _t(g["mask_p"]), _t(g["eps_q"]), _t(g["eps_p"]), None, update=False, alpha=1.0)
        assert tr.dominant_launch() == "step_small"
        outs.append((tr.loss_value(), tr.grad.clone()))
    assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1])
    loss, _, _, _ = O.closed_form_reg_step(P, L, g["x"], g["mask"], g["mask_p"], g["eps_q"], g["eps_p"], alpha=1.0)
    assert abs(outs[1][0] - loss) <= 2e-5 * abs(loss)
