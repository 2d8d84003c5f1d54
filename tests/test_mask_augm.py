"""Mask-augmented variants (Reg_VAE_mask / vanilla_VAE_mask, src/models/VAE.py:510-667, 995-1116; SURVEY f-2):
encoder input [x*mask | mask], first layer 2d -> 100.  CPU: oracle vs reference goldens; GPU: API path and fused
step vs the same goldens."""
import numpy as np
import pytest
import torch

import vpc_amd as vpc
from conftest import load_golden
from oracle import vae_oracle as O

L = 10
TP = {"batch_size": 48, "patience": 1}


def _t(a, dev="cpu"):
    return torch.from_numpy(np.array(a)).to(dev)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


def _params(g, kind):
    pre = f"{kind}.param."
    return {k[len(pre):]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith(pre) and "prior" not in k}


def test_oracle_mask_augm_matches_reference():
    g = load_golden("maskaugm_d14.npz")
    x, m, mp = _t(g["x"]), _t(g["mask"]), _t(g["mask_p"])
    loss, grads, outs = O.torch_reg_step(_params(g, "reg"), L, x, m, mp, _t(g["reg.eps_q"]), _t(g["reg.eps_p"]),
                                         alpha=0.7, beta=0.9, mask_augm=True)
    assert abs(loss.item() - float(g["reg.loss"])) <= 2e-6 * float(g["reg.loss"])
    # same ops on the same inputs: identical up to the host BLAS kernel choice (1 ulp between CPU models)
    assert np.allclose(outs[4].numpy(), g["reg.mean_q"], rtol=2e-6, atol=1e-7)
    assert np.allclose(outs[2].numpy(), g["reg.x_mean_p"], rtol=2e-6, atol=1e-7)
    for k in O.PARAM_KEYS:
        assert rel(grads[k].numpy(), g[f"reg.grad.{k}"]) < 5e-6, k
    mf = m * torch.ones(x.shape)
    loss, grads, outs = O.torch_vanilla_step(_params(g, "vanilla"), L, x, mf, _t(g["vanilla.eps_q"]), beta=0.9,
                                             mask_augm=True)
    assert abs(loss.item() - float(g["vanilla.loss"])) <= 2e-6 * float(g["vanilla.loss"])
    for k in O.PARAM_KEYS:
        assert rel(grads[k].numpy(), g[f"vanilla.grad.{k}"]) < 5e-6, k


def test_model_loader_dispatch_and_shapes():
    m = vpc.model_loader("train", 14, 500, 10, L, 30, "synth", TP, 1, 1, 1, "exp", "kl_reg", "reg_vae1_mask_augm")
    assert isinstance(m, vpc.Reg_VAE_mask) and m.seq_encoder[0].weight.shape == (100, 28)
    m = vpc.model_loader("train", 14, 500, 10, L, 30, "synth", TP, 1, 1, 1, "exp", "kl_reg", "vanilla_vae2_mask_augm")
    assert isinstance(m, vpc.vanilla_VAE_mask)
    m = vpc.Reg_VAE_mask(65, 500, 10, L, TP, "exp", "kl_reg")  # 2d > 128: the generic-GEMM path (wide.py, tests/test_wide.py)
    assert m._wide and m.seq_encoder[0].weight.shape == (100, 130)
    with pytest.raises(vpc.VpcError):
        vpc.Reg_VAE(14, 500, 10, 65, TP, "exp", "kl_reg")  # latent_dim > 64


def _model(cls, g, kind, **kw):
    m = cls(14, 500, 10, L, TP, "exp", "kl_reg") if kind == "reg" else cls(14, 500, 10, L, TP, "exp")
    sd = m.state_dict(); sd.update({k: v.clone() for k, v in _params(g, kind).items()}); m.load_state_dict(sd)
    return m.to("cuda")


@pytest.mark.gpu
def test_gpu_mask_augm_api_and_fused():
    g = load_golden("maskaugm_d14.npz")
    x, mk, mp = _t(g["x"], "cuda"), _t(g["mask"], "cuda"), _t(g["mask_p"], "cuda")
    # ---- Reg_VAE_mask, API path with injected eps
    m = _model(vpc.Reg_VAE_mask, g, "reg")
    zq, mq, lq = vpc.ops.EncoderFn.apply(m, x, vpc.ops.as_mask_u8(mk), _t(g["reg.eps_q"], "cuda"), *m.trainable()[:6])
    xq, xlv = m.decoder(zq)
    zp, mpn, lp = vpc.ops.EncoderFn.apply(m, x, vpc.ops.as_mask_u8(mp), _t(g["reg.eps_p"], "cuda"), *m.trainable()[:6])
    xp, _ = m.decoder(zp)
    assert np.allclose(mq.detach().cpu().numpy(), g["reg.mean_q"], atol=2e-5)
    assert np.allclose(xp.detach().cpu().numpy(), g["reg.x_mean_p"], atol=2e-5)
    _, tl = m.loss(x, xp, xlv, mpn, lp, xq, xlv, mq, lq, mk, mp, 1, beta=0.9, alpha=0.7, stage="train")
    tl.backward()
    assert abs(tl.item() - float(g["reg.loss"])) <= 2e-5 * float(g["reg.loss"])
    for k, p in zip(O.PARAM_KEYS, m.trainable()):
        assert rel(p.grad.cpu().numpy(), g[f"reg.grad.{k}"]) < 2e-4, k
    # ---- fused step
    m = _model(vpc.Reg_VAE_mask, g, "reg")
    tr = vpc.FusedTrainer(m)
    tr.step(x, mk, mp, _t(g["reg.eps_q"], "cuda"), _t(g["reg.eps_p"], "cuda"), alpha=0.7, beta=0.9, update=False)
    assert abs(tr.loss_value() - float(g["reg.loss"])) <= 2e-5 * float(g["reg.loss"])
    off, flat = 0, tr.grad.cpu().numpy()
    for k, p in zip(O.PARAM_KEYS, m.trainable()):
        assert rel(flat[off:off + p.numel()].reshape(p.shape), g[f"reg.grad.{k}"]) < 2e-4, k
        off += p.numel()
    # ---- vanilla_VAE_mask, fused
    m = _model(vpc.vanilla_VAE_mask, g, "vanilla")
    tr = vpc.FusedTrainer(m)
    tr.step(x, mk, eps_q=_t(g["vanilla.eps_q"], "cuda"), beta=0.9, update=False)
    assert abs(tr.loss_value() - float(g["vanilla.loss"])) <= 2e-5 * float(g["vanilla.loss"])
    off, flat = 0, tr.grad.cpu().numpy()
    for k, p in zip(O.PARAM_KEYS, m.trainable()):
        assert rel(flat[off:off + p.numel()].reshape(p.shape), g[f"vanilla.grad.{k}"]) < 2e-4, k
        off += p.numel()
    out = m.forward(x, mk * torch.ones(x.shape, device="cuda"))
    assert np.allclose(out[0].detach().cpu().numpy(), g["vanilla.mean_q"], atol=2e-5)
