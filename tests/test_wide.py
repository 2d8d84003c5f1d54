"""Encoder inputs wider than 128 columns: obs_dim = 129 ("UCI gas" with its target column), 200,
and the mask-augmented classes at obs_dim in (64, 128] - the reference takes any obs_dim (src/models/VAE.py:366-376,
527-533).  These shapes run the reference API on the generic GEMM kernels (wide.py); parity against the oracle's torch
port (pinned to the reference by the goldens): forward 2e-5 abs, loss 2e-5 rel, gradients 2e-4 of max."""
import numpy as np
import pytest
import torch

import vpc_amd as vpc
from oracle import vae_oracle as O

pytestmark = pytest.mark.gpu
L = 10
TP = {"batch_size": 64, "patience": 100}


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


def _model(cls, d, params, *extra):
    m = cls(d, 500, 10, L, TP, "exp", *extra)
    sd = m.state_dict(); sd.update({k: v.clone() for k, v in params.items()}); m.load_state_dict(sd)
    return m.cuda()


def _data(B, d, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, d, generator=g)
    mask = torch.rand(B, d, generator=g) < 0.7
    mask_p = mask & (torch.rand(B, d, generator=g) < 0.7)
    return x, mask, mask_p, torch.randn(B, L, generator=g), torch.randn(B, L, generator=g)


@pytest.mark.parametrize("d,B,augm", [(129, 70, False), (200, 257, False), (1000, 33, False), (100, 64, True)])
def test_reg_api_and_trainer_vs_oracle(d, B, augm):
    params = O.init_params(d, L, seed=3, mask_augm=augm)
    x, mask, mask_p, eq, ep = _data(B, d, seed=d)
    loss_ref, grads_ref, outs = O.torch_reg_step(params, L, x, mask, mask_p, eq, ep, alpha=0.8, beta=0.9,
                                                 **({"mask_augm": True} if augm else {}))
    cls = vpc.Reg_VAE_mask if augm else vpc.Reg_VAE
    m = _model(cls, d, params, "kl_reg")
    assert m._wide
    # API path with injected eps
    from vpc_amd.wide import WideDecoderFn, WideEncoderFn
    t = m.trainable()
    xd, md, mpd = x.cuda(), vpc.ops.as_mask_u8(mask.cuda()), vpc.ops.as_mask_u8(mask_p.cuda())
    zq, mq, lq = WideEncoderFn.apply(m, xd, md, eq.cuda(), *t[:6])
    xq = WideDecoderFn.apply(m, zq, *t[6:])
    zp, mp_, lp = WideEncoderFn.apply(m, xd, mpd, ep.cuda(), *t[:6])
    xp = WideDecoderFn.apply(m, zp, *t[6:])
    for got, want in ((mp_, outs[0]), (lp, outs[1]), (xp, outs[2]), (mq, outs[4]), (lq, outs[5]), (xq, outs[6])):
        assert torch.allclose(got.detach().cpu(), want, atol=2e-5)
    _, tl = m.loss(xd, xp, m.x_logvar, mp_, lp, xq, m.x_logvar, mq, lq, mask.cuda(), mask_p.cuda(), 1, beta=0.9, alpha=0.8)
    assert abs(tl.item() - loss_ref.item()) <= 2e-5 * abs(loss_ref.item())
    tl.backward()
    for k, p in zip(O.PARAM_KEYS, t):
        assert rel(p.grad.cpu().numpy(), grads_ref[k].numpy()) < 2e-4, k
    # the public forward (device-side eps) has the reference's return order / shapes
    out = m.forward(xd, mask.cuda(), mask_p.cuda(), "train")
    assert len(out) == 8 and out[2].shape == (B, d) and out[0].shape == (B, L) and out[3].shape == (1,)
    # the training-step object: same loss / gradients, one Adam update moves the parameters
    m2 = _model(cls, d, params, "kl_reg")
    tr = vpc.WideTrainer(m2)
    before = m2._flat.clone()
    tr.step(xd, mask.cuda(), mask_p.cuda(), eq.cuda(), ep.cuda(), alpha=0.8, beta=0.9)
    assert abs(tr.loss_value() - loss_ref.item()) <= 2e-5 * abs(loss_ref.item())
    flat, off = tr.grad.cpu().numpy(), 0
    for k, p in zip(O.PARAM_KEYS, m2.trainable()):
        assert rel(flat[off:off + p.numel()].reshape(p.shape), grads_ref[k].numpy()) < 2e-4, k
        off += p.numel()
    assert not torch.equal(before, m2._flat) and abs(tr.epoch_total() - loss_ref.item()) <= 2e-5 * abs(loss_ref.item())
    with pytest.raises(vpc.VpcError):
        vpc.FusedTrainer(m2)


def test_vanilla_wide_and_train_harness(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    d, B = 129, 96
    params = O.init_params(d, L, seed=5)
    x, mask, _, eq, _ = _data(B, d, seed=9)
    loss_ref, grads_ref, _ = O.torch_vanilla_step(params, L, x, mask, eq)
    m = _model(vpc.vanilla_VAE, d, params)
    tr = vpc.WideTrainer(m)
    tr.step(x.cuda(), mask.cuda(), eps_q=eq.cuda())
    assert abs(tr.loss_value() - loss_ref.item()) <= 2e-5 * abs(loss_ref.item())
    flat, off = tr.grad.cpu().numpy(), 0
    for k, p in zip(O.PARAM_KEYS, m.trainable()):
        assert rel(flat[off:off + p.numel()].reshape(p.shape), grads_ref[k].numpy()) < 2e-4, k
        off += p.numel()
    # train() picks the wide step by itself; device-side draws; the loss goes down and the checkpoint round-trips
    g = torch.Generator().manual_seed(0)
    xs = torch.rand(256, d, generator=g); ms = torch.rand(256, d, generator=g) < 0.7
    loader = [(xs[i:i + 64], ms[i:i + 64]) for i in range(0, 256, 64)]
    torch.manual_seed(1)
    m = vpc.train((loader, None), 30, d, 500, 10, 1, L, "synth", TP, "exp", "reg_vae1", 20, 10, max_epochs=4,
                  device=torch.device("cuda"), alpha=1.0, p_missingness=30, reg_type="kl_reg", verbose=False)
    m2 = vpc.model_loader("test", d, 500, 10, L, 30, "synth", TP, 4, 20, 10, "exp", "kl_reg", "reg_vae1", alpha=1.0,
                          p_missingness=30)
    for a, b in zip(m.state_dict().values(), m2.state_dict().values()):
        assert torch.equal(a.cpu(), b.cpu())
