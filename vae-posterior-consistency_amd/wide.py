"""Reg_VAE / vanilla_VAE for encoder inputs wider than the 128 columns the register-chained kernels tile
(obs_dim > 128, or the mask-augmented classes with 2 * obs_dim > 128) - the reference takes any obs_dim
(src/models/VAE.py:366-376, 527-533; "UCI gas" with its target column appended is d = 129).

Same API path, same loss kernel (K4: vpc_loss_fwd_bwd has no width limit); the six affine layers run on the generic
fp32 MFMA GEMMs of the MNAR path (vpc_linear_fwd / _dgrad / _wgrad: any M, N, K, bias + ReLU / Sigmoid epilogues, ReLU'
and Sigmoid' gates in the backward GEMMs), the encoder input x * mask and the reparameterisation on the small
elementwise kernels of that path (vpc_nm_mul, vpc_nm_sample with K = 1).  Activations go through HBM between layers,
so this is the unfused design of SURVEY.md section 8(d) - correct at any width, not the throughput path.
`WideTrainer` is the training-step object for these models (train.py:53-117): device-side draws, forward, K4, backward,
flat Adam, no host synchronisation.
"""
from __future__ import annotations

import torch

from . import _lib as L
from . import ops
from .notmiwae import (ACT_NONE, ACT_RELU, ACT_SIGMOID_HARDTANH, linear_dgrad, linear_fwd, linear_wgrad, nm_mul,
                       nm_sample, nm_sample_bwd)
from .ops import _f32c, as_mask_u8

H1, H2 = 100, 50


def encoder_input(x, mask_u8, mask_augm):
    """x * mask, or [x * mask | mask] for the mask-augmented classes (VAE.py:388, 545-548)."""
    mf = mask_u8.float()
    xin = torch.empty_like(x)
    nm_mul(x, mf, xin)
    return torch.cat([xin, mf], 1).contiguous() if mask_augm else xin


class WideEncoderFn(torch.autograd.Function):
    """(x, mask, eps, 6 encoder tensors) -> (z, mean, logvar).  VAE.py:387-395."""

    @staticmethod
    def forward(ctx, model, x, mask_u8, eps, W1, b1, W2, b2, W3, b3):
        L.require_cuda(x, mask_u8, eps, W1)
        B, dev, Ld = x.shape[0], x.device, model.latent_dim
        xin = encoder_input(x, mask_u8, model.mask_augm)
        din = xin.shape[1]
        h1, h2 = torch.empty(B, H1, device=dev), torch.empty(B, H2, device=dev)
        heads = torch.empty(B, 2 * Ld, device=dev)
        linear_fwd(xin, W1, b1, h1, B, H1, din, ACT_RELU)
        linear_fwd(h1, W2, b2, h2, B, H2, H1, ACT_RELU)
        linear_fwd(h2, W3, b3, heads, B, 2 * Ld, H2, ACT_NONE)
        z = torch.empty(B, Ld, device=dev)
        nm_sample(heads, eps, z, B, 1, Ld)  # z = mean + eps * exp(logvar / 2); eps None -> z = mean
        ctx.save_for_backward(xin, h1, h2, heads, eps if eps is not None else torch.empty(0, device=dev), W2, W3)
        ctx.has_eps = eps is not None
        mean, logvar = heads[:, :Ld], heads[:, Ld:]  # chunk(2, dim=1): mean first
        return z, mean, logvar

    @staticmethod
    def backward(ctx, dz, dmean, dlogvar):
        xin, h1, h2, heads, eps, W2, W3 = ctx.saved_tensors
        B, dev = xin.shape[0], xin.device
        Ld = heads.shape[1] // 2
        dh = torch.zeros(B, 2 * Ld, device=dev)
        if dmean is not None:
            dh[:, :Ld] += dmean
        if dlogvar is not None:
            dh[:, Ld:] += dlogvar
        dht = torch.empty(B, 2 * Ld, device=dev)
        dzc = _f32c(dz) if dz is not None else torch.zeros(B, Ld, device=dev)
        nm_sample_bwd(dzc, eps if ctx.has_eps else None, heads, dh, dht, B, 1, Ld)
        gW1, gb1 = torch.empty(H1, xin.shape[1], device=dev), torch.empty(H1, device=dev)
        gW2, gb2 = torch.empty(H2, H1, device=dev), torch.empty(H2, device=dev)
        gW3, gb3 = torch.empty(2 * Ld, H2, device=dev), torch.empty(2 * Ld, device=dev)
        dh2, dh1 = torch.empty(B, H2, device=dev), torch.empty(B, H1, device=dev)
        linear_wgrad(dht, h2, gW3, gb3, B, 2 * Ld, H2)
        linear_dgrad(dht, W3, dh2, B, 2 * Ld, H2, x_out=h2, act_prev=ACT_RELU)
        linear_wgrad(dh2, h1, gW2, gb2, B, H2, H1)
        linear_dgrad(dh2, W2, dh1, B, H2, H1, x_out=h1, act_prev=ACT_RELU)
        linear_wgrad(dh1, xin, gW1, gb1, B, H1, xin.shape[1])  # x needs no gradient (layer-0 dgrad skipped)
        return None, None, None, None, gW1, gb1, gW2, gb2, gW3, gb3


class WideDecoderFn(torch.autograd.Function):
    """(z, 6 decoder tensors) -> xhat = sigmoid(MLP(z)).  VAE.py:397-401."""

    @staticmethod
    def forward(ctx, model, z, W4, b4, W5, b5, W6, b6):
        L.require_cuda(z, W4)
        z = _f32c(z)
        B, dev, d, Ld = z.shape[0], z.device, model.obs_dim, model.latent_dim
        g1, g2 = torch.empty(B, H2, device=dev), torch.empty(B, H1, device=dev)
        xhat = torch.empty(B, d, device=dev)
        linear_fwd(z, W4, b4, g1, B, H2, Ld, ACT_RELU)
        linear_fwd(g1, W5, b5, g2, B, H1, H2, ACT_RELU)
        linear_fwd(g2, W6, b6, xhat, B, d, H1, ACT_SIGMOID_HARDTANH, d)  # split = d: Sigmoid on every output
        ctx.save_for_backward(z, g1, g2, xhat, W4, W5, W6)
        return xhat

    @staticmethod
    def backward(ctx, dxhat):
        z, g1, g2, xhat, W4, W5, W6 = ctx.saved_tensors
        B, dev, d, Ld = z.shape[0], z.device, xhat.shape[1], z.shape[1]
        dxhat = _f32c(dxhat)
        gW6, gb6 = torch.empty(d, H1, device=dev), torch.empty(d, device=dev)
        gW5, gb5 = torch.empty(H1, H2, device=dev), torch.empty(H1, device=dev)
        gW4, gb4 = torch.empty(H2, Ld, device=dev), torch.empty(H2, device=dev)
        dg2, dg1, dz = torch.empty(B, H1, device=dev), torch.empty(B, H2, device=dev), torch.empty(B, Ld, device=dev)
        sig = dict(y_gate=xhat, gate=ACT_SIGMOID_HARDTANH, gate_split=d)  # dpre = dxhat * xhat (1 - xhat) inside the GEMMs
        linear_wgrad(dxhat, g2, gW6, gb6, B, d, H1, **sig)
        linear_dgrad(dxhat, W6, dg2, B, d, H1, x_out=g2, act_prev=ACT_RELU, **sig)
        linear_wgrad(dg2, g1, gW5, gb5, B, H1, H2)
        linear_dgrad(dg2, W5, dg1, B, H1, H2, x_out=g1, act_prev=ACT_RELU)
        linear_wgrad(dg1, z, gW4, gb4, B, H2, Ld)
        linear_dgrad(dg1, W4, dz, B, H2, Ld)
        return None, dz, gW4, gb4, gW5, gb5, gW6, gb6


class WideTrainer:
    """Training step (train.py:53-117) for the wide models: on-device mask_p / eps draws, the API-path forward and K4
    loss, backward, flat Adam (vpc_adam_step on the flat parameter buffer), loss accumulated on the device.  Mirrors
    FusedTrainer's interface (step / loss_value / epoch_total); data parallel as there: ONE all-reduce of [grads | loss]."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, seed=0, process_group=None, world_size=1, rank=0):
        self.model, self.lr, self.betas, self.eps = model, lr, betas, eps
        self.seed, self.rng_offset, self.step_count = seed, 0, 0
        self.pg, self.world_size, self.rank = process_group, world_size, rank
        flat = model.flatten_parameters()
        L.require_cuda(flat)
        self.dev = flat.device
        n = flat.numel()
        self.bucket = torch.zeros(n + 1, device=self.dev)
        self.grad, self.loss = self.bucket[:n], self.bucket[n:]
        self.exp_avg, self.exp_avg_sq = torch.zeros(n, device=self.dev), torch.zeros(n, device=self.dev)
        self.accum = torch.zeros(1, device=self.dev)
        self.vanilla = not hasattr(model, "reg_type")

    def step(self, x, mask, mask_p=None, eps_q=None, eps_p=None, *, epoch=1, alpha=1.0, beta=1.0, beta_annealing=False,
             p_missingness=30, global_batch=None, row_lo=None):
        m = self.model
        x = _f32c(x)
        B, d, Ld = x.shape[0], m.obs_dim, m.latent_dim
        Bg = global_batch if global_batch is not None else B * self.world_size
        if row_lo is None:
            row_lo = self.rank * B if self.world_size > 1 else 0
        mu8 = as_mask_u8(mask)
        two = not self.vanilla
        if two and mask_p is None:  # Philox counters of the GLOBAL row
            mask_p = torch.empty(B, d, dtype=torch.uint8, device=self.dev)
            ops.draw_mask(mu8, mask_p, 1.0 - p_missingness / 100.0, self.seed, self.rng_offset, row_lo * d)
            self.rng_offset += (Bg * d + 7) // 8 + 1
        n_eps = 2 if two else 1
        if eps_q is None or (two and eps_p is None):
            e = torch.empty(n_eps, B, 4 * ((Ld + 3) // 4), device=self.dev)
            ops.fill_normal(e, self.seed, self.rng_offset, None, (B, Bg, row_lo, e.shape[2]))
            self.rng_offset += n_eps * Bg * (e.shape[2] // 4)
            eps_q = e[0, :, :Ld].contiguous() if eps_q is None else eps_q
            eps_p = (e[1, :, :Ld].contiguous() if eps_p is None else eps_p) if two else None
        for p in m.trainable():
            p.grad = None
        t = m.trainable()
        zq, mq, lq = WideEncoderFn.apply(m, x, mu8, _f32c(eps_q), *t[:6])
        xq = WideDecoderFn.apply(m, zq, *t[6:])
        if two:
            mpu8 = as_mask_u8(mask_p)
            zp, mp_, lp = WideEncoderFn.apply(m, x, mpu8, _f32c(eps_p), *t[:6])
            xp = WideDecoderFn.apply(m, zp, *t[6:])
            _, tl = m.loss(x, xp, m.x_logvar, mp_, lp, xq, m.x_logvar, mq, lq, mu8, mpu8, epoch, beta_annealing=beta_annealing,
                           beta=beta, alpha=alpha, stage="train")
        else:
            _, tl = m.loss(x, xq, m.x_logvar, mq, lq, epoch, mu8, beta_annealing=beta_annealing, beta=beta, stage="train")
        tl = tl * (B / Bg)  # every rank normalises by the GLOBAL batch: SUM over ranks = the concatenated batch
        tl.backward()
        off = 0
        for p in t:
            self.grad[off:off + p.numel()].copy_(p.grad.reshape(-1))
            off += p.numel()
        self.loss.copy_(tl.detach().reshape(1))
        if self.world_size > 1:  # ONE collective per step: RCCL on the compute stream, or torch.distributed (dist.py)
            from . import dist as dp_mod
            if not getattr(self, "_coll_ready", False):
                self.collective = dp_mod.make_collective(self.world_size, self.rank, self.dev, self.pg)
                self._coll_ready = True
            dp_mod.allreduce_bucket(self.bucket, self.pg, self.collective)
        self.step_count += 1
        ops.adam_step(m._flat, self.grad, self.exp_avg, self.exp_avg_sq, self.step_count, self.lr, self.betas[0],
                      self.betas[1], self.eps, loss_in=self.loss, accum=self.accum)

    def loss_value(self) -> float:
        return float(self.loss.item())

    def epoch_total(self, reset=True) -> float:
        v = float(self.accum.item())
        if reset:
            self.accum.zero_()
        return v
