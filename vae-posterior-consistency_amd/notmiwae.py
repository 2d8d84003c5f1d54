"""MNAR path (BASELINE config 3): drop-in classes for the reference's

    REG_notMIWAE_v2      src/models/VAE.py:2327-2505
    notMIWAE_myversion   src/models/VAE.py:2691-2847

with the same constructor arguments, `encoder` / `decoder` / `forward` / `loss` signatures, return order and
state_dict keys (W, b, seq_encoder.{0,2}, q_mu.0, q_logstd.0, seq_decoder.{0,2}, x_mean.0, x_logvar.0 and, for the
regularised class, the unused float64 `logits.0`).  Every layer runs as an fp32 MFMA GEMM (csrc/vpc_gemm.hip), the
importance-weighted loss with the self-masking missingness model and all of its gradients in one fused kernel
(csrc/vpc_nm.hip).  The two heads of the encoder (q_mu | q_logstd) and of the decoder (x_mean | x_logvar) are
adjacent in one flat parameter buffer, so each pair is ONE GEMM.  No CPU fallback: CPU tensors raise.
"""
from __future__ import annotations

import ctypes as C
import math

import torch
import torch.nn as nn

from . import _lib as L
from ._lib import check, lib, ptr, require_cuda, stream_ptr

HID = 128  # VAE.py:2343-2363 hard-codes 128 (hid_dim is ignored by the reference too)
ACT_NONE, ACT_ELU, ACT_SIGMOID_HARDTANH, ACT_RELU = 0, 1, 2, 3


# ------------------------------------------------------------------------------------------------ raw ops
def linear_fwd(x, w, b, y, M, N, K, act=ACT_NONE, split=0, ldx=None, ldy=None, precision=0):
    check(lib().vpc_linear_fwd(ptr(x), ldx or K, ptr(w), ptr(b), ptr(y), ldy or N, M, N, K, act, split, int(precision),
                               stream_ptr()), "vpc_linear_fwd")


def linear_dgrad(dy, w, dx, M, N, K, y_gate=None, gate=ACT_NONE, gate_split=0, x_out=None, act_prev=ACT_NONE,
                 lddy=None, lddx=None, precision=0):
    check(lib().vpc_linear_dgrad(ptr(dy), lddy or N, ptr(y_gate), lddy or N, gate, gate_split, ptr(w), ptr(x_out), K,
                                 act_prev, ptr(dx), lddx or K, M, N, K, int(precision), stream_ptr()), "vpc_linear_dgrad")


_scratch = {}


def _wgrad_scratch(device, floats):
    key = str(device)
    buf = _scratch.get(key)
    if buf is None or buf.numel() < floats:
        buf = torch.empty(max(floats, 1 << 20), device=device)
        _scratch[key] = buf
    return buf


def linear_wgrad(dy, x, dw, db, M, N, K, y_gate=None, gate=ACT_NONE, gate_split=0, accumulate=False, lddy=None,
                 ldx=None, precision=0, scratch=None):
    """dw = None: write only the per-split partials into `scratch` (the caller's own buffer for this layer); they are summed
    later, together with other layers', by wgrad_reduce (one launch)."""
    sc = _wgrad_scratch(dy.device, int(lib().vpc_linear_wgrad_scratch(M, N, K))) if scratch is None else scratch
    check(lib().vpc_linear_wgrad(ptr(dy), lddy or N, ptr(y_gate), lddy or N, gate, gate_split, ptr(x), ldx or K,
                                 ptr(dw), ptr(db), ptr(sc), sc.numel(), M, N, K, int(accumulate), int(precision),
                                 stream_ptr()), "vpc_linear_wgrad")


def wgrad_reduce(layers, cache=None):
    """layers: [(scratch, M, N, K, dw, db, accumulate)] of linear_wgrad(dw=None) calls -> all gradients in ONE launch.
    `cache` (a dict owned by the caller): the argument arrays are built once per set of buffers, not per step."""
    if cache is not None and "args" in cache:
        check(lib().vpc_linear_wgrad_reduce(*cache["args"], stream_ptr()), "vpc_linear_wgrad_reduce")
        return
    n = len(layers)
    sc = (C.c_void_p * n)(*[t[0].data_ptr() for t in layers])
    Ms = (C.c_long * n)(*[int(t[1]) for t in layers])
    Ns = (C.c_int * n)(*[int(t[2]) for t in layers])
    Ks = (C.c_int * n)(*[int(t[3]) for t in layers])
    dw = (C.c_void_p * n)(*[t[4].data_ptr() for t in layers])
    db = (C.c_void_p * n)(*[None if t[5] is None else t[5].data_ptr() for t in layers])
    acc = (C.c_int * n)(*[int(bool(t[6])) for t in layers])
    if cache is not None:
        cache["args"] = (n, sc, Ms, Ns, Ks, dw, db, acc)
    check(lib().vpc_linear_wgrad_reduce(n, sc, Ms, Ns, Ks, dw, db, acc, stream_ptr()), "vpc_linear_wgrad_reduce")


def nm_sample(heads, eps, z, B, K, Ld):
    check(lib().vpc_nm_sample(ptr(heads), 2 * Ld, ptr(eps), ptr(z), Ld, B, K, Ld, stream_ptr()), "vpc_nm_sample")


def nm_sample_bwd(dz, eps, heads, g_heads, out, B, K, Ld):
    check(lib().vpc_nm_sample_bwd(ptr(dz), Ld, ptr(eps), ptr(heads), 2 * Ld, ptr(g_heads), 2 * Ld, ptr(out), 2 * Ld, B,
                                  K, Ld, stream_ptr()), "vpc_nm_sample_bwd")


def nm_mul(x, mask, out):
    check(lib().vpc_nm_mul(ptr(x), ptr(mask), ptr(out), out.numel(), stream_ptr()), "vpc_nm_mul")


def nm_prep(x, mask, mask_p_out, xin, B, d, keep_prob, seed, offset, eps_out=None, offset_eps=0, state=None,
            elem_lo=0, eps_shard=None):
    """eps_shard = (rows_local, rows_global, row_lo, pitch) of eps_out as rows of a global batch (None: flat)."""
    sh = (0, 0, 0, 4) if eps_shard is None else tuple(int(v) for v in eps_shard)
    check(lib().vpc_nm_prep(ptr(x), ptr(mask), ptr(mask_p_out), ptr(xin), B, d, float(keep_prob), ptr(eps_out),
                            0 if eps_out is None else eps_out.numel(), int(seed), int(offset), int(offset_eps),
                            ptr(state), int(elem_lo), *sh, stream_ptr()), "vpc_nm_prep")


def nm_loss(x, mask, mask_p, xm_q, xl_q, ldq, xm_p, xl_p, ldp, hq, hp, W, b, eps_kl, g_xm_q, g_xl_q, g_xm_p, g_xl_p,
            ldg, ghq, ghp, gW, gb, xm_imp, scratch, out8, loss_f32, accum, B, B_global, K, d, Ld, alpha, state=None,
            rng_inc=0, gated=False):
    check(lib().vpc_nm_loss(ptr(x), ptr(mask), ptr(mask_p), ptr(xm_q), ptr(xl_q), ldq, ptr(xm_p), ptr(xl_p), ldp,
                            ptr(hq), ptr(hp), 2 * Ld, ptr(W), ptr(b), ptr(eps_kl), ptr(g_xm_q), ptr(g_xl_q), ldg,
                            ptr(g_xm_p), ptr(g_xl_p), ldg, ptr(ghq), ptr(ghp), 2 * Ld, ptr(gW), ptr(gb), 0,
                            ptr(xm_imp), ptr(scratch), scratch.numel() * scratch.element_size(), ptr(out8),
                            ptr(loss_f32), ptr(accum), ptr(state), int(rng_inc), int(gated), B, B_global, K, d, Ld,
                            float(alpha), stream_ptr()),
          "vpc_nm_loss")


def nmdec_applicable(B, K, d, Ld):
    return bool(lib().vpc_nmdec_applicable(int(B), int(K), int(d), int(Ld)))


def nmdec_tables(model, device):
    """(pack_idx, grad_idx) on the device + image size in floats for the layer-fused decoder kernel (csrc/vpc_nmdec.hip)."""
    import numpy as np
    d, Ld = model.obs_dim, model.latent_dim
    n = 2 * d + model._n_enc + model._n_dec
    pidx, gidx = np.empty(n, np.int32), np.empty(n, np.int32)
    check(lib().vpc_nmdec_build_indices(d, Ld, HID, pidx.ctypes.data_as(L.P), gidx.ctypes.data_as(L.P), n),
          "vpc_nmdec_build_indices")
    # inverse table for the layout-order reduction of the partial blocks: parameter index of a block position, -1 where none
    npart = C.c_long()
    check(lib().vpc_nmdec_layout(1, 8, d, Ld, None, C.byref(npart), None), "vpc_nmdec_layout")
    inv = np.full(npart.value, -1, np.int32)
    own = np.nonzero(gidx >= 0)[0]
    inv[gidx[own]] = own.astype(np.int32)
    # the same for the encoder-backward kernel's partial blocks
    nparte = C.c_long()
    check(lib().vpc_nmenc_build_indices(d, Ld, HID, None, C.byref(nparte), n), "vpc_nmenc_build_indices")
    inve = np.empty(nparte.value, np.int32)
    check(lib().vpc_nmenc_build_indices(d, Ld, HID, inve.ctypes.data_as(L.P), C.byref(nparte), n), "vpc_nmenc_build_indices")
    return tuple(torch.from_numpy(a).to(device) for a in (pidx, gidx, inv, inve))


def nmenc_fwd(img, xin, h1, h2, heads, R, d, Ld):
    check(lib().vpc_nmenc_fwd(ptr(img), ptr(xin), ptr(h1), ptr(h2), ptr(heads), R, d, Ld, stream_ptr()), "vpc_nmenc_fwd")


def nmenc_bwd(img, xin, h1, h2, dht, part, inv, grad, R, d, Ld):
    check(lib().vpc_nmenc_bwd(ptr(img), ptr(xin), ptr(h1), ptr(h2), ptr(dht), ptr(part), part.numel(), ptr(inv), ptr(grad),
                              R, d, Ld, stream_ptr()), "vpc_nmenc_bwd")


def nm_fused_bwd_step(img, x, mask, mask_p, xin, h1, h2, heads, eps, dht, part, stat, part_e, inv, inv_e, grad, out8, loss_f32, accum,
                      B, B_global, K, d, Ld, alpha, params, exp_avg, exp_avg_sq, lr, beta1, beta2, eps_adam, step, pack_idx):
    check(lib().vpc_nm_fused_bwd_step(ptr(img), ptr(x), ptr(mask), ptr(mask_p), ptr(xin), ptr(h1), ptr(h2), ptr(heads), 2 * Ld,
                                      ptr(eps), ptr(dht), ptr(part), ptr(stat), ptr(part_e), part_e.numel(), ptr(inv), ptr(inv_e),
                                      ptr(grad), grad.numel(), ptr(out8), ptr(loss_f32), ptr(accum), B, B_global, K, d, Ld,
                                      float(alpha), ptr(params), ptr(exp_avg), ptr(exp_avg_sq), lr, beta1, beta2, eps_adam,
                                      int(step), ptr(pack_idx), stream_ptr()), "vpc_nm_fused_bwd_step")


def nmdec_step(img, x, mask, mask_p, heads, eps, dht, part, stat, gidx, inv, grad, out8, loss_f32, accum, B, B_global, K, d,
               Ld, alpha, state=None, rng_inc=0):
    check(lib().vpc_nmdec_step(ptr(img), ptr(x), ptr(mask), ptr(mask_p), ptr(heads), 2 * Ld, ptr(eps), ptr(dht),
                               ptr(part), ptr(stat), ptr(gidx), ptr(inv), ptr(grad), grad.numel(), ptr(out8), ptr(loss_f32),
                               ptr(accum), ptr(state), int(rng_inc), B, B_global, K, d, Ld, float(alpha), stream_ptr()),
          "vpc_nmdec_step")


def _f32c(t):
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


def _rows_view(t, B, K, d):
    """[B, K, d] tensor -> (tensor to pass, row pitch) without copying when rows b*K+k are equally spaced."""
    if t.dtype == torch.float32 and t.dim() == 3 and t.stride(2) == 1 and t.stride(0) == K * t.stride(1) \
            and t.stride(1) >= d:
        return t, t.stride(1)
    t = _f32c(t).reshape(B, K, d)
    return t, d


# ------------------------------------------------------------------------------------------------ autograd
class NMEncoderFn(torch.autograd.Function):
    """(x, mask, eps) -> (z [B,K,L], heads [B, mean L | logvar L]).  VAE.py:2378-2391 / :2749-2765."""

    @staticmethod
    def forward(ctx, model, x, mask, eps, K, *weights):
        require_cuda(x, mask, eps, *weights)
        v = model._views()
        d, Ld = model.obs_dim, model.latent_dim
        B, dev = x.shape[0], x.device
        xin = torch.empty(B, d, device=dev)
        nm_mul(x, mask, xin)
        h1 = torch.empty(B, HID, device=dev)
        h2 = torch.empty(B, HID, device=dev)
        heads = torch.empty(B, 2 * Ld, device=dev)
        linear_fwd(xin, v["We1"], v["be1"], h1, B, HID, d, ACT_ELU)
        linear_fwd(h1, v["We2"], v["be2"], h2, B, HID, HID, ACT_ELU)
        linear_fwd(h2, v["Wh"], v["bh"], heads, B, 2 * Ld, HID, ACT_NONE)
        z = torch.empty(B * K, Ld, device=dev)
        nm_sample(heads, eps, z, B, K, Ld)
        ctx.model, ctx.K = model, K
        ctx.save_for_backward(xin, h1, h2, heads, eps if eps is not None else torch.empty(0, device=dev))
        ctx.has_eps = eps is not None
        return z.view(B, K, Ld), heads

    @staticmethod
    def backward(ctx, dz, dheads):
        model, K = ctx.model, ctx.K
        xin, h1, h2, heads, eps = ctx.saved_tensors
        eps = eps if ctx.has_eps else None
        v = model._views()
        d, Ld = model.obs_dim, model.latent_dim
        B, dev = xin.shape[0], xin.device
        dht = torch.empty(B, 2 * Ld, device=dev)
        nm_sample_bwd(_f32c(dz).reshape(B * K, Ld), eps, heads, _f32c(dheads), dht, B, K, Ld)
        g = model._grad_views(torch.empty(model._n_enc, device=dev), "enc")
        dh2 = torch.empty(B, HID, device=dev)
        dh1 = torch.empty(B, HID, device=dev)
        linear_wgrad(dht, h2, g["Wh"], g["bh"], B, 2 * Ld, HID)
        linear_dgrad(dht, v["Wh"], dh2, B, 2 * Ld, HID, x_out=h2, act_prev=ACT_ELU)
        linear_wgrad(dh2, h1, g["We2"], g["be2"], B, HID, HID)
        linear_dgrad(dh2, v["We2"], dh1, B, HID, HID, x_out=h1, act_prev=ACT_ELU)
        linear_wgrad(dh1, xin, g["We1"], g["be1"], B, HID, d)
        return (None, None, None, None, None, g["We1"], g["be1"], g["We2"], g["be2"], g["Wmu"], g["bmu"], g["Wls"],
                g["bls"])


class NMDecoderFn(torch.autograd.Function):
    """z [.., L] -> (x_mean, x_logvar) as the two halves of ONE [M, 2d] buffer.  VAE.py:2393-2397 / :2767-2772."""

    @staticmethod
    def forward(ctx, model, z, *weights):
        require_cuda(z, *weights)
        v = model._views()
        d, Ld = model.obs_dim, model.latent_dim
        lead = z.shape[:-1]
        z2 = _f32c(z).reshape(-1, Ld)
        M, dev = z2.shape[0], z2.device
        g1 = torch.empty(M, HID, device=dev)
        g2 = torch.empty(M, HID, device=dev)
        Y = torch.empty(M, 2 * d, device=dev)
        linear_fwd(z2, v["Wd1"], v["bd1"], g1, M, HID, Ld, ACT_ELU)
        linear_fwd(g1, v["Wd2"], v["bd2"], g2, M, HID, HID, ACT_ELU)
        linear_fwd(g2, v["Wx"], v["bx"], Y, M, 2 * d, HID, ACT_SIGMOID_HARDTANH, d)
        ctx.model, ctx.lead = model, lead
        ctx.save_for_backward(z2, g1, g2, Y)
        Y3 = Y.view(*lead, 2 * d)
        return Y3[..., :d], Y3[..., d:]

    @staticmethod
    def backward(ctx, gxm, gxl):
        model = ctx.model
        z2, g1, g2, Y = ctx.saved_tensors
        v = model._views()
        d, Ld = model.obs_dim, model.latent_dim
        M, dev = z2.shape[0], z2.device
        # the fused loss hands back the two halves of one [M, 2d] buffer: use it in place
        G = None
        if (gxm.dtype == torch.float32 and gxl.dtype == torch.float32 and gxm.dim() >= 2
                and gxm.stride() == gxl.stride() and gxm.stride(-1) == 1 and gxm.stride(-2) == 2 * d
                and gxl.data_ptr() == gxm.data_ptr() + 4 * d
                and all(gxm.stride(i) == gxm.stride(i + 1) * gxm.shape[i + 1] for i in range(gxm.dim() - 2))):
            G = gxm.as_strided((M, 2 * d), (2 * d, 1))
        if G is None:
            G = torch.cat([_f32c(gxm).reshape(M, d), _f32c(gxl).reshape(M, d)], 1)
        g = model._grad_views(torch.empty(model._n_dec, device=dev), "dec")
        dg2 = torch.empty(M, HID, device=dev)
        dg1 = torch.empty(M, HID, device=dev)
        dz = torch.empty(M, Ld, device=dev)
        linear_wgrad(G, g2, g["Wx"], g["bx"], M, 2 * d, HID, y_gate=Y, gate=ACT_SIGMOID_HARDTANH, gate_split=d)
        linear_dgrad(G, v["Wx"], dg2, M, 2 * d, HID, y_gate=Y, gate=ACT_SIGMOID_HARDTANH, gate_split=d, x_out=g2,
                     act_prev=ACT_ELU)
        linear_wgrad(dg2, g1, g["Wd2"], g["bd2"], M, HID, HID)
        linear_dgrad(dg2, v["Wd2"], dg1, M, HID, HID, x_out=g1, act_prev=ACT_ELU)
        linear_wgrad(dg1, z2, g["Wd1"], g["bd1"], M, HID, Ld)
        linear_dgrad(dg1, v["Wd1"], dz, M, HID, Ld)
        return (None, dz.view(*ctx.lead, Ld), g["Wd1"], g["bd1"], g["Wd2"], g["bd2"], g["Wxm"], g["bxm"], g["Wxl"],
                g["bxl"])


class NMLossFn(torch.autograd.Function):
    """Fused importance-weighted loss + every gradient.  Returns (loss fp32, out8 fp64, xm_imp or empty)."""

    @staticmethod
    def forward(ctx, cfg, x, mask, mask_p, xm_q, xl_q, heads_q, xm_p, xl_p, heads_p, W, b, eps_kl):
        reg = mask_p is not None
        require_cuda(x, mask, mask_p, xm_q, xl_q, heads_q, xm_p, xl_p, heads_p, W, b, eps_kl)
        B, K, d, Ld = cfg["B"], cfg["K"], cfg["d"], cfg["L"]
        dev = x.device
        xm_q, ldq = _rows_view(xm_q, B, K, d)
        xl_q, ldq2 = _rows_view(xl_q, B, K, d)
        if ldq2 != ldq:
            xm_q, xl_q = xm_q.contiguous(), xl_q.contiguous()
            ldq = d
        ldp = d
        if reg:
            xm_p, ldp = _rows_view(xm_p, B, K, d)
            xl_p, ldp2 = _rows_view(xl_p, B, K, d)
            if ldp2 != ldp:
                xm_p, xl_p = xm_p.contiguous(), xl_p.contiguous()
                ldp = d
        heads_q = _f32c(heads_q)
        heads_p = _f32c(heads_p) if reg else None
        need_grad = cfg["grad"] and any(ctx.needs_input_grad)
        Gq = Gp = ghq = ghp = gW = gb = None
        if need_grad:
            Gq = torch.empty(B * K, 2 * d, device=dev)
            ghq = torch.empty(B, 2 * Ld, device=dev)
            gW = torch.empty(d, device=dev)
            gb = torch.empty(d, device=dev)
            if reg:
                Gp = torch.empty(B * K, 2 * d, device=dev)
                ghp = torch.empty(B, 2 * Ld, device=dev)
        xm_imp = torch.empty(B, d, device=dev) if cfg["impute"] else None
        nbytes = int(lib().vpc_nm_loss_scratch(B, d))
        scratch = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=dev)
        out8 = torch.empty(8, dtype=torch.float64, device=dev)
        half = lambda G: None if G is None else G[:, d:]
        nm_loss(x, mask, mask_p, xm_q, xl_q, ldq, xm_p, xl_p, ldp, heads_q, heads_p, W, b, eps_kl, Gq, half(Gq), Gp,
                half(Gp), 2 * d, ghq, ghp, gW, gb, xm_imp, scratch, out8, None, None, B, cfg.get("B_global", B), K, d,
                Ld, cfg["alpha"])
        ctx.reg, ctx.need_grad, ctx.dims = reg, need_grad, (B, K, d, Ld)
        ctx.wshape = W.shape
        if need_grad:
            ctx.save_for_backward(*[t for t in (Gq, ghq, gW, gb, Gp, ghp) if t is not None])
        loss = out8[0].float()
        imp = xm_imp if xm_imp is not None else torch.empty(0, device=dev)
        ctx.mark_non_differentiable(out8, imp)
        return loss, out8, imp

    @staticmethod
    def backward(ctx, gloss, _g8, _gi):
        if not ctx.need_grad:
            return (None,) * 13
        B, K, d, Ld = ctx.dims
        t = ctx.saved_tensors
        Gq, ghq, gW, gb = t[:4]
        Gq, ghq, gW, gb = Gq * gloss, ghq * gloss, gW * gloss, gb * gloss
        Gq3 = Gq.view(B, K, 2 * d)
        gWs, gbs = gW.view(ctx.wshape), gb.view(ctx.wshape)
        if ctx.reg:
            Gp, ghp = t[4] * gloss, t[5] * gloss
            Gp3 = Gp.view(B, K, 2 * d)
            return (None, None, None, None, Gq3[..., :d], Gq3[..., d:], ghq, Gp3[..., :d], Gp3[..., d:], ghp, gWs, gbs,
                    None)
        return None, None, None, None, Gq3[..., :d], Gq3[..., d:], ghq, None, None, None, gWs, gbs, None


# ------------------------------------------------------------------------------------------------ model classes
class _NMBase(nn.Module):
    regularised = False

    def __init__(self, obs_dim, hid_dim, K, latent_dim, training_parameters, num_samples, num_estimates):
        super().__init__()
        if obs_dim > 256 or latent_dim > 64:
            raise L.VpcError("the MNAR-path kernels support obs_dim <= 256 and latent_dim <= 64")
        self.obs_dim = obs_dim
        self.hid_dim = hid_dim
        self.emb_dim = 10
        self.num_samples = num_samples
        self.num_estimates = num_estimates
        self.latent_dim = latent_dim
        self.batch_size = training_parameters["batch_size"]
        self.K = K
        self.obs_std = 0.1
        self.number_components = 500
        self.training_paramters = training_parameters  # (sic) VAE.py:2341
        d, Ld = obs_dim, latent_dim
        # parameter containers, created in the reference's order (same seed -> same initial weights)
        self.seq_encoder = nn.Sequential(nn.Linear(d, HID), nn.ELU(), nn.Linear(HID, HID), nn.ELU())
        self.q_mu = nn.Sequential(nn.Linear(HID, Ld))
        self.q_logstd = nn.Sequential(nn.Linear(HID, Ld))
        self.seq_decoder = nn.Sequential(nn.Linear(Ld, HID), nn.ELU(), nn.Linear(HID, HID), nn.ELU())
        self.x_mean = nn.Sequential(nn.Linear(HID, d), nn.Sigmoid())
        self.x_logvar = nn.Sequential(nn.Linear(HID, d), nn.Hardtanh(min_val=-10.0, max_val=0))
        emb1 = torch.empty([1, 1, d])
        nn.init.xavier_uniform_(emb1)
        self.W = nn.Parameter(emb1)
        emb2 = torch.empty([1, 1, d])
        nn.init.xavier_uniform_(emb2)
        self.b = nn.Parameter(emb2)
        self.activation = nn.Softplus()
        self._flat = None
        self._n_enc = HID * d + HID + HID * HID + HID + 2 * Ld * HID + 2 * Ld
        self._n_dec = HID * Ld + HID + HID * HID + HID + 2 * d * HID + 2 * d

    # ---- flat parameter buffer: [W b | We1 be1 We2 be2 Wmu Wls bmu bls | Wd1 bd1 Wd2 bd2 Wxm Wxl bxm bxl]
    def _flat_order(self):
        se, sd = self.seq_encoder, self.seq_decoder
        return [("W", self.W), ("b", self.b),
                ("We1", se[0].weight), ("be1", se[0].bias), ("We2", se[2].weight), ("be2", se[2].bias),
                ("Wmu", self.q_mu[0].weight), ("Wls", self.q_logstd[0].weight),
                ("bmu", self.q_mu[0].bias), ("bls", self.q_logstd[0].bias),
                ("Wd1", sd[0].weight), ("bd1", sd[0].bias), ("Wd2", sd[2].weight), ("bd2", sd[2].bias),
                ("Wxm", self.x_mean[0].weight), ("Wxl", self.x_logvar[0].weight),
                ("bxm", self.x_mean[0].bias), ("bxl", self.x_logvar[0].bias)]

    def trainable(self):
        """The 18 trainable tensors in flat-buffer order."""
        return [p for _, p in self._flat_order()]

    def flatten_parameters(self):
        """Make the 18 trainable tensors views of ONE flat fp32 buffer.  Idempotent; call again after .to()."""
        order = self._flat_order()
        flat = self._flat
        ok = flat is not None and flat.device == order[0][1].device
        off = 0
        if ok:
            for _, p in order:
                if p.data.data_ptr() != flat.data_ptr() + 4 * off or not p.data.is_contiguous():
                    ok = False
                    break
                off += p.numel()
        if not ok:
            flat = torch.cat([p.data.detach().reshape(-1).float() for _, p in order]).contiguous()
            off = 0
            for _, p in order:
                p.data = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
            self._flat = flat
            self._view_cache = None
        return self._flat

    def _segment_views(self, buf, which):
        """Named views into a buffer laid out like the 'wb' / 'enc' / 'dec' segment of the flat buffer."""
        d, Ld = self.obs_dim, self.latent_dim
        spec = {"wb": [("W", (d,)), ("b", (d,))],
                "enc": [("We1", (HID, d)), ("be1", (HID,)), ("We2", (HID, HID)), ("be2", (HID,)), ("Wmu", (Ld, HID)),
                        ("Wls", (Ld, HID)), ("bmu", (Ld,)), ("bls", (Ld,))],
                "dec": [("Wd1", (HID, Ld)), ("bd1", (HID,)), ("Wd2", (HID, HID)), ("bd2", (HID,)), ("Wxm", (d, HID)),
                        ("Wxl", (d, HID)), ("bxm", (d,)), ("bxl", (d,))]}[which]
        out, off = {}, 0
        for name, shp in spec:
            n = math.prod(shp)
            out[name] = buf[off:off + n].view(shp)
            off += n
        if which == "enc":
            o = out["Wmu"].storage_offset() - buf.storage_offset()
            out["Wh"] = buf[o:o + 2 * Ld * HID].view(2 * Ld, HID)
            o += 2 * Ld * HID
            out["bh"] = buf[o:o + 2 * Ld]
        elif which == "dec":
            o = out["Wxm"].storage_offset() - buf.storage_offset()
            out["Wx"] = buf[o:o + 2 * d * HID].view(2 * d, HID)
            o += 2 * d * HID
            out["bx"] = buf[o:o + 2 * d]
        return out

    def _views(self):
        # fast path (every training step): the flat buffer and the cached views are current when the first and the last parameter
        # still sit where the flat buffer has them - `.to()` / re-assignment moves all 18, in-place loads keep them (the full
        # check of flatten_parameters walks every parameter: ~12 us of a host-paced 0.19 ms step)
        vc = getattr(self, "_view_cache", None)
        flat = self._flat
        if vc is not None and flat is not None and vc[0] is flat and self.W.data.data_ptr() == flat.data_ptr() and \
                self.x_logvar[0].bias.data.data_ptr() == flat.data_ptr() + 4 * (flat.numel() - self.obs_dim):
            return vc[1]
        flat = self.flatten_parameters()
        L.require_cuda(flat)
        if getattr(self, "_view_cache", None) is None or self._view_cache[0] is not flat:
            d = self.obs_dim
            v = self._segment_views(flat[:2 * d], "wb")
            v.update(self._segment_views(flat[2 * d:2 * d + self._n_enc], "enc"))
            v.update(self._segment_views(flat[2 * d + self._n_enc:], "dec"))
            self._view_cache = (flat, v)
        return self._view_cache[1]

    def _grad_views(self, buf, which):
        return self._segment_views(buf, which)

    def _enc_weights(self):
        se = self.seq_encoder
        return (se[0].weight, se[0].bias, se[2].weight, se[2].bias, self.q_mu[0].weight, self.q_mu[0].bias,
                self.q_logstd[0].weight, self.q_logstd[0].bias)

    def _dec_weights(self):
        sd = self.seq_decoder
        return (sd[0].weight, sd[0].bias, sd[2].weight, sd[2].bias, self.x_mean[0].weight, self.x_mean[0].bias,
                self.x_logvar[0].weight, self.x_logvar[0].bias)

    # ---- reference API
    def _encode(self, x, mask, sample=True, eps=None):
        L.require_cuda(x)
        d, Ld, K = self.obs_dim, self.latent_dim, self.num_samples
        xf = _f32c(x.reshape(-1, d))
        mf = _f32c(mask.reshape(-1, d).to(x.device))
        B = xf.shape[0]
        if sample and eps is None:
            eps = torch.randn(B, K, Ld, device=xf.device)  # Normal(mean, std).rsample(), VAE.py:2387 / :2761
        z, heads = NMEncoderFn.apply(self, xf, mf, _f32c(eps) if sample else None, K, *self._enc_weights())
        mean = heads[:, :Ld].unsqueeze(1).expand(B, K, Ld)
        log_var = heads[:, Ld:].unsqueeze(1).expand(B, K, Ld)
        mean._vpc_heads = heads
        log_var._vpc_heads = heads
        return z, mean, log_var

    def encoder(self, x, mask, sample=True):
        """VAE.py:2378-2391 / :2749-2765 -> (z, mean, log_var), each [B, num_samples, latent_dim]."""
        return self._encode(x, mask, sample)

    def decoder(self, z_int):
        """VAE.py:2393-2397 / :2767-2772 -> (x_mean, x_logvar)."""
        L.require_cuda(z_int)
        return NMDecoderFn.apply(self, z_int, *self._dec_weights())

    @staticmethod
    def _heads_of(mean, logvar):
        h = getattr(mean, "_vpc_heads", None)
        if h is not None and h is getattr(logvar, "_vpc_heads", None):
            return h
        return torch.cat([mean[:, 0, :], logvar[:, 0, :]], 1)  # any [B,K,L] pair replicated over K

    def _loss(self, x, mask, mask_p, outs_q, outs_p, alpha, eps_kl, llh_eval):
        d, Ld, K = self.obs_dim, self.latent_dim, self.num_samples
        xf = _f32c(x.reshape(-1, d))
        B = xf.shape[0]
        cfg = dict(B=B, K=K, d=d, L=Ld, alpha=alpha, grad=torch.is_grad_enabled(), impute=bool(llh_eval))
        xm_q, xl_q, mean_q, logvar_q = outs_q
        hq = self._heads_of(mean_q, logvar_q)
        mf = _f32c(mask.reshape(-1, d).to(xf.device))
        if outs_p is not None:
            xm_p, xl_p, mean_p, logvar_p = outs_p
            hp = self._heads_of(mean_p, logvar_p)
            mpf = _f32c(mask_p.reshape(-1, d).to(xf.device))
        else:
            xm_p = xl_p = hp = mpf = None
        loss, out8, imp = NMLossFn.apply(cfg, xf, mf, mpf, xm_q, xl_q, hq, xm_p, xl_p, hp, self.W, self.b, eps_kl)
        if llh_eval:  # VAE.py:2458-2461 / :2810-2813
            return imp, loss, out8[5].float()
        return loss, loss  # (print_loss, train_loss)


class REG_notMIWAE_v2(_NMBase):
    """Posterior-consistency regularised not-MIWAE.  Reference: src/models/VAE.py:2327-2505."""
    regularised = True

    def __init__(self, obs_dim, hid_dim, K, latent_dim, training_parameters, num_samples, num_estimates):
        super().__init__(obs_dim, hid_dim, K, latent_dim, training_parameters, num_samples, num_estimates)
        self.logits = nn.Sequential(nn.Linear(obs_dim, obs_dim)).double()  # VAE.py:2371: unused, but in state_dict
        self.max_epoch = 2800

    def forward(self, data, mask, mask_p, stage="train"):
        # VAE.py:2500-2505: q pass first (RNG order), p outputs returned first
        z_q, mean_q, logvar_q = self.encoder(data, mask)
        x_mean_q, x_logvar_q = self.decoder(z_q)
        z_p, mean_p, logvar_p = self.encoder(data, mask_p)
        x_mean_p, x_logvar_p = self.decoder(z_p)
        return mean_p, logvar_p, x_mean_p, x_logvar_p, mean_q, logvar_q, x_mean_q, x_logvar_q

    def loss(self, x, x_recon_p, x_logvar_p, mean_p, logvar_p, x_recon_q, x_logvar_q, mean_q, logvar_q, mask, mask_p,
             epoch, vae_elbo=False, llh_eval=False, MI=False, beta_annealing=False, beta=1.0, alpha=1.0,
             alpha_annealing=False, stage="train", missing_process="selfmasking_known"):
        """VAE.py:2398-2471.  epoch / beta / annealing flags are accepted and, as in the reference, unused."""
        if missing_process != "selfmasking_known":
            raise NotImplementedError("only the reference's default missing_process='selfmasking_known' is accelerated")
        if MI:
            raise NotImplementedError("the MI branch of the reference reads undefined names (VAE.py:2463-2467)")
        return self._loss(x, mask, mask_p, (x_recon_q, x_logvar_q, mean_q, logvar_q),
                          (x_recon_p, x_logvar_p, mean_p, logvar_p), alpha, None, llh_eval)


class notMIWAE_myversion(_NMBase):
    """not-MIWAE with a Monte-Carlo KL.  Reference: src/models/VAE.py:2691-2847."""

    def forward(self, data, mask, stage="train"):
        z, mean, logvar = self.encoder(data, mask)
        x_mean, x_logvar = self.decoder(z)
        return mean, logvar, x_mean, x_logvar

    def loss(self, x, x_recon, x_logvar, mean, logvar, epoch, mask, vae_elbo=False, llh_eval=False, MI=False,
             beta_annealing=False, beta=1.0, stage="train", missing_process="selfmasking_known", eps_kl=None):
        """VAE.py:2774-2823; draws the fresh z of :2791-2793 on the device unless eps_kl [B,K,L] is given."""
        if missing_process != "selfmasking_known":
            raise NotImplementedError("only the reference's default missing_process='selfmasking_known' is accelerated")
        if MI:
            raise NotImplementedError("the MI branch of the reference reads undefined names (VAE.py:2815-2819)")
        B = x.reshape(-1, self.obs_dim).shape[0]
        if eps_kl is None:
            eps_kl = torch.randn(B, self.num_samples, self.latent_dim, device=x.device)
        return self._loss(x, mask, None, (x_recon, x_logvar, mean, logvar), None, 0.0, _f32c(eps_kl), llh_eval)


# ------------------------------------------------------------------------------------------------ fused step
class NMTrainer:
    """The whole training step of the MNAR path (train.py:28-117 for 'reg_notMIWAE*' / 'vanilla_notMIWAE*') as a
    fixed sequence of HIP launches, no host synchronisation: float mask_p draw + stacked encoder input, Philox
    normals, encoder / decoder GEMM chains with the q and p passes STACKED along the batch (one GEMM per layer for
    both passes), the fused loss kernel, the backward GEMM chain writing straight into one flat gradient buffer,
    one all-reduce of [grads | loss] under data parallelism, flat Adam."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, seed=0, process_group=None, world_size=1, rank=0,
                 precision="f32"):
        """precision: "f32", "bf16x3" or "bf16" for the 17 GEMMs of the step (csrc/vpc_bf16.h; operands stay fp32 in
        memory and are converted in registers); the loss kernel and Adam are fp32 in every mode."""
        if not isinstance(model, _NMBase):
            raise TypeError("NMTrainer supports REG_notMIWAE_v2 and notMIWAE_myversion")
        self.model, self.reg = model, model.regularised
        self.lr, self.betas, self.eps_adam = lr, betas, eps
        self.seed, self.rng_offset, self.step_count = seed, 0, 0
        self.pg, self.world_size, self.rank = process_group, world_size, rank
        from .ops import PRECISIONS
        if precision not in PRECISIONS:
            raise ValueError(f"precision {precision!r}: expected one of {sorted(PRECISIONS)}")
        self.precision, self.prec = precision, PRECISIONS[precision]
        flat = model.flatten_parameters()
        L.require_cuda(flat)
        self.dev = flat.device
        n = flat.numel()
        self.bucket = torch.zeros(n + 1, device=self.dev)  # [grads | loss] -> ONE all-reduce per step
        self.grad, self.loss = self.bucket[:n], self.bucket[n:]
        self.exp_avg = torch.zeros(n, device=self.dev)
        self.exp_avg_sq = torch.zeros(n, device=self.dev)
        self.accum = torch.zeros(1, device=self.dev)
        self.out8 = torch.zeros(8, dtype=torch.float64, device=self.dev)
        d = model.obs_dim
        self.g = model._segment_views(self.grad[:2 * d], "wb")
        self.g.update(model._segment_views(self.grad[2 * d:2 * d + model._n_enc], "enc"))
        self.g.update(model._segment_views(self.grad[2 * d + model._n_enc:], "dec"))
        off = 0
        for p in model.trainable():
            p.grad = self.grad[off:off + p.numel()].view_as(p)
            off += p.numel()
        self._B = None
        self.timers = None
        # the bf16 image of the layer-fused path is current when nothing has written the parameters since the launch that packed it
        # (the fused tail of a single-device step re-packs what its Adam updates): the flat buffer's address + the parameters'
        # version counters at that time.  Writes torch does not count (p.data.copy_, raw kernels): call invalidate_image().
        self._plist = model.trainable()
        self._img_key = None

    def invalidate_image(self):
        """Force a re-pack of the bf16 weight image at the next step (after writing parameters behind torch's version counters)."""
        self._img_key = None

    def _ws(self, B):
        if self._B == B:
            return
        m, dev = self.model, self.dev
        d, Ld, K = m.obs_dim, m.latent_dim, m.num_samples
        P = 2 if self.reg else 1
        R, M = P * B, P * B * K
        e = lambda *s: torch.empty(*s, device=dev)
        self.xin, self.mask_p = e(R, d), e(B, d)
        self.h1, self.h2, self.heads = e(R, HID), e(R, HID), e(R, 2 * Ld)
        self.eps = e(2, B, K, Ld)  # reg: eps_q, eps_p; vanilla: eps (sampling), eps_kl (MC KL)
        # layer-fused decoder (plain bf16, obs_dim 128, both model classes): nothing of B * K rows is materialised
        self.use_nmdec = bool(self.prec == 2 and nmdec_applicable(B, K, d, Ld))
        Mg = 0 if self.use_nmdec else M  # rows of the GEMM chain's decoder-side workspaces
        self.z, self.g1, self.g2, self.Y = e(Mg, Ld), e(Mg, HID), e(Mg, HID), e(Mg, 2 * d)
        self.G, self.gheads, self.dht = e(Mg, 2 * d), e(R, 2 * Ld), e(R, 2 * Ld)
        self.dg2, self.dg1, self.dz = e(Mg, HID), e(Mg, HID), e(Mg, Ld)
        Rg = 0 if self.use_nmdec else R     # (the fused path's encoder backward is one kernel too: nmenc_bwd)
        self.dh2, self.dh1 = e(Rg, HID), e(Rg, HID)
        # per-layer partial buffers of the six weight gradients (summed by ONE launch at the end of the backward pass)
        self.wg_shapes = [(M, 2 * d, HID), (M, HID, HID), (M, HID, Ld), (R, 2 * Ld, HID), (R, HID, HID), (R, HID, d)]
        sizes = [0 if self.use_nmdec else int(lib().vpc_linear_wgrad_scratch(*sh))
                 for i, sh in enumerate(self.wg_shapes)]
        buf = e(sum(sizes))
        self._wg_cache = {}
        self.wg_scratch, o = [], 0
        for n in sizes:
            self.wg_scratch.append(buf[o:o + n])
            o += n
        nbytes = int(lib().vpc_nm_loss_scratch(B, d))
        self.scratch = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=dev)
        # the slices the step passes to its launches, made once per batch size (a tensor view costs 1-3 us of host time, and the
        # eager step at the reference's batch 128 is paced by the host)
        BK = B * K
        Y, G = self.Y, self.G
        self._sl = dict(Yl=Y[:, d:], Yp=Y[BK:] if self.reg else None, Ypl=Y[BK:, d:] if self.reg else None, Gl=G[:, d:],
                        Gp=G[BK:] if self.reg else None, Gpl=G[BK:, d:] if self.reg else None,
                        hp=self.heads[B:] if self.reg else None, ghp=self.gheads[B:] if self.reg else None,
                        eps0=self.eps[0], eps1=self.eps[1])
        if self.use_nmdec:  # image, partial blocks, index tables
            nimg, npart, nblk = C.c_int(), C.c_long(), C.c_int()
            check(lib().vpc_nmdec_layout(B, K, d, Ld, C.byref(nimg), C.byref(npart), C.byref(nblk)), "vpc_nmdec_layout")
            if getattr(self, "_nd_tables", None) is None:
                self._nd_tables = nmdec_tables(m, dev)
                self.nd_img = torch.zeros(nimg.value, device=dev)
            self.nd_part = e(nblk.value * npart.value)
            self.ne_part = e(nblk.value * int(self._nd_tables[3].numel()))  # the encoder-backward kernel's blocks (fused tail)
            self.nd_stat = torch.empty(nblk.value * 5, dtype=torch.float64, device=dev)
        self._B = B

    def _t(self, name, fn, *a, **kw):
        if self.timers is None:
            return fn(*a, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **kw)
        e1.record()
        self.timers.setdefault(name, []).append((e0, e1))
        return r

    def step(self, x, mask, mask_p=None, eps=None, *, alpha=1.0, p_missingness=30, global_batch=None, row_lo=None,
             _state=None):
        """One optimiser step.  mask_p / eps ([2, B, K, L]: (eps_q, eps_p) or (eps, eps_kl)) may be injected for
        parity tests; otherwise they are drawn on the device (one launch: mask_p + stacked encoder input + normals).
        Data parallel: this rank holds rows [row_lo, row_lo + B) (default rank * B) of `global_batch` rows; the Philox
        counters are those of the global row, so the draws do not depend on the world size."""
        m = self.model
        v = m._views()
        d, Ld, K = m.obs_dim, m.latent_dim, m.num_samples
        xf, mf = _f32c(x.reshape(-1, d)), _f32c(mask.reshape(-1, d))
        L.require_cuda(xf, mf)
        B = xf.shape[0]
        Bg = global_batch or B * self.world_size  # every rank normalises by the GLOBAL batch: SUM over ranks = result
        self._ws(B)
        reg = self.reg
        P = 2 if reg else 1
        R, M, BK = P * B, P * B * K, B * K
        t = self._t
        if row_lo is None:
            row_lo = self.rank * B if self.world_size > 1 else 0
        rng_inc = (2 * Bg * K * Ld + 3) // 4 + (Bg * d + 3) // 4 + 1  # what the GLOBAL batch consumes
        sharded = (row_lo * d) % 4 == 0 and (K * Ld) % 4 == 0
        # shapes whose shards do not start on a Philox group: per-rank counter streams instead of global-row counters
        rank_off = 0 if sharded else (self.rank << 44)
        eps_shard = (B, Bg, row_lo, K * Ld) if sharded else None
        # ---- inputs
        if reg and mask_p is not None:
            mp = _f32c(mask_p.reshape(-1, d))
            nm_mul(xf, mf, self.xin[:B])
            nm_mul(xf, mp, self.xin[B:])
            if eps is None:
                from .ops import fill_normal
                fill_normal(self.eps, self.seed, self.rng_offset + (1 << 40) + rank_off, _state, eps_shard)
        else:
            mp = self.mask_p if reg else None
            t("prep", nm_prep, xf, mf, mp, self.xin, B, d, 1.0 - p_missingness / 100.0, self.seed,
              self.rng_offset + rank_off, self.eps if eps is None else None, self.rng_offset + (1 << 40) + rank_off,
              _state, row_lo * d if sharded else 0, eps_shard)
        if eps is not None:
            self.eps.copy_(eps)
        self.rng_offset += rng_inc
        # ---- forward
        # single device, eager: everything behind the encoder forward is three launches (decoder tiles, encoder-backward tiles, one tail
        # that reduces both sets of blocks, applies Adam and re-packs the image: vpc_nm_fused_bwd_step) and the pack launch goes
        fuse_tail = self.use_nmdec and self.world_size == 1 and _state is None and self.timers is None
        if self.use_nmdec:
            # plain bf16 at obs_dim 128: ONE image (decoder, missingness model, encoder) packed by one launch, the encoder forward
            # as one kernel (csrc/vpc_nmdec.hip: nmenc_fwd_kernel) instead of three GEMM launches
            key = (m._flat.data_ptr(), tuple(p._version for p in self._plist)) if fuse_tail else None
            if key is None or key != self._img_key:
                from .ops import step_pack_weights_bf16
                t("pack", step_pack_weights_bf16, m._flat, self._nd_tables[0], self.nd_img)
            self._img_key = None  # (whoever updates the parameters below says whether the image followed)
            t("enc_fwd", nmenc_fwd, self.nd_img, self.xin, self.h1, self.h2, self.heads, R, d, Ld)
        else:
            t("enc_fwd", linear_fwd, self.xin, v["We1"], v["be1"], self.h1, R, HID, d, ACT_ELU, precision=self.prec)
            t("enc_fwd", linear_fwd, self.h1, v["We2"], v["be2"], self.h2, R, HID, HID, ACT_ELU, precision=self.prec)
            t("enc_fwd", linear_fwd, self.h2, v["Wh"], v["bh"], self.heads, R, 2 * Ld, HID, ACT_NONE, precision=self.prec)
        g = self.g
        # weight gradients: partials per layer, all summed by one launch after the last one (6 reduction launches less;
        # the timer mode keeps the per-layer form so that every entry brackets a complete gradient)
        defer = self.timers is None
        pend = []

        def wgrad(name, i, dy, xx, dw, db):
            Mi, Ni, Ki = self.wg_shapes[i]
            if not defer:
                return t(name, linear_wgrad, dy, xx, dw, db, Mi, Ni, Ki, precision=self.prec)
            linear_wgrad(dy, xx, None, None, Mi, Ni, Ki, precision=self.prec, scratch=self.wg_scratch[i])
            pend.append((self.wg_scratch[i], Mi, Ni, Ki, dw, db, False))

        if self.use_nmdec:
            # K-fold rsample, decoder, loss, decoder backward and the K-fold sum of dz in ONE kernel (csrc/vpc_nmdec.hip): the
            # decoder / missingness-model gradients land in self.grad, d loss / d heads in self.dht
            pidx, gidx, ginv, einv = self._nd_tables
            if fuse_tail:
                self.step_count += 1
                nm_fused_bwd_step(self.nd_img, xf, mf, mp, self.xin, self.h1, self.h2, self.heads, self.eps, self.dht, self.nd_part,
                                  self.nd_stat, self.ne_part, ginv, einv, self.grad, self.out8, self.loss, self.accum, B, Bg, K, d,
                                  Ld, alpha, m._flat, self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0], self.betas[1],
                                  self.eps_adam, self.step_count, pidx)
                self._img_key = key
                return
            t("dec_fused", nmdec_step, self.nd_img, xf, mf, mp, self.heads, self.eps, self.dht, self.nd_part, self.nd_stat,
              gidx, ginv, self.grad, self.out8, self.loss, self.accum if self.world_size == 1 else None, B, Bg, K, d, Ld, alpha,
              _state, rng_inc)
            # the encoder's backward in one kernel + its reduction (nmenc_bwd_kernel), through the decoder's partial-block buffer
            t("enc_bwd", nmenc_bwd, self.nd_img, self.xin, self.h1, self.h2, self.dht, self.nd_part, einv, self.grad, R, d, Ld)
        else:
            self._step_decoder_gemms(xf, mf, mp, B, Bg, alpha, _state, rng_inc, t, v, wgrad)
            self._step_encoder_bwd_gemms(R, t, v, wgrad)
        if pend:
            wgrad_reduce(pend, self._wg_cache)  # (buffers and gradient views are fixed for a batch size: arrays built once)
        self._step_tail(t, _state)

    def _step_encoder_bwd_gemms(self, R, t, v, wgrad):
        m, g = self.model, self.g
        d, Ld = m.obs_dim, m.latent_dim
        wgrad("enc_bwd", 3, self.dht, self.h2, g["Wh"], g["bh"])
        t("enc_bwd", linear_dgrad, self.dht, v["Wh"], self.dh2, R, 2 * Ld, HID, x_out=self.h2, act_prev=ACT_ELU, precision=self.prec)
        wgrad("enc_bwd", 4, self.dh2, self.h1, g["We2"], g["be2"])
        t("enc_bwd", linear_dgrad, self.dh2, v["We2"], self.dh1, R, HID, HID, x_out=self.h1, act_prev=ACT_ELU, precision=self.prec)
        wgrad("enc_bwd", 5, self.dh1, self.xin, g["We1"], g["be1"])

    def _step_tail(self, t, _state):
        m = self.model
        if self.world_size > 1:  # ONE collective per step: RCCL on the compute stream, or torch.distributed (dist.py)
            from . import dist as dp_mod
            if not getattr(self, "_coll_ready", False):
                self.collective = dp_mod.make_collective(self.world_size, self.rank, self.dev, self.pg)
                self._coll_ready = True
            dp_mod.allreduce_bucket(self.bucket, self.pg, self.collective)
        self.step_count += 1
        from .ops import adam_step
        t("adam", adam_step, m._flat, self.grad, self.exp_avg, self.exp_avg_sq, self.step_count, self.lr,
          self.betas[0], self.betas[1], self.eps_adam, None, None, None if _state is None else _state[0:1],
          loss_in=self.loss if self.world_size > 1 else None, accum=self.accum if self.world_size > 1 else None)

    def _step_decoder_gemms(self, xf, mf, mp, B, Bg, alpha, _state, rng_inc, t, v, wgrad):
        """Decoder forward, loss and decoder backward as the GEMM chain (every precision, both model classes)."""
        m, reg = self.model, self.reg
        d, Ld, K = m.obs_dim, m.latent_dim, m.num_samples
        P = 2 if reg else 1
        R, M, BK = P * B, P * B * K, B * K
        sl = self._sl
        t("sample", nm_sample, self.heads, self.eps if reg else sl["eps0"], self.z, R, K, Ld)
        t("dec_fwd1", linear_fwd, self.z, v["Wd1"], v["bd1"], self.g1, M, HID, Ld, ACT_ELU, precision=self.prec)
        t("dec_fwd2", linear_fwd, self.g1, v["Wd2"], v["bd2"], self.g2, M, HID, HID, ACT_ELU, precision=self.prec)
        t("dec_fwd3", linear_fwd, self.g2, v["Wx"], v["bx"], self.Y, M, 2 * d, HID, ACT_SIGMOID_HARDTANH, d, precision=self.prec)
        # ---- loss + output-side gradients
        Y, G = self.Y, self.G
        t("loss", nm_loss, xf, mf, mp, Y, sl["Yl"], 2 * d, sl["Yp"], sl["Ypl"], 2 * d,
          self.heads, sl["hp"], v["W"], v["b"], None if reg else sl["eps1"], G, sl["Gl"],
          sl["Gp"], sl["Gpl"], 2 * d, self.gheads, sl["ghp"],
          self.g["W"], self.g["b"], None, self.scratch, self.out8, self.loss,
          self.accum if self.world_size == 1 else None,  # data parallel: the epoch total takes the ALL-REDUCED loss (below)
          B, Bg, K, d, Ld, alpha, _state, rng_inc, True)
        # ---- backward (G already holds the head pre-activation gradients: no gate pass over Y)
        g = self.g
        wgrad("dec_wgrad3", 0, G, self.g2, g["Wx"], g["bx"])
        t("dec_dgrad3", linear_dgrad, G, v["Wx"], self.dg2, M, 2 * d, HID, x_out=self.g2, act_prev=ACT_ELU, precision=self.prec)
        wgrad("dec_wgrad2", 1, self.dg2, self.g1, g["Wd2"], g["bd2"])
        t("dec_dgrad2", linear_dgrad, self.dg2, v["Wd2"], self.dg1, M, HID, HID, x_out=self.g1, act_prev=ACT_ELU, precision=self.prec)
        wgrad("dec_wgrad1", 2, self.dg1, self.z, g["Wd1"], g["bd1"])
        t("dec_dgrad1", linear_dgrad, self.dg1, v["Wd1"], self.dz, M, HID, Ld, precision=self.prec)
        t("sample_bwd", nm_sample_bwd, self.dz, self.eps if reg else sl["eps0"], self.heads, self.gheads, self.dht, R,
          K, Ld)

    def step_graph(self, x, mask, *, alpha=1.0, p_missingness=30):
        """The same step replayed from a captured HIP graph (torch.cuda.CUDAGraph): ONE host call instead of ~30
        launches.  Step count and Philox offsets live on the device (`state`), bumped by the loss kernel's finalize.
        The first call with a new (shape, alpha, p_missingness) runs one eager step and captures.  Measured on
        MI355X this removes the host cost but not the ~10 us dependent-dispatch latency per kernel node, so at
        B = 128 it is no faster than step() (profiles/r01_notes.md).  Single process only."""
        if self.world_size > 1:
            return self.step(x, mask, alpha=alpha, p_missingness=p_missingness)
        d = self.model.obs_dim
        xf, mf = _f32c(x.reshape(-1, d)), _f32c(mask.reshape(-1, d))
        L.require_cuda(xf, mf)
        key = (tuple(xf.shape), float(alpha), p_missingness)
        if getattr(self, "_graph_key", None) != key:
            self.step(xf, mf, alpha=alpha, p_missingness=p_missingness)  # eager warm-up (LDS attributes, workspaces)
            self._gx, self._gm = xf.clone(), mf.clone()
            self.state = torch.tensor([self.step_count, 0], dtype=torch.int64, device=self.dev)
            timers, self.timers = self.timers, None
            base_rng, base_step = self.rng_offset, self.step_count
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.step(self._gx, self._gm, alpha=alpha, p_missingness=p_missingness, _state=self.state)
            self._graph_rng_inc = self.rng_offset - base_rng
            self.rng_offset, self.step_count = base_rng, base_step  # capture executed nothing
            self.timers = timers
            self._graph, self._graph_key = g, key
            return
        if xf.data_ptr() != self._gx.data_ptr():
            self._gx.copy_(xf)
        if mf.data_ptr() != self._gm.data_ptr():
            self._gm.copy_(mf)
        self._graph.replay()
        self.step_count += 1
        self.rng_offset += self._graph_rng_inc

    def loss_value(self) -> float:
        return float(self.loss.item())

    def epoch_total(self, reset=True) -> float:
        v = float(self.accum.item())
        if reset:
            self.accum.zero_()
        return v
