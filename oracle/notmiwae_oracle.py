"""CPU oracle for the MNAR path: REG_notMIWAE_v2 / notMIWAE_myversion.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``, ``__graft_entry__.smoke()`` and bench scripts'
cpu-baseline legs may import it; the product path never routes through ``oracle/``.

Two independent restatements:

* ``NMTorchPort`` - stock-PyTorch (CPU, fp32, autograd) restatement of the reference classes
  ``REG_notMIWAE_v2`` (src/models/VAE.py:2327-2505) and ``notMIWAE_myversion`` (src/models/VAE.py:2691-2847):
  encoder d->128->128 (ELU) with two heads 128->L, K-fold replicated reparameterised draw, decoder L->128->128
  (ELU) with heads 128->d (+Sigmoid) and 128->d (+Hardtanh(-10,0)), self-masking missingness model
  ``logits = -softplus(W)(x_mixed - b)``, importance-weighted bound with the reference's sign convention
  (logsumexp of +l_w, quirk 15 of SURVEY Appendix B).
* ``loss_closed_form`` - float64 numpy closed form of the loss AND its hand-derived gradients with respect to the
  network outputs and (W, b) - the maths the fused HIP loss kernel implements.

Parity pinning: ``tests/test_notmiwae_oracle.py`` checks both against ``tests/golden/nm_*.npz``, vectors captured
by importing the reference itself (``tests/golden/make_golden_notmiwae.py``).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

HID = 128  # VAE.py:2343-2363 hard-codes 128 regardless of hid_dim
HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)

# state_dict order of the reference classes (own Parameters first, then children in registration order)
NM_KEYS = (
    "W", "b",
    "seq_encoder.0.weight", "seq_encoder.0.bias", "seq_encoder.2.weight", "seq_encoder.2.bias",
    "q_mu.0.weight", "q_mu.0.bias", "q_logstd.0.weight", "q_logstd.0.bias",
    "seq_decoder.0.weight", "seq_decoder.0.bias", "seq_decoder.2.weight", "seq_decoder.2.bias",
    "x_mean.0.weight", "x_mean.0.bias", "x_logvar.0.weight", "x_logvar.0.bias",
)


def nm_param_shapes(d: int, L: int):
    return {
        "W": (1, 1, d), "b": (1, 1, d),
        "seq_encoder.0.weight": (HID, d), "seq_encoder.0.bias": (HID,),
        "seq_encoder.2.weight": (HID, HID), "seq_encoder.2.bias": (HID,),
        "q_mu.0.weight": (L, HID), "q_mu.0.bias": (L,),
        "q_logstd.0.weight": (L, HID), "q_logstd.0.bias": (L,),
        "seq_decoder.0.weight": (HID, L), "seq_decoder.0.bias": (HID,),
        "seq_decoder.2.weight": (HID, HID), "seq_decoder.2.bias": (HID,),
        "x_mean.0.weight": (d, HID), "x_mean.0.bias": (d,),
        "x_logvar.0.weight": (d, HID), "x_logvar.0.bias": (d,),
    }


def nm_init_params(d: int, L: int, seed: int = 0) -> Dict[str, torch.Tensor]:
    """nn.Linear default init (U(+-1/sqrt(fan_in))) and xavier_uniform on the [1,1,d] W / b (VAE.py:2365-2370)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, shp in nm_param_shapes(d, L).items():
        if k in ("W", "b"):
            # xavier_uniform_ on [1, 1, d]: fan_in = 1 * d, fan_out = 1 * d -> bound = sqrt(6 / (2 d))
            bound = math.sqrt(6.0 / (2.0 * d))
        else:
            fan_in = shp[1] if len(shp) == 2 else nm_param_shapes(d, L)[k.replace("bias", "weight")][1]
            bound = 1.0 / math.sqrt(fan_in)
        out[k] = (torch.rand(shp, generator=g) * 2 - 1) * bound
    return out


def _bf16_t(t):
    """tensor -> fp32 -> bf16 (round to nearest even) -> the tensor's dtype."""
    return t.to(torch.float32).to(torch.bfloat16).to(t.dtype)


def _split_t(t):
    v = t.to(torch.float32)
    hi = v.to(torch.bfloat16).to(torch.float32)
    lo = (v - hi).to(torch.bfloat16).to(torch.float32)
    return hi.to(t.dtype), lo.to(t.dtype)


def _mm_mode(a, b, mode):
    if mode == "bf16":
        return _bf16_t(a) @ _bf16_t(b)
    ah, al = _split_t(a)
    bh, bl = _split_t(b)
    return ah @ bh + ah @ bl + al @ bh


class _RoundedLinear(torch.autograd.Function):
    """nn.Linear as the generic GEMM kernels compute it with precision bf16 / bf16x3 (csrc/vpc_gemm.hip: fp32 tiles in LDS,
    fragments converted in registers): both operands of EVERY product - forward x W^T, dgrad dY W, wgrad dY^T x - are
    rounded to bf16 (or split hi + lo with the lo * lo term dropped), the bias is added and db is summed in fp32 from the
    unrounded dY.  Run it on float64 tensors: the accumulation is then exact and only the operand rounding is modelled."""

    @staticmethod
    def forward(ctx, x, w, b, mode, db_rounded=False):
        ctx.mode, ctx.shape, ctx.db_rounded = mode, x.shape, db_rounded
        x2 = x.reshape(-1, x.shape[-1])
        ctx.save_for_backward(x2, w)
        return (_mm_mode(x2, w.t(), mode) + b).reshape(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        dx = _mm_mode(dy2, w, ctx.mode).reshape(ctx.shape)
        # db_rounded (the layer-fused decoder kernel, csrc/vpc_nmdec.hip): db is the column sum of the STAGED bf16 dY
        db = (_bf16_t(dy2) if ctx.db_rounded else dy2).sum(0)
        return dx, _mm_mode(dy2.t(), x2, ctx.mode), db, None, None


def rounded_linear(mode, db_rounded=False):
    """F.linear replacement for NMTorchPort(linear=...): the bf16-emulating oracle of the MNAR step."""
    return lambda x, w, b: _RoundedLinear.apply(x, w, b, mode, db_rounded)


class _EluRoundedGate(torch.autograd.Function):
    """ELU whose derivative is taken from the bf16-rounded OUTPUT (1 where it is > 0, else output + 1): the layer-fused
    decoder kernel keeps its hidden activations only as packed bf16 MFMA operands (csrc/vpc_nmdec.hip, elu_gate)."""

    @staticmethod
    def forward(ctx, x):
        y = F.elu(x)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        yr = _bf16_t(ctx.saved_tensors[0])
        return dy * torch.where(yr > 0, torch.ones_like(yr), yr + 1)


class NMTorchPort:
    """Functional restatement over a dict of tensors (keys NM_KEYS).  ``regularised`` selects REG_notMIWAE_v2.
    ``linear``: the affine layer (default F.linear; `rounded_linear("bf16")` models the bf16 GEMM kernels).
    ``fused_decoder``: model the rounding points of the layer-fused kernels (csrc/vpc_nmdec.hip).  Decoder side: bias gradients
    from the bf16-rounded dY, ELU' from the bf16-rounded activation (the missingness model's dW / db are fp32 row sums there as
    everywhere).  Encoder side (nmenc_fwd / nmenc_bwd kernels): the bf16 GEMM rounding points, bias gradients from the
    bf16-rounded dY, ELU' from the fp32 activation (plain F.elu)."""

    def __init__(self, params: Dict[str, torch.Tensor], L: int, K: int, regularised: bool, linear=None, fused_decoder=False):
        self.p = params
        self.L, self.K, self.reg = L, K, regularised
        self.lin = linear or F.linear
        self.fused_decoder = fused_decoder
        self.lin_dec = rounded_linear("bf16", db_rounded=True) if fused_decoder else self.lin
        self.elu_dec = _EluRoundedGate.apply if fused_decoder else F.elu
        self.lin_enc = self.lin_dec if fused_decoder else self.lin

    # VAE.py:2378-2391 / :2749-2765
    def encoder(self, x, mask, eps: Optional[torch.Tensor]):
        p = self.p
        dt = p["seq_encoder.0.weight"].dtype  # fp32 in the reference (x.float()); float64 only in the oracle self-check
        lin = self.lin_enc
        h = F.elu(lin(x.to(dt) * mask.to(dt), p["seq_encoder.0.weight"], p["seq_encoder.0.bias"]))
        h = F.elu(lin(h, p["seq_encoder.2.weight"], p["seq_encoder.2.bias"]))
        mean = lin(h, p["q_mu.0.weight"], p["q_mu.0.bias"])
        logvar = lin(h, p["q_logstd.0.weight"], p["q_logstd.0.bias"])
        mean = mean.unsqueeze(1).expand(-1, self.K, -1)
        logvar = logvar.unsqueeze(1).expand(-1, self.K, -1)
        z = mean if eps is None else mean + eps * torch.exp(logvar / 2)
        return z, mean, logvar

    # VAE.py:2393-2397 / :2767-2772
    def decoder(self, z):
        p = self.p
        lin, elu = self.lin_dec, self.elu_dec
        g = elu(lin(z, p["seq_decoder.0.weight"], p["seq_decoder.0.bias"]))
        g = elu(lin(g, p["seq_decoder.2.weight"], p["seq_decoder.2.bias"]))
        xm = torch.sigmoid(lin(g, p["x_mean.0.weight"], p["x_mean.0.bias"]))
        xl = F.hardtanh(lin(g, p["x_logvar.0.weight"], p["x_logvar.0.bias"]), -10.0, 0.0)
        return xm, xl

    @staticmethod
    def _nll(t, mean, logvar):  # neg_gaussian_log_likelihood VAE.py:2488-2490 (sum over features)
        return torch.sum(HALF_LOG_2PI + 0.5 * logvar + (t - mean) ** 2 / (2 * torch.exp(logvar)), 2)

    def _logp_s(self, x3, m3, xm):  # VAE.py:2413-2432 'selfmasking_known'
        mixed = xm * (1 - m3) + x3 * m3
        logits = -F.softplus(self.p["W"]) * (mixed - self.p["b"])
        return torch.sum(m3 * logits - F.softplus(logits), 2)  # Bernoulli(logits).log_prob(m)

    def reg_forward(self, x, mask, mask_p, eps_q, eps_p):  # VAE.py:2500-2505 (p outputs first)
        zq, mq, lq = self.encoder(x, mask, eps_q)
        xmq, xlq = self.decoder(zq)
        zp, mp, lp = self.encoder(x, mask_p, eps_p)
        xmp, xlp = self.decoder(zp)
        return mp, lp, xmp, xlp, mq, lq, xmq, xlq

    def reg_loss(self, x, outs, mask, mask_p, alpha=1.0, llh_eval=False):  # VAE.py:2398-2471
        mp, lp, xmp, xlp, mq, lq, xmq, xlq = outs
        K = self.K
        x3, m3, p3 = (t.unsqueeze(1).expand(-1, K, -1) for t in (x, mask, mask_p))
        RE_q = self._nll(x3 * m3, xmq * m3, xlq * m3)
        RE_p = self._nll(x3 * p3, xmp * p3, xlp * p3)
        KL_q = torch.sum(0.5 * (torch.exp(lq) + mq ** 2 - 1 - lq), 2)
        KL_p = torch.sum(0.5 * (torch.exp(lp) + mp ** 2 - 1 - lp), 2)
        l_w_q = RE_q + KL_q - self._logp_s(x3, m3, xmq)
        l_w_p = RE_p + KL_p
        loss_q = torch.mean(torch.logsumexp(l_w_q, 1) - math.log(float(K)))
        loss_p = torch.mean(torch.logsumexp(l_w_p, 1) - math.log(float(K)))
        kl_el = 0.5 * (torch.exp(lq - lp) + (mq - mp) ** 2 / torch.exp(lp) - 1 - (lq - lp))
        e3 = m3 * (1 - p3)
        nll_e = self._nll(x3 * e3, xmq * e3, xlq * e3).mean()
        loss = loss_q + alpha * (kl_el.mean() - loss_q + loss_p + nll_e)
        if llh_eval:
            wl = torch.softmax(-l_w_q, 1)
            return torch.sum(xmq * wl.unsqueeze(2), 1), loss, RE_q.mean()
        return loss

    def van_forward(self, x, mask, eps):  # VAE.py:2844-2847
        z, m, l = self.encoder(x, mask, eps)
        xm, xl = self.decoder(z)
        return m, l, xm, xl

    def van_loss(self, x, outs, mask, eps_kl, llh_eval=False):  # VAE.py:2774-2823 (MC KL with a fresh draw)
        m, l, xm, xl = outs
        K = self.K
        x3, m3 = (t.unsqueeze(1).expand(-1, K, -1) for t in (x, mask))
        RE = self._nll(x3 * m3, xm * m3, xl * m3)
        sd = torch.exp(l / 2)
        z = m + eps_kl * sd
        log_q = torch.sum(-0.5 * ((z - m) / sd) ** 2 - torch.log(sd) - HALF_LOG_2PI, 2)
        log_p = torch.sum(-0.5 * z ** 2 - HALF_LOG_2PI, 2)
        l_w = RE + (log_q - log_p) - self._logp_s(x3, m3, xm)
        loss = torch.mean(torch.logsumexp(l_w, 1) - math.log(float(K)))
        if llh_eval:
            wl = torch.softmax(-l_w, 1)
            return torch.sum(xm * wl.unsqueeze(2), 1), loss, RE.mean()
        return loss


# ----------------------------------------------------------------------------------------------------------------
# float64 closed form of the loss + gradients w.r.t. network outputs (what the fused HIP loss kernel computes)
# ----------------------------------------------------------------------------------------------------------------
def _softplus(a):
    return np.logaddexp(0.0, a)


def _sigmoid(a):
    return 1.0 / (1.0 + np.exp(-a))


def _pass_terms(x, m, xm, xl, W=None, b=None):
    """Per (row, sample): RE and -log p(s|x); also the per-element derivative pieces."""
    x3, m3 = x[:, None, :], m[:, None, :]
    iv = np.exp(-xl)
    r = x3 - xm
    RE = np.sum(HALF_LOG_2PI + m3 * (0.5 * xl + 0.5 * r * r * iv), 2)
    dRE_dxm = -m3 * r * iv
    dRE_dxl = m3 * (0.5 - 0.5 * r * r * iv)
    if W is None:
        return RE, dRE_dxm, dRE_dxl, None
    sp = _softplus(W)[None, None, :]
    mixed = xm * (1 - m3) + x3 * m3
    logits = -sp * (mixed - b[None, None, :])
    nlogp = -np.sum(m3 * logits - _softplus(logits), 2)
    dn_dlogit = _sigmoid(logits) - m3
    extra = dict(nlogp=nlogp, d_xm=dn_dlogit * (-sp) * (1 - m3),
                 d_W=dn_dlogit * (-_sigmoid(W)[None, None, :]) * (mixed - b[None, None, :]),
                 d_b=dn_dlogit * sp)
    return RE, dRE_dxm, dRE_dxl, extra


def _lse_weights(l_w):
    mx = l_w.max(1, keepdims=True)
    e = np.exp(l_w - mx)
    s = e.sum(1, keepdims=True)
    return (mx + np.log(s))[:, 0], e / s


def loss_closed_form(x, mask, outs_q, W, b, K, mask_p=None, outs_p=None, alpha=1.0, eps_kl=None):
    """Returns (loss, grads) with grads keyed d_xm_q, d_xl_q, d_mean_q, d_logvar_q, d_W, d_b (+ *_p).

    outs_* = (mean [B,L], logvar [B,L], x_mean [B,K,d], x_logvar [B,K,d]) in float64.
    Regularised form (mask_p / outs_p given): VAE.py:2398-2447.  Vanilla form (eps_kl [B,K,L] given): VAE.py:2774-2805.
    """
    x, m = x.astype(np.float64), mask.astype(np.float64)
    mq, lq, xmq, xlq = (np.asarray(t, np.float64) for t in outs_q)
    W, b = W.reshape(-1).astype(np.float64), b.reshape(-1).astype(np.float64)
    B, L = mq.shape
    RE, dxm, dxl, ex = _pass_terms(x, m, xmq, xlq, W, b)
    g = {}
    if outs_p is None:
        sd = np.exp(lq / 2)[:, None, :]
        z = mq[:, None, :] + eps_kl * sd
        KL = np.sum(-0.5 * eps_kl ** 2 - 0.5 * lq[:, None, :] + 0.5 * z * z, 2)
        lse, w = _lse_weights(RE + KL + ex["nlogp"])
        loss = np.mean(lse - math.log(K))
        om = w / B
        g["d_xm_q"] = om[:, :, None] * (dxm + ex["d_xm"])
        g["d_xl_q"] = om[:, :, None] * dxl
        g["d_mean_q"] = np.sum(om[:, :, None] * z, 1)
        g["d_logvar_q"] = np.sum(om[:, :, None] * (-0.5 + 0.5 * z * eps_kl * sd), 1)
        g["d_W"] = np.sum(om[:, :, None] * ex["d_W"], (0, 1))
        g["d_b"] = np.sum(om[:, :, None] * ex["d_b"], (0, 1))
        return loss, g
    p = mask_p.astype(np.float64)
    mp, lp, xmp, xlp = (np.asarray(t, np.float64) for t in outs_p)
    KL_q = np.sum(0.5 * (np.exp(lq) + mq * mq - 1 - lq), 1)
    KL_p = np.sum(0.5 * (np.exp(lp) + mp * mp - 1 - lp), 1)
    lse_q, wq = _lse_weights(RE + KL_q[:, None] + ex["nlogp"])
    REp, dxm_p, dxl_p, _ = _pass_terms(x, p, xmp, xlp)
    lse_p, wp = _lse_weights(REp + KL_p[:, None])
    e = m * (1 - p)
    REe, dxm_e, dxl_e, _ = _pass_terms(x, e, xmq, xlq)
    kl_el = 0.5 * (np.exp(lq - lp) + (mq - mp) ** 2 * np.exp(-lp) - 1 - (lq - lp))
    loss_q, loss_p = np.mean(lse_q - math.log(K)), np.mean(lse_p - math.log(K))
    loss = loss_q + alpha * (kl_el.mean() - loss_q + loss_p + REe.mean())
    oq, op, oe = (1 - alpha) * wq / B, alpha * wp / B, alpha / (B * K)
    g["d_xm_q"] = oq[:, :, None] * (dxm + ex["d_xm"]) + oe * dxm_e
    g["d_xl_q"] = oq[:, :, None] * dxl + oe * dxl_e
    g["d_xm_p"] = op[:, :, None] * dxm_p
    g["d_xl_p"] = op[:, :, None] * dxl_p
    cr = alpha / (B * L)
    iv = np.exp(-lp)
    g["d_mean_q"] = (1 - alpha) / B * mq + cr * (mq - mp) * iv
    g["d_logvar_q"] = (1 - alpha) / B * 0.5 * (np.exp(lq) - 1) + cr * 0.5 * (np.exp(lq - lp) - 1)
    g["d_mean_p"] = alpha / B * mp - cr * (mq - mp) * iv
    g["d_logvar_p"] = alpha / B * 0.5 * (np.exp(lp) - 1) + cr * 0.5 * (-np.exp(lq - lp) - (mq - mp) ** 2 * iv + 1)
    g["d_W"] = np.sum(oq[:, :, None] * ex["d_W"], (0, 1))
    g["d_b"] = np.sum(oq[:, :, None] * ex["d_b"], (0, 1))
    return loss, g


def adam_reference(params, grads, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam (no weight decay / amsgrad), float64 numpy, one step; returns new (params, m, v)."""
    out_p, out_m, out_v = {}, {}, {}
    for k in params:
        g = grads[k].astype(np.float64)
        out_m[k] = b1 * m[k] + (1 - b1) * g
        out_v[k] = b2 * v[k] + (1 - b2) * g * g
        mh = out_m[k] / (1 - b1 ** step)
        vh = out_v[k] / (1 - b2 ** step)
        out_p[k] = params[k] - lr * mh / (np.sqrt(vh) + eps)
    return out_p, out_m, out_v
