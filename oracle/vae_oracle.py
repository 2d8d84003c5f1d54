"""CPU oracle for the VAE posterior-consistency training step.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The product path (``vae-posterior-consistency_amd``) never routes through
anything in ``oracle/``; it fails loudly when the HIP extension is missing.

Two independent restatements of the reference algorithm live here:

* ``TorchPort`` - an op-for-op stock-PyTorch (CPU, fp32, eager,
  ``torch.distributions``) restatement of the reference's model classes
  ``Reg_VAE`` (src/models/VAE.py:350-507) and ``vanilla_VAE``
  (src/models/VAE.py:1119-1240) plus the step body of
  src/experiment_main/train.py:28-117.  It executes the same torch op sequence
  as the reference (un-fused), so it doubles as the "reference CPU path" that
  ``bench.py`` times as ``cpu_baseline`` (kind = "port").
* ``closed_form_*`` - a float64 numpy closed form of the same maths with
  hand-derived gradients.  It shares no code with the
  torch port, so agreement of the two pins both.

Parity pinning: both are checked in ``tests/test_oracle_golden.py`` against
``tests/golden/*.npz``, vectors produced by importing the reference itself in
the authoring container (``tests/golden/make_golden.py``; the reference never
travels to the GPU box).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np
import torch
from torch.distributions import Normal, kl_divergence

HID1 = 100  # src/models/VAE.py:366-376 hard-codes 100 / 50 regardless of hid_dim
HID2 = 50
MAX_EPOCH = 2800  # src/models/VAE.py:384
X_LOGVAR = math.log((0.1 * math.sqrt(2.0)) ** 2)  # src/models/VAE.py:379 -> log 0.02
HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)

PARAM_KEYS = (
    "seq_encoder.0.weight", "seq_encoder.0.bias",
    "seq_encoder.2.weight", "seq_encoder.2.bias",
    "seq_encoder.4.weight", "seq_encoder.4.bias",
    "seq_decoder.0.weight", "seq_decoder.0.bias",
    "seq_decoder.2.weight", "seq_decoder.2.bias",
    "seq_decoder.4.weight", "seq_decoder.4.bias",
)


def param_shapes(obs_dim: int, latent_dim: int, mask_augm: bool = False) -> Dict[str, Tuple[int, ...]]:
    """Shapes of the 12 trainable tensors (src/models/VAE.py:366-376; first layer 2d wide for the *_mask
    classes, VAE.py:527-533)."""
    d, L = obs_dim, latent_dim
    return {
        "seq_encoder.0.weight": (HID1, 2 * d if mask_augm else d), "seq_encoder.0.bias": (HID1,),
        "seq_encoder.2.weight": (HID2, HID1), "seq_encoder.2.bias": (HID2,),
        "seq_encoder.4.weight": (2 * L, HID2), "seq_encoder.4.bias": (2 * L,),
        "seq_decoder.0.weight": (HID2, L), "seq_decoder.0.bias": (HID2,),
        "seq_decoder.2.weight": (HID1, HID2), "seq_decoder.2.bias": (HID1,),
        "seq_decoder.4.weight": (d, HID1), "seq_decoder.4.bias": (d,),
    }


def init_params(obs_dim: int, latent_dim: int, seed: int = 0, mask_augm: bool = False) -> Dict[str, torch.Tensor]:
    """nn.Linear default init (Kaiming-uniform a=sqrt(5) == U(-1/sqrt(in), 1/sqrt(in)))."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, shp in param_shapes(obs_dim, latent_dim, mask_augm).items():
        layer = k.rsplit(".", 1)[0]
        fan_in = param_shapes(obs_dim, latent_dim, mask_augm)[layer + ".weight"][1]
        bound = 1.0 / math.sqrt(fan_in)
        out[k] = (torch.rand(shp, generator=g) * 2 - 1) * bound
    return out


# --------------------------------------------------------------------------
# torch port (op-for-op; this is the timed "reference CPU path")
# --------------------------------------------------------------------------
class TorchPort:
    """Functional restatement of Reg_VAE / vanilla_VAE on stock PyTorch CPU."""

    def __init__(self, params: Dict[str, torch.Tensor], latent_dim: int, reg_type: str = "kl_reg",
                 mask_augm: bool = False):
        self.p = params
        self.latent_dim = latent_dim
        self.reg_type = reg_type
        self.mask_augm = mask_augm  # Reg_VAE_mask / vanilla_VAE_mask (VAE.py:510-667, 995-1116)
        # VAE.py:379 - shape (1,) CPU tensor
        self.x_logvar = torch.log(torch.square(torch.tensor([0.1 * math.sqrt(2.0)])))
        # VAE.py:381-383
        self.prior = Normal(torch.zeros(latent_dim), torch.ones(latent_dim))

    # -- VAE.py:387-395 / 1155-1163
    def encoder(self, x, mask, eps: Optional[torch.Tensor] = None, sample: bool = True):
        p = self.p
        if self.mask_augm:  # VAE.py:545-548
            h = torch.stack([x.float() * mask, mask * 1.0], 1).reshape(-1, 2 * x.shape[1])
        else:
            h = x.float() * mask
        h = torch.relu(torch.nn.functional.linear(h, p["seq_encoder.0.weight"], p["seq_encoder.0.bias"]))
        h = torch.relu(torch.nn.functional.linear(h, p["seq_encoder.2.weight"], p["seq_encoder.2.bias"]))
        h = torch.nn.functional.linear(h, p["seq_encoder.4.weight"], p["seq_encoder.4.bias"])
        mean, logvar = h.chunk(2, dim=1)
        if sample:
            std = torch.exp(logvar / 2)
            if eps is None:
                z = Normal(mean, std).rsample()
            else:  # rsample() is loc + eps * scale; injected eps reproduces it bit-for-bit
                z = mean + eps * std
        else:
            z = mean
        return z, mean, logvar

    # -- VAE.py:397-401
    def decoder(self, z):
        p = self.p
        g = torch.relu(torch.nn.functional.linear(z, p["seq_decoder.0.weight"], p["seq_decoder.0.bias"]))
        g = torch.relu(torch.nn.functional.linear(g, p["seq_decoder.2.weight"], p["seq_decoder.2.bias"]))
        g = torch.sigmoid(torch.nn.functional.linear(g, p["seq_decoder.4.weight"], p["seq_decoder.4.bias"]))
        return g, self.x_logvar

    # -- VAE.py:496-507 (q pass first, then p pass; outputs p first)
    def reg_forward(self, data, mask, mask_p, eps_q=None, eps_p=None):
        z_q, mean_q, logvar_q = self.encoder(data, mask, eps_q)
        x_mean_q, x_logvar_q = self.decoder(z_q)
        z_p, mean_p, logvar_p = self.encoder(data, mask_p, eps_p)
        x_mean_p, x_logvar_p = self.decoder(z_p)
        return mean_p, logvar_p, x_mean_p, x_logvar_p, mean_q, logvar_q, x_mean_q, x_logvar_q

    # -- VAE.py:1237-1240
    def vanilla_forward(self, data, mask, eps_q=None):
        z_q, mean_q, logvar_q = self.encoder(data, mask, eps_q)
        x_mean_q, x_logvar_q = self.decoder(z_q)
        return mean_q, logvar_q, x_mean_q, x_logvar_q

    # -- helpers VAE.py:469-494
    @staticmethod
    def _nll(t, mean, log_var):
        return torch.sum(-Normal(mean, torch.exp(log_var / 2.0)).log_prob(t))

    @staticmethod
    def _ll(t, mean, log_var):
        return torch.sum(Normal(mean, torch.exp(log_var / 2.0)).log_prob(t))

    def _kl_std(self, mean, log_var):
        return torch.sum(kl_divergence(Normal(mean, torch.exp(log_var / 2.0)), self.prior))

    @staticmethod
    def _kl_diag(m1, lv1, m2, lv2):
        return torch.sum(kl_divergence(Normal(m1, torch.exp(lv1 / 2)), Normal(m2, torch.exp(lv2 / 2))))

    # -- VAE.py:403-467
    def reg_loss(self, x, x_recon_p, x_logvar_p, mean_p, logvar_p, x_recon_q, x_logvar_q, mean_q, logvar_q,
                 mask, mask_p, epoch, vae_elbo=False, llh_eval=False, MI=False, beta_annealing=False,
                 beta=1.0, alpha=0.8, stage="train", alpha_annealing=True, eps_ml=None):
        x_logvar_q = torch.ones_like(x_recon_q) * x_logvar_q
        x_logvar_p = torch.ones_like(x_recon_p) * x_logvar_p
        bw = (epoch / MAX_EPOCH) * beta if beta_annealing else beta
        if stage == "evaluate":
            RE_q = self._nll(x * mask, x_recon_q * mask, x_logvar_q * mask)
            RE_q_imputed = self._nll(x * ~mask, x_recon_q * ~mask, x_logvar_q * ~mask)
            KL_q = self._kl_std(mean_q, logvar_q)
            loss = RE_q + bw * KL_q
        else:
            RE_q = self._nll(x * mask, x_recon_q * mask, x_logvar_q * mask)
            RE_p = self._nll(x * mask_p, x_recon_p * mask_p, x_logvar_p * mask_p)
            KL_q = self._kl_std(mean_q, logvar_q)
            KL_p = self._kl_std(mean_p, logvar_p)
            loss_q = RE_q + bw * KL_q
            loss_p = RE_p + bw * KL_p
            if self.reg_type == "ml_reg":
                std_q = torch.exp(logvar_q / 2)
                if eps_ml is None:
                    z_q = Normal(mean_q, std_q).rsample()
                else:
                    z_q = mean_q + eps_ml * std_q
                z_ll = self._ll(z_q, mean_p, logvar_p)
                loss = loss_q - (epoch / MAX_EPOCH) * alpha * z_ll
            elif self.reg_type == "kl_reg":
                KL_reg = self._kl_diag(mean_q, logvar_q, mean_p, logvar_p)
                extra = self._nll(x * mask * ~mask_p, x_recon_q * mask * ~mask_p, x_logvar_q * mask * ~mask_p)
                loss = loss_q + alpha * (KL_reg - loss_q + loss_p + extra)
            else:
                raise ValueError("reg_type not implemented")
            RE_q_imputed = 0
        train_loss = loss / x.shape[0]
        if llh_eval:
            return train_loss, train_loss, RE_q / x.shape[0], RE_q_imputed / x.shape[0]
        if MI:
            agg_m, agg_lv = torch.mean(mean_q, 0), torch.mean(logvar_q, 0)
            KL_agg = self._kl_std(agg_m, agg_lv)
            return train_loss, train_loss, KL_q / x.shape[0] - KL_agg, KL_q / x.shape[0]
        return train_loss, train_loss

    # -- VAE.py:1171-1208
    def vanilla_loss(self, x, x_recon_q, x_logvar_q, mean_q, logvar_q, epoch, mask, vae_elbo=False,
                     llh_eval=False, MI=False, beta_annealing=False, beta=1.0, alpha=0.8,
                     alpha_annealing=True, stage="train"):
        x_logvar_q = torch.ones_like(x_recon_q) * x_logvar_q
        RE_q = self._nll(x * mask, x_recon_q * mask, x_logvar_q * mask)
        if stage == "evaluate":
            inv = 1 - mask * 1.0
            RE_q_imputed = self._nll(x * inv, x_recon_q * inv, x_logvar_q * inv)
        else:
            RE_q_imputed = 0
        KL_q = self._kl_std(mean_q, logvar_q)
        bw = (epoch / MAX_EPOCH) * beta if beta_annealing else beta
        loss = RE_q + bw * KL_q
        train_loss = loss / x.shape[0]
        if llh_eval:
            return train_loss, train_loss, RE_q / x.shape[0], RE_q_imputed / x.shape[0]
        if MI:
            agg_m, agg_lv = torch.mean(mean_q, 0), torch.mean(logvar_q, 0)
            KL_agg = self._kl_std(agg_m, agg_lv)
            return train_loss, train_loss, KL_q / x.shape[0] - KL_agg, KL_q / x.shape[0]
        return train_loss, train_loss


def create_missing_uci_host(shape, missing_rate, rng: np.random.Generator | None = None):
    """src/utils/utils.py:36-39 - keep-mask, True = keep, P(keep) = 1 - rate/100 (numpy, host)."""
    r = np.random.rand(*shape) if rng is None else rng.random(shape)
    return torch.from_numpy(r < (1 - missing_rate / 100))


def make_leaf_params(params: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}


def torch_reg_step(params, latent_dim, x, mask, mask_p, eps_q, eps_p, *, reg_type="kl_reg", alpha=1.0,
                   beta=1.0, beta_annealing=False, epoch=1, eps_ml=None, mask_augm=False):
    """forward + loss + backward of one Reg_VAE step (train.py:87-115). Returns (loss, grads, outs)."""
    leaf = make_leaf_params(params)
    port = TorchPort(leaf, latent_dim, reg_type, mask_augm)
    outs = port.reg_forward(x, mask, mask_p, eps_q, eps_p)
    mean_p, logvar_p, x_mean_p, x_logvar_p, mean_q, logvar_q, x_mean_q, x_logvar_q = outs
    _, train_loss = port.reg_loss(x, x_mean_p, x_logvar_p, mean_p, logvar_p, x_mean_q, x_logvar_q, mean_q,
                                  logvar_q, mask, mask_p, epoch, beta_annealing=beta_annealing, beta=beta,
                                  alpha=alpha, eps_ml=eps_ml)
    train_loss.backward()
    grads = {k: v.grad.detach().clone() for k, v in leaf.items()}
    return train_loss.detach(), grads, [o.detach() for o in outs]


def torch_vanilla_step(params, latent_dim, x, mask, eps_q, *, beta=1.0, beta_annealing=False, epoch=1,
                       mask_augm=False):
    leaf = make_leaf_params(params)
    port = TorchPort(leaf, latent_dim, mask_augm=mask_augm)
    mean_q, logvar_q, x_mean_q, x_logvar_q = port.vanilla_forward(x, mask, eps_q)
    _, train_loss = port.vanilla_loss(x, x_mean_q, x_logvar_q, mean_q, logvar_q, epoch, mask,
                                      beta_annealing=beta_annealing, beta=beta)
    train_loss.backward()
    grads = {k: v.grad.detach().clone() for k, v in leaf.items()}
    return train_loss.detach(), grads, [mean_q.detach(), logvar_q.detach(), x_mean_q.detach()]


class TorchTrainer:
    """train.py:17-21,28-117 restated: the model + stock optim.Adam(lr=1e-3) on CPU."""

    def __init__(self, params, latent_dim, reg_type="kl_reg", vanilla=False, lr=1e-3):
        self.leaf = make_leaf_params(params)
        self.port = TorchPort(self.leaf, latent_dim, reg_type)
        self.vanilla = vanilla
        self.opt = torch.optim.Adam(list(self.leaf.values()), lr=lr)

    def step(self, x, mask, mask_p=None, eps_q=None, eps_p=None, *, p_missingness=30, alpha=1.0, beta=1.0,
             beta_annealing=False, epoch=1):
        port = self.port
        if self.vanilla:
            mask_f = mask * torch.ones(x.shape)  # train.py:58,97 (mask_drop = ones -> float mask)
            mean_q, logvar_q, x_mean_q, x_logvar_q = port.vanilla_forward(x, mask_f, eps_q)
            _, train_loss = port.vanilla_loss(x, x_mean_q, x_logvar_q, mean_q, logvar_q, epoch, mask_f,
                                              beta_annealing=beta_annealing, beta=beta)
        else:
            if mask_p is None:  # train.py:53-55
                mask_p = create_missing_uci_host(x.shape, p_missingness) * mask
            outs = port.reg_forward(x, mask, mask_p, eps_q, eps_p)
            mean_p, logvar_p, x_mean_p, x_logvar_p, mean_q, logvar_q, x_mean_q, x_logvar_q = outs
            _, train_loss = port.reg_loss(x, x_mean_p, x_logvar_p, mean_p, logvar_p, x_mean_q, x_logvar_q,
                                          mean_q, logvar_q, mask, mask_p, epoch,
                                          beta_annealing=beta_annealing, beta=beta, alpha=alpha)
        self.opt.zero_grad()
        train_loss.backward()
        self.opt.step()
        return train_loss.item()

    def state(self):
        return {k: v.detach().clone() for k, v in self.leaf.items()}


# --------------------------------------------------------------------------
# float64 numpy closed form, independent of the torch port
# --------------------------------------------------------------------------
def _np(params):
    return {k: np.asarray(v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v, dtype=np.float64)
            for k, v in params.items()}


@dataclass
class PassCache:
    xin: np.ndarray
    h1: np.ndarray
    h2: np.ndarray
    mean: np.ndarray
    logvar: np.ndarray
    eps: np.ndarray
    z: np.ndarray
    g1: np.ndarray
    g2: np.ndarray
    xhat: np.ndarray


# ---- bf16 emulation: the float64 closed form with every GEMM operand rounded exactly where the
# bf16 kernels round it (csrc/vpc_bf16.h).  What the kernels do, restated:
#   * a product's operands - activations, dY and the packed weight image - are bf16 (round to nearest even from the fp32
#     value), accumulation and everything elementwise (bias of layer 1, ReLU, reparameterisation, sigmoid, loss, KL, seeds)
#     stays fp32; "bf16x3" carries every operand as hi = bf16(v), lo = bf16(v - hi) and forms hi*hi + hi*lo + lo*hi
#   * biases of layers 2..6 live in the weight image (a column that multiplies a constant-1 unit): they are rounded like
#     weights, and their gradients are column sums of the ROUNDED dY (wgrad against the constant 1); b1 is added in fp32 and
#     db1 is the fp32 sum of the unrounded dh1 (enc_bwd_kernel's dbacc)
#   * a value is rounded once and that packed operand feeds the next layer, the dgrad and the wgrad alike
def bf16_round(a):
    """float -> fp32 -> bf16 (round to nearest even, as v_cvt_pk_bf16_f32) -> float64."""
    u = np.ascontiguousarray(np.asarray(a, np.float32)).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32).astype(np.float64).reshape(np.shape(a))


def bf16_split(a):
    """hi = bf16(v), lo = bf16(v - hi) with v the fp32 value (pk_bf16 / pk_bf16_lo in vpc_bf16.h)."""
    v = np.asarray(a, np.float32)
    hi = bf16_round(v)
    lo = bf16_round(v.astype(np.float64) - hi)  # v - hi is exact in fp32
    return hi, lo


class GemmModel:
    """How a matrix product sees its operands: "f64" exact, "bf16" plain bf16 inputs, "bf16x3" split bf16."""

    def __init__(self, mode="f64", db1_rounded=False):
        if mode in ("f32", None):
            mode = "f64"
        assert mode in ("f64", "bf16", "bf16x3"), mode
        self.mode = mode
        # the whole-step kernel (csrc/vpc_step.hip) takes db1 as a wgrad against a constant-1 operand, like the other
        # biases: the sum of the ROUNDED dh1; the separate encoder-backward kernels sum the unrounded fp32 dh1
        self.db1_rounded = db1_rounded

    def op(self, a):  # the value a product effectively multiplies with
        if self.mode == "f64":
            return np.asarray(a, np.float64)
        if self.mode == "bf16":
            return bf16_round(a)
        hi, lo = bf16_split(a)
        return hi + lo

    def mm(self, a, b):
        if self.mode == "f64":
            return a @ b
        if self.mode == "bf16":
            return bf16_round(a) @ bf16_round(b)
        ah, al = bf16_split(a)
        bh, bl = bf16_split(b)
        return ah @ bh + ah @ bl + al @ bh  # the lo * lo term is dropped (bf_mma)

    def colsum(self, dy):  # bias gradient through the constant-1 column of a wgrad
        return self.op(dy).sum(0)


def closed_form_pass(P, x, m, eps, L, gm=None):
    gm = gm or GemmModel()
    xin = x * m
    h1 = np.maximum(gm.mm(xin, P["seq_encoder.0.weight"].T) + P["seq_encoder.0.bias"], 0)
    h2 = np.maximum(gm.mm(h1, P["seq_encoder.2.weight"].T) + gm.op(P["seq_encoder.2.bias"]), 0)
    o = gm.mm(h2, P["seq_encoder.4.weight"].T) + gm.op(P["seq_encoder.4.bias"])
    mean, logvar = o[:, :L], o[:, L:]
    z = mean + eps * np.exp(logvar / 2)
    g1 = np.maximum(gm.mm(z, P["seq_decoder.0.weight"].T) + gm.op(P["seq_decoder.0.bias"]), 0)
    g2 = np.maximum(gm.mm(g1, P["seq_decoder.2.weight"].T) + gm.op(P["seq_decoder.2.bias"]), 0)
    xhat = 1.0 / (1.0 + np.exp(-(gm.mm(g2, P["seq_decoder.4.weight"].T) + gm.op(P["seq_decoder.4.bias"]))))
    return PassCache(xin, h1, h2, mean, logvar, eps, z, g1, g2, xhat)


def _nll_np(x, xhat, m):
    s2 = math.exp(X_LOGVAR)
    return float(np.sum(HALF_LOG_2PI + m * (0.5 * X_LOGVAR + (x - xhat) ** 2 / (2 * s2))))


def _kl0_np(mean, lv):
    return float(np.sum(0.5 * (np.exp(lv) + mean ** 2 - 1.0 - lv)))


def _klr_np(mq, lq, mp, lp):
    return float(np.sum(0.5 * (np.exp(lq - lp) + (mq - mp) ** 2 * np.exp(-lp) - 1.0 - (lq - lp))))


def _pass_backward(P, c: PassCache, dxhat, dmean, dlogvar, grads, gm=None):
    """Backprop one encoder/decoder pass given seeds on xhat, mean, logvar; accumulates into grads."""
    gm = gm or GemmModel()
    dpre3 = dxhat * c.xhat * (1 - c.xhat)
    grads["seq_decoder.4.weight"] += gm.mm(dpre3.T, c.g2)
    grads["seq_decoder.4.bias"] += gm.colsum(dpre3)
    dg2 = gm.mm(dpre3, P["seq_decoder.4.weight"]) * (c.g2 > 0)
    grads["seq_decoder.2.weight"] += gm.mm(dg2.T, c.g1)
    grads["seq_decoder.2.bias"] += gm.colsum(dg2)
    dg1 = gm.mm(dg2, P["seq_decoder.2.weight"]) * (c.g1 > 0)
    grads["seq_decoder.0.weight"] += gm.mm(dg1.T, c.z)
    grads["seq_decoder.0.bias"] += gm.colsum(dg1)
    dz = gm.mm(dg1, P["seq_decoder.0.weight"])
    dm = dmean + dz
    dl = dlogvar + dz * c.eps * 0.5 * np.exp(c.logvar / 2)
    do = np.concatenate([dm, dl], axis=1)
    grads["seq_encoder.4.weight"] += gm.mm(do.T, c.h2)
    grads["seq_encoder.4.bias"] += gm.colsum(do)
    dh2 = gm.mm(do, P["seq_encoder.4.weight"]) * (c.h2 > 0)
    grads["seq_encoder.2.weight"] += gm.mm(dh2.T, c.h1)
    grads["seq_encoder.2.bias"] += gm.colsum(dh2)
    dh1 = gm.mm(dh2, P["seq_encoder.2.weight"]) * (c.h1 > 0)
    grads["seq_encoder.0.weight"] += gm.mm(dh1.T, c.xin)
    # explicit bias vector of layer 1: fp32 sum of the unrounded dh1, or (whole-step kernel) a wgrad against ones
    grads["seq_encoder.0.bias"] += gm.colsum(dh1) if gm.db1_rounded else dh1.sum(0)


def closed_form_reg_step(params, L, x, mask, mask_p, eps_q, eps_p, *, alpha=1.0, beta=1.0,
                         beta_annealing=False, epoch=1, reg_type="kl_reg", eps_ml=None, gemm="f64", db1_rounded=False):
    """float64 loss + analytic grads of the Reg_VAE training loss (train stage). Appendix A.
    gemm = "bf16" / "bf16x3": matrix products with operands rounded where the bf16 kernels round them (GemmModel)."""
    gm = GemmModel(gemm, db1_rounded)
    P = _np(params)
    x = np.asarray(x, np.float64)
    M = np.asarray(mask, np.float64)
    Pm = np.asarray(mask_p, np.float64)
    E = M * (1 - Pm)
    eq = np.asarray(eps_q, np.float64)
    ep = np.asarray(eps_p, np.float64)
    B = x.shape[0]
    bw = (epoch / MAX_EPOCH) * beta if beta_annealing else beta
    s2 = math.exp(X_LOGVAR)
    cq = closed_form_pass(P, x, M, eq, L, gm)
    cp = closed_form_pass(P, x, Pm, ep, L, gm)
    RE_q, RE_p = _nll_np(x, cq.xhat, M), _nll_np(x, cp.xhat, Pm)
    KL_q, KL_p = _kl0_np(cq.mean, cq.logvar), _kl0_np(cp.mean, cp.logvar)
    loss_q, loss_p = RE_q + bw * KL_q, RE_p + bw * KL_p
    grads = {k: np.zeros_like(v) for k, v in P.items()}
    if reg_type == "kl_reg":
        KLr = _klr_np(cq.mean, cq.logvar, cp.mean, cp.logvar)
        loss = loss_q + alpha * (KLr - loss_q + loss_p + _nll_np(x, cq.xhat, E))
        dxq = ((1 - alpha) * M + alpha * E) * (cq.xhat - x) / s2
        dxp = alpha * Pm * (cp.xhat - x) / s2
        diff = cq.mean - cp.mean
        eip = np.exp(-cp.logvar)
        r = np.exp(cq.logvar - cp.logvar)
        dmq = (1 - alpha) * bw * cq.mean + alpha * diff * eip
        dlq = (1 - alpha) * bw * 0.5 * (np.exp(cq.logvar) - 1) + alpha * 0.5 * (r - 1)
        dmp = alpha * bw * cp.mean - alpha * diff * eip
        dlp = alpha * bw * 0.5 * (np.exp(cp.logvar) - 1) + alpha * 0.5 * (1 - r - diff ** 2 * eip)
    elif reg_type == "ml_reg":
        e3 = np.asarray(eps_ml, np.float64)
        w = (epoch / MAX_EPOCH) * alpha
        sq = np.exp(cq.logvar / 2)
        zq = cq.mean + e3 * sq
        eip = np.exp(-cp.logvar)
        dlt = zq - cp.mean
        zll = float(np.sum(-HALF_LOG_2PI - 0.5 * cp.logvar - dlt ** 2 * eip / 2))
        loss = loss_q - w * zll
        dxq = M * (cq.xhat - x) / s2
        dxp = np.zeros_like(x)
        dmq = bw * cq.mean + w * dlt * eip
        dlq = bw * 0.5 * (np.exp(cq.logvar) - 1) + w * dlt * eip * e3 * 0.5 * sq
        dmp = -w * dlt * eip
        dlp = w * (0.5 - 0.5 * dlt ** 2 * eip)
    else:
        raise ValueError(reg_type)
    _pass_backward(P, cq, dxq / B, dmq / B, dlq / B, grads, gm)
    _pass_backward(P, cp, dxp / B, dmp / B, dlp / B, grads, gm)
    terms = dict(RE_q=RE_q, RE_p=RE_p, KL_q=KL_q, KL_p=KL_p)
    return loss / B, grads, (cq, cp), terms


def closed_form_vanilla_step(params, L, x, mask, eps_q, *, beta=1.0, beta_annealing=False, epoch=1, gemm="f64",
                             db1_rounded=False):
    gm = GemmModel(gemm, db1_rounded)
    P = _np(params)
    x = np.asarray(x, np.float64)
    M = np.asarray(mask, np.float64)
    B = x.shape[0]
    bw = (epoch / MAX_EPOCH) * beta if beta_annealing else beta
    s2 = math.exp(X_LOGVAR)
    cq = closed_form_pass(P, x, M, np.asarray(eps_q, np.float64), L, gm)
    loss = _nll_np(x, cq.xhat, M) + bw * _kl0_np(cq.mean, cq.logvar)
    grads = {k: np.zeros_like(v) for k, v in P.items()}
    _pass_backward(P, cq, M * (cq.xhat - x) / s2 / B, bw * cq.mean / B,
                   bw * 0.5 * (np.exp(cq.logvar) - 1) / B, grads, gm)
    return loss / B, grads, cq


def closed_form_eval(params, L, x, mask, eps_q, *, beta=1.0, beta_annealing=False, epoch=1):
    """evaluate stage (VAE.py:410-420): loss = loss_q, RE_q, RE_q_imputed on ~mask."""
    P = _np(params)
    x = np.asarray(x, np.float64)
    M = np.asarray(mask, np.float64)
    B = x.shape[0]
    bw = (epoch / MAX_EPOCH) * beta if beta_annealing else beta
    cq = closed_form_pass(P, x, M, np.asarray(eps_q, np.float64), L)
    RE_q, RE_i = _nll_np(x, cq.xhat, M), _nll_np(x, cq.xhat, 1 - M)
    return (RE_q + bw * _kl0_np(cq.mean, cq.logvar)) / B, RE_q / B, RE_i / B


def adam_reference(params, grads_seq, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """Plain float64 Adam over a sequence of grad dicts (torch.optim.Adam semantics, train.py:21)."""
    P = {k: np.array(v, np.float64) for k, v in params.items()}
    m = {k: np.zeros_like(v) for k, v in P.items()}
    v_ = {k: np.zeros_like(v) for k, v in P.items()}
    for t, g in enumerate(grads_seq, start=1):
        for k in P:
            m[k] = b1 * m[k] + (1 - b1) * g[k]
            v_[k] = b2 * v_[k] + (1 - b2) * g[k] ** 2
            mh = m[k] / (1 - b1 ** t)
            vh = v_[k] / (1 - b2 ** t)
            P[k] = P[k] - lr * mh / (np.sqrt(vh) + eps)
    return P


# --------------------------------------------------------------------------
# active variable selection reward (config 5): src/experiment_main/evaluate.py:514-634
# --------------------------------------------------------------------------
def chain_kl(mean, logvar, mean_i, logvar_i):
    """evaluate.py:582-583 / 631-632: note the first term divides by v = exp(logvar/2) (the std), not the
    variance - a quirk of the reference that is reproduced as is."""
    var, v, var_i = torch.exp(logvar), torch.exp(logvar / 2), torch.exp(logvar_i)
    return 0.5 * torch.sum(torch.square(mean_i - mean) / v + var_i / var - 1.0 - logvar_i + logvar, 1)


def chaini_I(port: "TorchPort", x, mask, i):
    """evaluate.py:546-586 (the z drawn by encoder(sample=True) is discarded, so sample=False is equivalent)."""
    tm = mask.clone()
    _, mean, logvar = port.encoder(x, tm, sample=False)
    tm[:, i] = 1
    _, mean_i, logvar_i = port.encoder(x, tm, sample=False)
    return chain_kl(mean, logvar, mean_i, logvar_i)


def chaini_II(port: "TorchPort", x, mask, i):
    """evaluate.py:590-634: as chaini_I with the target (last column) revealed in both encodings."""
    tm = mask.clone()
    tm[:, -1] = 1
    _, mean, logvar = port.encoder(x, tm, sample=False)
    tm[:, i] = 1
    _, mean_i, logvar_i = port.encoder(x, tm, sample=False)
    return chain_kl(mean, logvar, mean_i, logvar_i)


def R_lindley_chain(port: "TorchPort", i, x, mask, M, im, loc):
    """evaluate.py:514-542.  `temp_x[loc, -1]` is not reset between MC samples, so from the second sample on
    chaini_I sees the previous sample's imputed target when the target column is observed - kept as is."""
    im_i, im_target = im[:, :, i], im[:, :, -1]
    temp_x = x.clone()
    approx = 0
    for m in range(M):
        temp_x[loc, i] = im_i[m, loc].float()
        approx = approx + chaini_I(port, temp_x[loc, :], mask[loc, :], i)
        temp_x[loc, -1] = im_target[m, loc].float()
        approx = approx - chaini_II(port, temp_x[loc, :], mask[loc, :], i)
    return approx / M


def reward_matrix(port: "TorchPort", x, mask, M, im):
    """The candidate loop of active_learning_func (evaluate.py:424-433): R[n, u] for every unobserved feature u
    of every row, -1e4 elsewhere."""
    n, d = x.shape
    R = -1e4 * torch.ones(n, d - 1)
    for u in range(d - 1):
        loc = np.where(mask[:, u].numpy() == 0)[0]
        if len(loc):
            R[loc, u] = R_lindley_chain(port, u, x, mask, M, im, loc).float()
    return R


def active_learning_loop(x, M, forward_xmean, reward_fn, max_steps=None):
    """CPU restatement of the acquisition loop of active_learning_func (src/experiment_main/evaluate.py:352-456, one
    repeat, the non-flow branch): mask starts all-unobserved; per step M forward passes give the MC imputations `im`, the
    reward matrix R (evaluate.py:424-433 -> R_lindley_chain) picks argmax per row, the chosen feature is revealed and the
    target MSE of M further forward passes is recorded.  `forward_xmean(mask) -> x_mean_q [n, d]` is one model.forward
    (the reference draws its eps from the global RNG there, so callers replay recorded outputs or bring their own model);
    `reward_fn(x, mask, im) -> R [n, d-1]`.  Returns dict(info_curve [d], action [n, d-1], R_hist [d-1, n, d-1],
    im [d-1, M, n, d])."""
    n, d = x.shape
    mask = torch.zeros(n, d)
    mse = lambda xm: torch.nn.functional.mse_loss(xm[:, -1], x[:, -1])  # noqa: E731  (evaluate.py:390)
    curve = [torch.stack([mse(forward_xmean(mask)) for _ in range(M)]).mean()]
    actions, R_hist, ims = [], [], []
    for t in range(d - 1 if max_steps is None else min(max_steps, d - 1)):
        im = torch.stack([forward_xmean(mask) for _ in range(M)], 0)
        R = reward_fn(x, mask, im)
        i_opt = R.argmax(1)
        mask = mask + torch.eye(d)[i_opt]
        curve.append(torch.stack([mse(forward_xmean(mask)) for _ in range(M)]).mean())
        actions.append(i_opt); R_hist.append(R); ims.append(im)
    return dict(info_curve=torch.stack(curve), action=torch.stack(actions, 1).float(), R_hist=torch.stack(R_hist),
                im=torch.stack(ims))
