"""Drop-in model classes with the reference's API, backed by the HIP kernels.

Mirrors (signatures, return order, state_dict keys, quirks) of the reference classes
    Reg_VAE      src/models/VAE.py:350-507
    vanilla_VAE  src/models/VAE.py:1119-1240
so that the reference's train / evaluate / active-learning code can call them unchanged:
`model.forward(...)`, `model.loss(...)` -> `(print_loss, train_loss[, extras])`, `train_loss.backward()`,
`model.encoder(x, mask, sample=True)`, `model.decoder(z)`, `state_dict()` with keys
`prior_mean, prior_std, seq_encoder.{0,2,4}.{weight,bias}, seq_decoder.{0,2,4}.{weight,bias}`.

Two execution paths share the same kernels:
  * the API path (this file): forward / loss as separate calls, autograd.Functions around the encoder,
    decoder and fused-loss (K4) kernels;
  * the fused training step (`fused.FusedTrainer`): mask draw + both passes + loss + backward + Adam without
    materialising any B x d intermediate.
There is no CPU fallback: calling forward / loss with CPU tensors raises.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .ops import DecoderFn, EncoderFn, LossFn, as_mask_u8

MAX_EPOCH = 2800  # VAE.py:384

_ENC = ("seq_encoder.0", "seq_encoder.2", "seq_encoder.4")
_DEC = ("seq_decoder.0", "seq_decoder.2", "seq_decoder.4")


class _VAEBase(nn.Module):
    mask_augm = False  # True: encoder input is [x*mask | mask] (the reference's *_mask classes)

    def __init__(self, obs_dim, hid_dim, K, latent_dim, training_parameters, experiment_type, num_samples=1,
                 num_estimates=1):
        super().__init__()
        enc_in = 2 * obs_dim if self.mask_augm else obs_dim
        # encoder inputs wider than the 128 columns the register-chained kernels tile (or latent_dim > 15) run the same
        # API on the generic GEMM kernels instead (wide.py): the reference takes any obs_dim (VAE.py:366-376)
        self._wide = enc_in > 128 or obs_dim > 128 or latent_dim > 15
        if latent_dim > 64 or obs_dim > 4096:
            raise L.VpcError("supported shapes: obs_dim <= 4096, latent_dim <= 64")
        self.obs_dim = obs_dim
        self.hid_dim = hid_dim  # ignored by the reference too (VAE.py:366-376 hard-codes 100 / 50)
        self.latent_dim = latent_dim
        self.K = K
        self.num_samples = num_samples
        self.num_estimates = num_estimates
        self.training_parameters = training_parameters
        self.experiment_type = experiment_type
        # containers for the parameters (never called as modules); same indices / init as the reference
        self.seq_encoder = nn.Sequential(nn.Linear(enc_in, 100), nn.ReLU(), nn.Linear(100, 50), nn.ReLU(),
                                         nn.Linear(50, 2 * latent_dim))
        self.seq_decoder = nn.Sequential(nn.Linear(latent_dim, 50), nn.ReLU(), nn.Linear(50, 100), nn.ReLU(),
                                         nn.Linear(100, obs_dim), nn.Sigmoid())
        # VAE.py:379 - log((0.1*sqrt 2)^2), shape (1,); kept as a non-persistent buffer so that it follows .to()
        xlv = torch.log(torch.square(torch.Tensor([0.1 * np.sqrt(2)])))
        self.register_buffer("x_logvar", xlv, persistent=False)
        self._x_logvar_value = float(xlv.item())
        self.prior_mean = nn.Parameter(torch.zeros(latent_dim), requires_grad=False)
        self.prior_std = nn.Parameter(torch.ones(latent_dim), requires_grad=False)
        self.max_epoch = MAX_EPOCH
        self._layout = None
        self._img = None
        self._img_version = None
        self._part = {}

    # ------------------------------------------------------------------ parameter plumbing
    def trainable(self):
        """The 12 trainable tensors in state_dict (= flat) order.  (Cached: nn.Module attribute lookups cost ~0.4 ms per
        training step on the API path, which is host-bound at the reference's batch sizes; .to() / load_state_dict keep
        the Parameter objects.)"""
        out = self.__dict__.get("_trainable_cache")
        if out is None or out[0] is not self.seq_encoder[0].weight:
            out = []
            for name in _ENC + _DEC:
                mod = self.get_submodule(name)
                out += [mod.weight, mod.bias]
            self.__dict__["_trainable_cache"] = out
        return out

    def _lay(self):
        if self._layout is None:
            self._layout = L.layout(self.obs_dim, self.latent_dim, self.mask_augm)
        return self._layout

    def flatten_parameters(self):
        """Make the 12 trainable tensors views of ONE flat fp32 buffer (state_dict order).  Idempotent; call again
        after .to(device).  Returns the flat buffer."""
        ps = self.trainable()
        flat = self.__dict__.get("_flat")
        # fast path: first and last parameter still sit where the flat buffer puts them
        if flat is not None and ps[0].data_ptr() == flat.data_ptr() and \
                ps[-1].data_ptr() == flat.data_ptr() + 4 * (flat.numel() - ps[-1].numel()) and flat.device == ps[0].device:
            return flat
        off = 0
        ok = flat is not None and flat.device == ps[0].device
        if ok:
            for p in ps:
                if p.data.data_ptr() != flat.data_ptr() + 4 * off or not p.data.is_contiguous():
                    ok = False
                    break
                off += p.numel()
        if not ok:
            flat = torch.cat([p.data.detach().reshape(-1).float() for p in ps]).contiguous()
            off = 0
            for p in ps:
                p.data = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
            self._flat = flat
            self._img_version = None
        return self._flat

    def _versions(self):
        """What the packed image was built from: the version counters of the 12 parameters AND of the flat buffer they
        are views of (an in-place write through `_flat` - dist.broadcast, a raw kernel - bumps only the latter).  Writes
        through `p.data` bump neither: call invalidate_images() after those."""
        flat = self.__dict__.get("_flat")
        ps = self.trainable()
        return tuple(p._version for p in ps) + (ps[0].data_ptr(), -1 if flat is None else flat._version)

    def invalidate_images(self):
        """Force a re-pack of the weight images on the next forward (after writing parameters through `.data`)."""
        self._img_version = None

    def _images(self):
        """Packed weight images [enc | dec], re-packed when any parameter changed (in-place version counters)."""
        lay = self._lay()
        flat = self.flatten_parameters()
        L.require_cuda(flat)
        v = self._versions()
        if self._img is None or self._img.device != flat.device:
            self._img = torch.from_numpy(lay.img_template).to(flat.device)
            self._img_version = None
        if self._img_version != v:
            pidx, _ = lay.device_tables(flat.device)
            ops.pack_weights(flat, pidx, self._img)
            self._img_version = v
        return self._img

    def _enc_img(self):
        return self._images()[: self._lay().enc_img]

    def _dec_img(self):
        return self._images()[self._lay().enc_img:]

    def _partials(self, device, which):
        lay = self._lay()
        key = (which, str(device))
        if key not in self._part:
            n = lay.enc_part if which == "enc" else lay.dec_part
            self._part[key] = torch.empty(L.max_blocks() * n, device=device)
        return self._part[key]

    def _split_flat(self, flat, lo, hi):
        out, off = [], 0
        for p in self.trainable()[lo:hi]:
            out.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        return out

    # ------------------------------------------------------------------ reference API
    def encoder(self, x, mask, sample=True):
        """VAE.py:387-395: returns (z, mean, logvar); eps ~ N(0,1) drawn on the device when sample=True."""
        L.require_cuda(x)
        x = x.reshape(-1, self.obs_dim)
        xf = x.contiguous() if x.dtype == torch.float32 else x.float().contiguous()
        m = as_mask_u8(mask.reshape(-1, self.obs_dim).to(x.device))
        eps = torch.randn(xf.shape[0], self.latent_dim, device=xf.device) if sample else None
        if self._wide:
            from .wide import WideEncoderFn
            self.flatten_parameters()
            return WideEncoderFn.apply(self, xf, m, eps, *self.trainable()[:6])
        self._lay()
        self._images()
        return EncoderFn.apply(self, xf, m, eps, *self.trainable()[:6])

    def decoder(self, z_int):
        """VAE.py:397-401: returns (x_mean, x_logvar) with x_logvar the shape-(1,) constant."""
        L.require_cuda(z_int)
        if self._wide:
            from .wide import WideDecoderFn
            self.flatten_parameters()
            return WideDecoderFn.apply(self, z_int, *self.trainable()[6:]), self.x_logvar
        self._lay()
        self._images()
        return DecoderFn.apply(self, z_int, *self.trainable()[6:]), self.x_logvar

    def _xlv(self, x_logvar):
        if isinstance(x_logvar, torch.Tensor) and x_logvar.numel() != 1:
            raise NotImplementedError("only the reference's constant x_logvar (shape (1,)) is supported")
        return self._x_logvar_value

    def _kl_std_small(self, mean, log_var):
        """kl_diagnormal_stdnormal2 on the aggregated (L,) statistics of the MI branch (VAE.py:458-461)."""
        return torch.sum(0.5 * (torch.exp(log_var) + mean * mean - 1.0 - log_var))

    def _finish(self, train_loss, sums, B, d, llh_eval, MI, stage, mean_q, logvar_q, imputed_sum_idx=7):
        print_loss = train_loss
        c = ops.HALF_LOG_2PI * B * d
        if llh_eval:
            RE_q = ((sums[0] + c) / B).float()
            RE_imp = ((sums[imputed_sum_idx] + c) / B).float() if stage == "evaluate" else 0.0
            return print_loss, train_loss, RE_q, RE_imp
        if MI:
            KL_q = (sums[3] / B).float()
            KL_agg = self._kl_std_small(torch.mean(mean_q, 0), torch.mean(logvar_q, 0))
            return print_loss, train_loss, KL_q - KL_agg, KL_q
        return print_loss, train_loss


class Reg_VAE(_VAEBase):
    """Reg VAE (posterior-consistency regulariser).  Reference: src/models/VAE.py:350-507."""

    def __init__(self, obs_dim, hid_dim, K, latent_dim, training_parameters, experiment_type, reg_type, num_samples=1,
                 num_estimates=1):
        super().__init__(obs_dim, hid_dim, K, latent_dim, training_parameters, experiment_type, num_samples,
                         num_estimates)
        self.reg_type = reg_type

    def forward(self, data, mask, mask_p, stage="train"):
        # VAE.py:496-507 - q pass first (eps_q), then p pass (eps_p); p outputs are returned first
        if type(self).encoder is _VAEBase.encoder and not self._wide:
            # both encoder passes as one autograd node / one launch each way (ops.RegEncoderFn); eps_q is drawn before
            # eps_p, as the two rsample() calls of the reference are
            L.require_cuda(data)
            self._images()
            x = data.reshape(-1, self.obs_dim)
            xf = x.contiguous() if x.dtype == torch.float32 else x.float().contiguous()
            mq = as_mask_u8(mask.reshape(-1, self.obs_dim).to(x.device))
            mp = as_mask_u8(mask_p.reshape(-1, self.obs_dim).to(x.device))
            eps_q = torch.randn(xf.shape[0], self.latent_dim, device=xf.device)  # two draws, as the two rsample() calls
            eps_p = torch.randn(xf.shape[0], self.latent_dim, device=xf.device)
            z_q, mean_q, logvar_q, z_p, mean_p, logvar_p = ops.RegEncoderFn.apply(self, xf, mq, mp, eps_q, eps_p,
                                                                                 *self.trainable()[:6])
        else:  # subclasses with their own encoder (EDDI) and the wide path: pass by pass
            z_q, mean_q, logvar_q = self.encoder(data, mask)
            z_p, mean_p, logvar_p = self.encoder(data, mask_p)
        x_mean_q, x_logvar_q = self.decoder(z_q)
        x_mean_p, x_logvar_p = self.decoder(z_p)
        return mean_p, logvar_p, x_mean_p, x_logvar_p, mean_q, logvar_q, x_mean_q, x_logvar_q

    def loss(self, x, x_recon_p, x_logvar_p, mean_p, logvar_p, x_recon_q, x_logvar_q, mean_q, logvar_q, mask, mask_p,
             epoch, vae_elbo=False, llh_eval=False, MI=False, beta_annealing=False, beta=1.0, alpha=0.8, stage="train",
             alpha_annealing=True):
        """VAE.py:403-467.  One fused kernel (K4) computes every sum and, when gradients are needed, every seed."""
        L.require_cuda(x, x_recon_q, mean_q)
        B, d = x.shape[0], self.obs_dim
        xlv = self._xlv(x_logvar_q)
        bw = (epoch / self.max_epoch) * beta if beta_annealing else beta
        xf = ops._f32c(x)
        mA = as_mask_u8(mask.to(x.device))
        cfg = dict(d=d, L=self.latent_dim, x_logvar=xlv, wml=0.0, cr=0.0, bp=0.0)
        eps_ml = None
        if stage == "evaluate":
            cfg.update(maskA=[mA], maskB=[None], cA=[1.0], cE=[0.0], bq=float(bw))
            loss, sums = LossFn.apply(cfg, xf, x_recon_q.contiguous(), None, mean_q.contiguous(),
                                      logvar_q.contiguous(), None, None, None)
        else:
            mP = as_mask_u8(mask_p.to(x.device))
            if self.reg_type == "kl_reg":  # VAE.py:441-446
                cfg.update(maskA=[mA, mP], maskB=[mP, None], cA=[1.0 - alpha, alpha], cE=[alpha, 0.0],
                           bq=float((1.0 - alpha) * bw), bp=float(alpha * bw), cr=float(alpha))
            elif self.reg_type == "ml_reg":  # VAE.py:435-440 (draws one more eps)
                cfg.update(maskA=[mA, mP], maskB=[None, None], cA=[1.0, 0.0], cE=[0.0, 0.0], bq=float(bw),
                           wml=float((epoch / self.max_epoch) * alpha))
                eps_ml = torch.randn(B, self.latent_dim, device=x.device)
            else:  # VAE.py:447-449 prints and sets loss = 0, which then fails in backward; fail early instead
                print("Not implemented!")
                raise NotImplementedError(f"reg_type {self.reg_type!r}")
            loss, sums = LossFn.apply(cfg, xf, x_recon_q.contiguous(), x_recon_p.contiguous(), mean_q.contiguous(),
                                      logvar_q.contiguous(), mean_p.contiguous(), logvar_p.contiguous(), eps_ml)
        return self._finish(loss, sums, B, d, llh_eval, MI, stage, mean_q, logvar_q)


class vanilla_VAE(_VAEBase):
    """vanilla_VAE.  Reference: src/models/VAE.py:1119-1240."""

    def forward(self, data, mask):
        z_q, mean_q, logvar_q = self.encoder(data, mask)
        x_mean_q, x_logvar_q = self.decoder(z_q)
        return mean_q, logvar_q, x_mean_q, x_logvar_q

    def loss(self, x, x_recon_q, x_logvar_q, mean_q, logvar_q, epoch, mask, vae_elbo=False, llh_eval=False, MI=False,
             beta_annealing=False, beta=1.0, alpha=0.8, alpha_annealing=True, stage="train"):
        """VAE.py:1171-1208 (mask may be float: train.py:58,97 multiplies it by ones)."""
        L.require_cuda(x, x_recon_q, mean_q)
        B, d = x.shape[0], self.obs_dim
        xlv = self._xlv(x_logvar_q)
        bw = (epoch / self.max_epoch) * beta if beta_annealing else beta
        cfg = dict(d=d, L=self.latent_dim, x_logvar=xlv, wml=0.0, cr=0.0, bp=0.0, bq=float(bw),
                   maskA=[as_mask_u8(mask.to(x.device))], maskB=[None], cA=[1.0], cE=[0.0])
        loss, sums = LossFn.apply(cfg, ops._f32c(x), x_recon_q.contiguous(), None, mean_q.contiguous(),
                                  logvar_q.contiguous(), None, None, None)
        return self._finish(loss, sums, B, d, llh_eval, MI, stage, mean_q, logvar_q)


class Reg_VAE_mask(Reg_VAE):
    """Reg_VAE with the mask-augmented encoder input [x*mask | mask] (first layer 2d -> 100).
    Reference: src/models/VAE.py:510-667 (identical to Reg_VAE apart from encoder :545-555)."""
    mask_augm = True


class vanilla_VAE_mask(vanilla_VAE):
    """vanilla_VAE with the mask-augmented encoder input.  Reference: src/models/VAE.py:995-1116."""
    mask_augm = True
