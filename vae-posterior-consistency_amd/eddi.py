"""PNP / EDDI encoder family: drop-in classes for the reference's

    Reg_EDDI       src/models/VAE.py:670-853
    vanilla_EDDI   src/models/VAE.py:856-992

Same constructor arguments, `encoder` / `decoder` / `forward` / `loss` signatures, return order and state_dict keys
(type_pars1, type_bias1, prior_mean, prior_std, pnp_encoder1.0, pnp_encoder2.{0,2,4}, seq_decoder.{0,2,4}).  Decoder
and loss are those of Reg_VAE / vanilla_VAE (same HIP kernels, inherited); the encoder is the point-net front-end
(csrc/vpc_eddi.hip: folded per-feature affine + ReLU + mask-weighted sum, nothing of size B*d*(2+K) materialised)
followed by pnp_encoder2 as three fp32 MFMA GEMMs (csrc/vpc_gemm.hip).  No CPU fallback.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from ._lib import check, lib, ptr, require_cuda, stream_ptr
from .models import MAX_EPOCH, Reg_VAE, vanilla_VAE
from .notmiwae import ACT_NONE, ACT_RELU, linear_dgrad, linear_fwd, linear_wgrad, nm_sample, nm_sample_bwd, wgrad_reduce
from .ops import as_mask_u8

H1, H2 = 100, 50  # VAE.py:694-698 hard-codes 100 / 50


def eddi_fold(E, tb, Wp, cp, AC, d, K):
    check(lib().vpc_eddi_fold(ptr(E), ptr(tb), ptr(Wp), ptr(cp), ptr(AC), d, K, stream_ptr()), "vpc_eddi_fold")


def eddi_front_fwd(x, mask_u8, AC, agg, B, d, K, mask2_u8=None):
    """agg [B][K] (or [2B][K] when a second mask is given: the two passes of a step stacked)."""
    check(lib().vpc_eddi_front_fwd(ptr(x), ptr(mask_u8), ptr(mask2_u8), ptr(AC), ptr(agg), B, d, K, stream_ptr()),
          "vpc_eddi_front_fwd")


def eddi_front_bwd(x, mask_u8, AC, dagg, E, tb, Wp, gE, gtb, gWp, gcp, B, d, K, accumulate=False, mask2_u8=None,
                   scratch=None):
    rows = B * (2 if mask2_u8 is not None else 1)
    need = int(lib().vpc_eddi_front_scratch(rows, d, K))
    sc = scratch if scratch is not None and scratch.numel() >= need else torch.empty(need, device=x.device)
    check(lib().vpc_eddi_front_bwd(ptr(x), ptr(mask_u8), ptr(mask2_u8), ptr(AC), ptr(dagg), ptr(E), ptr(tb), ptr(Wp),
                                   ptr(sc), sc.numel(), ptr(gE), ptr(gtb), ptr(gWp), ptr(gcp), int(accumulate), B, d, K,
                                   stream_ptr()), "vpc_eddi_front_bwd")


class EDDIEncoderFn(torch.autograd.Function):
    """(x, mask, eps) -> (z, mean, logvar).  Reference: VAE.py:713-741 / 897-925."""

    @staticmethod
    def forward(ctx, model, x, mask_u8, eps, W1, b1, W2, b2, W3, b3, E, tb, Wp, cp):
        require_cuda(x, mask_u8, eps, W1, E)
        d, Ld, K = model.obs_dim, model.latent_dim, model.emb_dim
        B, dev = x.shape[0], x.device
        AC = torch.empty(2, K, d, device=dev)
        eddi_fold(E, tb, Wp, cp, AC, d, K)
        agg = torch.empty(B, K, device=dev)
        eddi_front_fwd(x, mask_u8, AC, agg, B, d, K)
        h1 = torch.empty(B, H1, device=dev)
        h2 = torch.empty(B, H2, device=dev)
        heads = torch.empty(B, 2 * Ld, device=dev)
        linear_fwd(agg, W1, b1, h1, B, H1, K, ACT_RELU)
        linear_fwd(h1, W2, b2, h2, B, H2, H1, ACT_RELU)
        linear_fwd(h2, W3, b3, heads, B, 2 * Ld, H2, ACT_NONE)
        z = torch.empty(B, Ld, device=dev)
        nm_sample(heads, eps, z, B, 1, Ld)
        ctx.model = model
        ctx.has_eps = eps is not None
        ctx.save_for_backward(x, mask_u8, AC, agg, h1, h2, heads, eps if eps is not None else torch.empty(0, device=dev),
                              W1, W2, W3, E, tb, Wp)
        return z, heads[:, :Ld], heads[:, Ld:]

    @staticmethod
    def backward(ctx, dz, dmean, dlogvar):
        model = ctx.model
        x, mask_u8, AC, agg, h1, h2, heads, eps, W1, W2, W3, E, tb, Wp = ctx.saved_tensors
        eps = eps if ctx.has_eps else None
        d, Ld, K = model.obs_dim, model.latent_dim, model.emb_dim
        B, dev = x.shape[0], x.device
        gh = torch.cat([dmean, dlogvar], 1).float().contiguous()
        dht = torch.empty(B, 2 * Ld, device=dev)
        nm_sample_bwd(dz.float().contiguous(), eps, heads, gh, dht, B, 1, Ld)
        e = lambda *s: torch.empty(*s, device=dev)
        gW1, gb1, gW2, gb2, gW3, gb3 = e(H1, K), e(H1), e(H2, H1), e(H2), e(2 * Ld, H2), e(2 * Ld)
        dh2, dh1, dagg = e(B, H2), e(B, H1), e(B, K)
        linear_wgrad(dht, h2, gW3, gb3, B, 2 * Ld, H2)
        linear_dgrad(dht, W3, dh2, B, 2 * Ld, H2, x_out=h2, act_prev=ACT_RELU)
        linear_wgrad(dh2, h1, gW2, gb2, B, H2, H1)
        linear_dgrad(dh2, W2, dh1, B, H2, H1, x_out=h1, act_prev=ACT_RELU)
        linear_wgrad(dh1, agg, gW1, gb1, B, H1, K)
        linear_dgrad(dh1, W1, dagg, B, H1, K)
        gE, gtb, gWp, gcp = e(d, K), e(d, 1), e(K, 2 + K), e(K)
        eddi_front_bwd(x, mask_u8, AC, dagg, E, tb, Wp, gE, gtb, gWp, gcp, B, d, K)
        return None, None, None, None, gW1, gb1, gW2, gb2, gW3, gb3, gE, gtb, gWp, gcp


class _EDDIBase:
    """Shared construction / parameter plumbing; mixed in BEFORE Reg_VAE / vanilla_VAE, whose decoder(), forward()
    and loss() (and the kernels behind them) are reused unchanged."""

    def _build(self, obs_dim, hid_dim, K, latent_dim, training_parameters, experiment_type, num_samples, num_estimates):
        nn.Module.__init__(self)
        if obs_dim > 128 or latent_dim > 15 or K > 32:
            raise L.VpcError("the gfx950 EDDI kernels support obs_dim <= 128, latent_dim <= 15 and K (emb_dim) <= 32")
        self.obs_dim, self.hid_dim, self.emb_dim, self.latent_dim = obs_dim, hid_dim, K, latent_dim
        self.K = K
        self.batch_size = training_parameters["batch_size"]
        self.training_parameters = training_parameters
        self.experiment_type = experiment_type
        self.num_samples, self.num_estimates = num_samples, num_estimates
        # containers in the reference's construction order (same seed -> same initial weights), VAE.py:687-709
        self.pnp_encoder1 = nn.Sequential(nn.Linear(2 + K, K), nn.ReLU())
        self.pnp_encoder2 = nn.Sequential(nn.Linear(K, H1), nn.ReLU(), nn.Linear(H1, H2), nn.ReLU(),
                                          nn.Linear(H2, 2 * latent_dim))
        self.seq_decoder = nn.Sequential(nn.Linear(latent_dim, H2), nn.ReLU(), nn.Linear(H2, H1), nn.ReLU(),
                                         nn.Linear(H1, obs_dim), nn.Sigmoid())
        xlv = torch.log(torch.square(torch.Tensor([0.1 * np.sqrt(2)])))
        self.register_buffer("x_logvar", xlv, persistent=False)
        self._x_logvar_value = float(xlv.item())
        self.type_pars1 = nn.Parameter(torch.zeros(obs_dim, K), requires_grad=True)
        nn.init.xavier_uniform_(self.type_pars1)
        self.type_bias1 = nn.Parameter(torch.zeros(obs_dim, 1), requires_grad=True)
        nn.init.xavier_uniform_(self.type_bias1)
        self.prior_mean = nn.Parameter(torch.zeros(latent_dim), requires_grad=False)
        self.prior_std = nn.Parameter(torch.ones(latent_dim), requires_grad=False)
        self.max_epoch = MAX_EPOCH
        self._layout = None
        self._img = None
        self._img_version = None
        self._part = {}

    # flat order: [pnp_encoder2 (6) | seq_decoder (6) | type_pars1, type_bias1, pnp_encoder1 (2)] - the decoder sits at
    # indices 6..11 as in the VAE classes, which is what DecoderFn's gradient split assumes
    def trainable(self):
        out = []
        for name in ("pnp_encoder2.0", "pnp_encoder2.2", "pnp_encoder2.4", "seq_decoder.0", "seq_decoder.2",
                     "seq_decoder.4"):
            mod = self.get_submodule(name)
            out += [mod.weight, mod.bias]
        return out + [self.type_pars1, self.type_bias1, self.pnp_encoder1[0].weight, self.pnp_encoder1[0].bias]

    def _versions(self):
        return tuple(p._version for p in self.trainable()) + (self.seq_decoder[0].weight.data_ptr(),)

    def _images(self):
        """Only the DECODER half of the packed image is used (the encoder is the front-end + GEMM trunk)."""
        lay = self._lay()
        flat = self.flatten_parameters()
        L.require_cuda(flat)
        v = self._versions()
        if self._img is None or self._img.device != flat.device:
            self._img = torch.from_numpy(lay.img_template).to(flat.device)
            self._img_version = None
        if self._img_version != v:
            pidx, _ = lay.device_tables(flat.device)
            n_trunk = sum(p.numel() for p in self.trainable()[:6])
            n_dec = lay.n_params - lay.n_enc
            ops.pack_weights(flat[n_trunk:n_trunk + n_dec], pidx[lay.n_enc:], self._img)
            self._img_version = v
        return self._img

    def decoder(self, z_int):
        """VAE.py:743-747: the Reg_VAE decoder kernels on this class's seq_decoder (trainable()[6:12])."""
        L.require_cuda(z_int)
        self._lay()
        self._images()
        return ops.DecoderFn.apply(self, z_int, *self.trainable()[6:12]), self.x_logvar

    def encoder(self, x, mask, sample=True):
        """VAE.py:713-741 / 897-925: returns (z, mean, logvar)."""
        L.require_cuda(x)
        if mask.shape[0] == 0:  # VAE.py:717-718
            return torch.empty(0, 10), torch.empty(0, 10), torch.empty(0, 10)
        self._images()
        xf = ops._f32c(x.reshape(-1, self.obs_dim))
        m = as_mask_u8(mask.reshape(-1, self.obs_dim).to(x.device))
        eps = torch.randn(xf.shape[0], self.latent_dim, device=xf.device) if sample else None
        t = self.trainable()
        return EDDIEncoderFn.apply(self, xf, m, eps, *t[:6], *t[12:])


class Reg_EDDI(_EDDIBase, Reg_VAE):
    """Reference: src/models/VAE.py:670-853."""

    def __init__(self, obs_dim, hid_dim, K, latent_dim, training_parameters, experiment_type, reg_type, num_samples=1,
                 num_estimates=1):
        self._build(obs_dim, hid_dim, K, latent_dim, training_parameters, experiment_type, num_samples, num_estimates)
        self.reg_type = reg_type

    def forward(self, data, mask, mask_p, stage="train"):
        return Reg_VAE.forward(self, data, mask, mask_p, stage)

    def loss(self, x, x_recon_p, x_logvar_p, mean_p, logvar_p, x_recon_q, x_logvar_q, mean_q, logvar_q, mask, mask_p,
             epoch, vae_elbo=False, llh_eval=False, MI=False, beta_annealing=False, beta=1.0, alpha=0.5, stage="train",
             alpha_annealing=False):
        """VAE.py:749-813: Reg_VAE.loss with this class's defaults (alpha = 0.5)."""
        return Reg_VAE.loss(self, x, x_recon_p, x_logvar_p, mean_p, logvar_p, x_recon_q, x_logvar_q, mean_q, logvar_q,
                            mask, mask_p, epoch, vae_elbo, llh_eval, MI, beta_annealing, beta, alpha, stage,
                            alpha_annealing)


class vanilla_EDDI(_EDDIBase, vanilla_VAE):
    """Reference: src/models/VAE.py:856-992."""

    def __init__(self, obs_dim, hid_dim, K, latent_dim, training_parameters, experiment_type, num_samples=1,
                 num_estimates=1):
        self._build(obs_dim, hid_dim, K, latent_dim, training_parameters, experiment_type, num_samples, num_estimates)

    def loss(self, x, x_recon_q, x_logvar_q, mean_q, logvar_q, epoch, mask, vae_elbo=False, llh_eval=False, MI=False,
             beta_annealing=False, beta=1.0, alpha=0.5, stage="train"):
        """VAE.py:933-964: vanilla_VAE.loss, except that RE_q_imputed is computed in EVERY stage (:938-939)."""
        return vanilla_VAE.loss(self, x, x_recon_q, x_logvar_q, mean_q, logvar_q, epoch, mask, vae_elbo, llh_eval, MI,
                                beta_annealing, beta, alpha, True, "evaluate")


# ------------------------------------------------------------------------------------------------ fused step
LP = 16  # row pitch of the padded latent workspaces of the fused decoder kernel


class EDDITrainer:
    """The EDDI training step (train.py:28-117 for 'reg_EDDI*' / 'vanilla_EDDI*') as a fixed launch sequence without
    host synchronisation: mask_p + eps draws -> fold -> front-end (both passes) -> pnp_encoder2 GEMMs on the stacked
    passes -> the SAME fused decoder + loss + decoder-backward kernel as the VAE step (nothing of size B x d is
    materialised) -> trunk backward GEMMs -> front-end backward -> [one all-reduce of [grads | loss terms]] -> flat
    Adam + decoder image re-pack."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, seed=0, process_group=None, world_size=1, rank=0):
        if not isinstance(model, _EDDIBase):
            raise TypeError("EDDITrainer supports Reg_EDDI and vanilla_EDDI")
        from .fused import FusedTrainer
        self.model = model
        self.vanilla = isinstance(model, vanilla_VAE)
        self.coefficients = lambda *a: FusedTrainer.coefficients(self, *a)
        self.lr, self.betas, self.eps = lr, betas, eps
        self.seed, self.rng_offset, self.step_count = seed, 0, 0
        self.pg, self.world_size, self.rank = process_group, world_size, rank
        self.lay = model._lay()
        flat = model.flatten_parameters()
        L.require_cuda(flat)
        self.dev = flat.device
        n = flat.numel()
        self.bucket = torch.zeros(n + 9, device=self.dev)
        self.grad, self.out9 = self.bucket[:n], self.bucket[n:]
        self.exp_avg = torch.zeros(n, device=self.dev)
        self.exp_avg_sq = torch.zeros(n, device=self.dev)
        self.accum = torch.zeros(1, device=self.dev)
        ncu = L.max_blocks()  # partial blocks any kernel may write
        self.partD = torch.empty(ncu * self.lay.dec_part, device=self.dev)
        self.loss_part = torch.empty(ncu, 8, dtype=torch.float64, device=self.dev)
        self.pidx, self.gidx = self.lay.device_tables(self.dev)
        # gradient views in flat order: trunk (6) | decoder (6) | front-end (4)
        self.g, off = [], 0
        for p in model.trainable():
            v = self.grad[off:off + p.numel()].view_as(p)
            p.grad = v
            self.g.append(v)
            off += p.numel()
        self.n_trunk = sum(p.numel() for p in model.trainable()[:6])
        self.n_dec = self.lay.n_params - self.lay.n_enc
        self._B = None
        self._flat_dec = None

    def _ws(self, B):
        if self._B == B:
            return
        m, dev = self.model, self.dev
        d, Ld, K = m.obs_dim, m.latent_dim, m.emb_dim
        P = 1 if self.vanilla else 2
        e = lambda *s: torch.empty(*s, device=dev)
        self.AC = e(2, K, d)
        self.agg, self.h1, self.h2, self.heads = e(P * B, K), e(P * B, H1), e(P * B, H2), e(P * B, 2 * Ld)
        self.lat = torch.zeros(P, 2, B, LP, device=dev)    # [pass][mean | logvar][B][16]
        self.dlat = torch.zeros(P, 2, B, LP, device=dev)
        self.dheads, self.dh2, self.dh1, self.dagg = e(P * B, 2 * Ld), e(P * B, H2), e(P * B, H1), e(P * B, K)
        self.eps_buf = e(3, B, LP)
        self.mask_p_buf = torch.empty(B, d, dtype=torch.uint8, device=dev)
        self.front_scratch = e(int(lib().vpc_eddi_front_scratch(P * B, d, K)))
        # per-layer partial buffers of the trunk's three weight gradients: summed by ONE launch (vpc_linear_wgrad_reduce)
        R = P * B
        self.wg_shapes = [(R, 2 * m.latent_dim, H2), (R, H2, H1), (R, H1, K)]
        self.wg_scratch = [e(int(lib().vpc_linear_wgrad_scratch(*sh))) for sh in self.wg_shapes]
        self._wg_cache = {}
        # the slices the step passes to its launches, made once per batch size (1-3 us of host time each; the step is host-paced)
        self._sl = dict(mean=[self.lat[p_, 0] for p_ in range(P)], logvar=[self.lat[p_, 1] for p_ in range(P)],
                        dmean=[self.dlat[p_, 0] for p_ in range(P)], dlogvar=[self.dlat[p_, 1] for p_ in range(P)],
                        eps=[self.eps_buf[p_] for p_ in range(3)], lat_dst=self.lat[..., :Ld],
                        heads_src=self.heads.view(P, B, 2, Ld).permute(0, 2, 1, 3), dheads_dst=self.dheads.view(P, B, 2, Ld),
                        dlat_src=self.dlat[..., :Ld].permute(0, 2, 1, 3),
                        gdec=self.grad[self.n_trunk:self.n_trunk + self.n_dec], gidx_dec=self.gidx[self.lay.n_enc:],
                        pidx_dec=self.pidx[self.lay.n_enc:])
        self._B = B

    def step(self, x, mask, mask_p=None, eps_q=None, eps_p=None, eps_ml=None, *, epoch=1, alpha=0.5, beta=1.0,
             beta_annealing=False, p_missingness=30, global_batch=None, row_lo=None):
        m, lay = self.model, self.lay
        d, Ld, K = m.obs_dim, m.latent_dim, m.emb_dim
        x = ops._f32c(x.reshape(-1, d))
        L.require_cuda(x)
        mask = as_mask_u8(mask.reshape(-1, d))
        B = x.shape[0]
        self._ws(B)
        Bg = global_batch if global_batch is not None else B * self.world_size
        co = self.coefficients(epoch, alpha, beta, beta_annealing)
        two = not self.vanilla
        P = 2 if two else 1
        t = m.trainable()
        W1, b1, W2, b2, W3, b3 = t[:6]
        E, tb, Wp, cp = t[12:]
        dec_img = m._dec_img()
        # ---- draws
        need_ml = two and co["wml"] != 0.0
        eps_view = self.eps_buf[: (3 if need_ml else 2 if two else 1)]
        inject = eps_q is not None
        if row_lo is None:
            row_lo = self.rank * B if self.world_size > 1 else 0
        # Philox counters of the GLOBAL row: draws do not depend on the world size
        eps_shard = (B, Bg, row_lo, LP)
        n_eps_groups = eps_view.shape[0] * Bg * (LP // 4)
        if two and mask_p is None:
            off_m = self.rng_offset
            self.rng_offset += (Bg * d + 7) // 8 + 1  # counters advance by what the GLOBAL batch consumes
            ops.draw_step(mask, self.mask_p_buf, 1.0 - p_missingness / 100.0, eps_view, self.seed, off_m,
                          self.rng_offset, None, row_lo * d, eps_shard)
            self.rng_offset += n_eps_groups
            mask_p = self.mask_p_buf
        else:
            if two:
                mask_p = as_mask_u8(mask_p.reshape(-1, d))
            if not inject:
                ops.fill_normal(eps_view, self.seed, self.rng_offset, None, eps_shard)
                self.rng_offset += n_eps_groups
        if eps_q is not None:
            self.eps_buf[0, :, :Ld].copy_(eps_q)
        if two and eps_p is not None:
            self.eps_buf[1, :, :Ld].copy_(eps_p)
        if need_ml and eps_ml is not None:
            self.eps_buf[2, :, :Ld].copy_(eps_ml)
        masks = [mask, mask_p] if two else [mask]
        # ---- encoder: front-end per pass, trunk on the stacked passes
        eddi_fold(E, tb, Wp, cp, self.AC, d, K)
        eddi_front_fwd(x, masks[0], self.AC, self.agg, B, d, K, masks[1] if two else None)  # both passes, one launch
        R = P * B
        linear_fwd(self.agg, W1, b1, self.h1, R, H1, K, ACT_RELU)
        linear_fwd(self.h1, W2, b2, self.h2, R, H2, H1, ACT_RELU)
        linear_fwd(self.h2, W3, b3, self.heads, R, 2 * Ld, H2, ACT_NONE)
        sl = self._sl
        sl["lat_dst"].copy_(sl["heads_src"])
        mean, logvar, dmean, dlogvar = sl["mean"], sl["logvar"], sl["dmean"], sl["dlogvar"]
        epss = sl["eps"][:P]
        eml = sl["eps"][2] if need_ml else None
        # ---- fused decoder + loss + decoder backward (the VAE step's kernel)
        maskB = [mask_p, None] if (two and co["cE"][0] != 0.0) else [None] * P
        nbD = ops.decoder_fused(x, dec_img, masks, maskB, co["cA"], co["cE"], mean, logvar, epss, eml, co["bq"],
                                co["bp"], co["cr"], co["wml"], 1.0 / Bg, m._x_logvar_value, dmean, dlogvar, self.partD,
                                self.loss_part, d, Ld, LP)
        ops.reduce_partials(self.partD, nbD, lay.dec_part, sl["gidx_dec"], sl["gdec"])
        cA1 = co["cA"][1] if two else 0.0
        ops.loss_finalize(self.loss_part, nbD, co["cA"][0], co["cE"][0], cA1, co["bq"], co["bp"], co["cr"], co["wml"], B,
                          Bg, d, self.out9, self.accum if self.world_size == 1 else None)
        # ---- encoder backward
        sl["dheads_dst"].copy_(sl["dlat_src"])
        g = self.g
        # (the three weight gradients leave their GEMMs as partials and are summed by ONE launch: two launches less per step)
        sc = self.wg_scratch
        linear_wgrad(self.dheads, self.h2, None, None, R, 2 * Ld, H2, scratch=sc[0])
        linear_dgrad(self.dheads, W3, self.dh2, R, 2 * Ld, H2, x_out=self.h2, act_prev=ACT_RELU)
        linear_wgrad(self.dh2, self.h1, None, None, R, H2, H1, scratch=sc[1])
        linear_dgrad(self.dh2, W2, self.dh1, R, H2, H1, x_out=self.h1, act_prev=ACT_RELU)
        linear_wgrad(self.dh1, self.agg, None, None, R, H1, K, scratch=sc[2])
        linear_dgrad(self.dh1, W1, self.dagg, R, H1, K)
        wgrad_reduce([(sc[0], R, 2 * Ld, H2, g[4], g[5], False), (sc[1], R, H2, H1, g[2], g[3], False),
                      (sc[2], R, H1, K, g[0], g[1], False)], self._wg_cache)
        eddi_front_bwd(x, masks[0], self.AC, self.dagg, E, tb, Wp, g[12], g[13], g[14], g[15], B, d, K,
                       mask2_u8=masks[1] if two else None, scratch=self.front_scratch)
        if self.world_size > 1:  # ONE collective per step: RCCL on the compute stream, or torch.distributed (dist.py)
            from . import dist as dp_mod
            if not getattr(self, "_coll_ready", False):
                self.collective = dp_mod.make_collective(self.world_size, self.rank, self.dev, self.pg)
                self._coll_ready = True
            dp_mod.allreduce_bucket(self.bucket, self.pg, self.collective)
        self.step_count += 1
        dp = self.world_size > 1  # the Adam launch also adds the all-reduced loss to the epoch accumulator
        ops.adam_step(m._flat, self.grad, self.exp_avg, self.exp_avg_sq, self.step_count, self.lr, self.betas[0],
                      self.betas[1], self.eps, loss_in=self.out9 if dp else None, accum=self.accum if dp else None)
        # keep the packed decoder image in step with the parameters (the version check would re-pack it anyway)
        if self._flat_dec is None or self._flat_dec[0] is not m._flat:
            self._flat_dec = (m._flat, m._flat[self.n_trunk:self.n_trunk + self.n_dec])
        ops.pack_weights(self._flat_dec[1], sl["pidx_dec"], m._img)
        m._img_version = m._versions()

    def loss_value(self) -> float:
        return float(self.out9[0].item())

    def epoch_total(self, reset=True) -> float:
        v = float(self.accum.item())
        if reset:
            self.accum.zero_()
        return v
