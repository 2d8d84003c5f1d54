"""Fused training step: the build's counterpart of the reference's step body
src/experiment_main/train.py:53-117 for Reg_VAE / vanilla_VAE:

    mask_p draw -> forward (2 x encoder, 2 x decoder) -> loss -> zero_grad -> backward -> Adam -> loss accumulate

as 5 kernel launches (6 under data parallelism) and no B x d intermediate:

    vpc_draw_step            mask_p = mask & Bernoulli(1 - p_missingness/100) and eps_q, eps_p (, eps_ml), Philox
                             counters keyed by the GLOBAL row                        (train.py:53-55, VAE.py:389-392)
    vpc_encoder_fwd          both passes, writes h1/h2 workspaces + mean/logvar      (VAE.py:387-395)
    vpc_decoder_fused        reparameterise + decoder + loss + seeds + decoder bwd   (VAE.py:397-467)
    vpc_encoder_bwd          encoder backward of both passes                         (train.py:115)
    vpc_reduce_step_adam     per-workgroup partial blocks -> flat gradient (fixed order) + loss scalar + epoch
                             accumulator + flat Adam + re-pack of the weight images  (train.py:114-117, no host sync)
      data parallel / graph replay: vpc_reduce_step -> [ONE all-reduce of the flat bucket: grads + loss terms]
                             -> vpc_adam_step
    precision "bf16", obs_dim in (64, 128], throughput shape: encoder_fwd + decoder_fused + encoder_bwd are ONE launch,
    vpc_step_fused_bf16 (csrc/vpc_step.hip) - h1 / h2 / latent statistics never leave the chip

Gradients are those of the GLOBAL mean loss: seeds carry 1/B_global, so summing the flat bucket over ranks is
exactly `train_loss = loss / x.shape[0]` (VAE.py:452) on the concatenated batch.
"""
from __future__ import annotations

import torch

from . import _lib as L
from . import dist as dp_mod
from . import ops
from .models import MAX_EPOCH, Reg_VAE, vanilla_VAE
from .ops import H1P, H2P, as_mask_u8

LP = 16  # row pitch of the padded latent workspaces


class FusedTrainer:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, seed=0, process_group=None, world_size=1, rank=0,
                 precision="f32", collective=None):
        """precision: "f32" (v_mfma_f32_16x16x4_f32, the parity path), "bf16x3" (split-bf16 products on
        v_mfma_f32_16x16x32_bf16: fp32-class accuracy) or "bf16" (plain bf16 inputs, fp32 accumulation and loss math);
        the bf16 forms exist for Reg_VAE / vanilla_VAE with obs_dim % 4 == 0, both workgroup shapes (csrc/vpc_bf16.h).
        collective: carrier of the per-step bucket under data parallelism (dist.FlatAllReduce = ncclAllReduce on the
        compute stream); None = chosen on the first multi-rank step by dist.make_collective (RCCL when the process group
        is NCCL, torch.distributed.all_reduce otherwise)."""
        if not isinstance(model, (Reg_VAE, vanilla_VAE)):
            raise TypeError("FusedTrainer supports Reg_VAE and vanilla_VAE")
        if getattr(model, "_wide", False):
            raise L.VpcError("FusedTrainer covers encoder inputs <= 128 wide and latent_dim <= 15; use wide.WideTrainer "
                             "(harness.train does) for wider models")
        self.model = model
        self.vanilla = isinstance(model, vanilla_VAE)
        self.lr, self.betas, self.eps = lr, betas, eps
        self.seed = seed
        self.rng_offset = 0
        self.step_count = 0
        self.pg = process_group
        self.world_size = world_size
        self.rank = rank  # data parallel: this rank's rows are [rank * B, (rank + 1) * B) of the global batch by default
        self.collective = collective
        self._coll_ready = collective is not None
        # the data-parallel tail (reduce_step -> all-reduce -> adam_step) instead of the single fused launch
        self.dp = world_size > 1 or collective is not None
        self.lay = model._lay()
        flat = model.flatten_parameters()
        L.require_cuda(flat)
        self.dev = flat.device
        if precision not in ops.PRECISIONS:
            raise ValueError(f"precision {precision!r}: expected one of {sorted(ops.PRECISIONS)}")
        self.precision = precision
        self.prec = ops.PRECISIONS[precision]
        if self.prec:
            lay = self.lay
            if lay.mask_augm or lay.d % 4:
                raise L.VpcError("the bf16 / bf16x3 kernels cover the plain encoder with obs_dim % 4 == 0 (<= 128)")
            self.pidx_bf, tmpl, self.enc_img_bf = lay.bf16_tables(self.dev)
            self.img_bf = torch.from_numpy(tmpl).to(self.dev)
            # plain bf16, obs_dim in (64, 128]: the whole-step kernel (csrc/vpc_step.hip) with its own compact image
            self._step_ok = self.prec == 2 and 64 < lay.d <= 128
            if self._step_ok:
                self.pidx_c, tmpl_c = lay.step_tables(self.dev)
                self.img_c = torch.from_numpy(tmpl_c).to(self.dev)
            # the bf16 images follow the flat parameters lazily: an optimiser step marks them stale, the next launch that
            # needs one re-packs it (one pack launch per step while the batch shape does not change)
            self._stale = {"pair": True, "step": True}
        n = self.lay.n_params
        # one flat bucket: [grads (n) | loss terms (9 floats)] -> a single all-reduce per step under DP
        self.bucket = torch.zeros(n + 9, device=self.dev)
        self.grad = self.bucket[:n]
        self.out9 = self.bucket[n:]
        self.exp_avg = torch.zeros(n, device=self.dev)
        self.exp_avg_sq = torch.zeros(n, device=self.dev)
        self.accum = torch.zeros(1, device=self.dev)
        ncu = L.max_blocks()  # partial blocks any kernel may write (2 x CUs: small-batch shape)
        self.partE = torch.empty(ncu * self.lay.enc_part, device=self.dev)
        self.partD = torch.empty(ncu * self.lay.dec_part, device=self.dev)
        self.loss_part = torch.empty(ncu, 8, dtype=torch.float64, device=self.dev)
        self.pidx, self.gidx = self.lay.device_tables(self.dev)
        self.inv = self.lay.inverse_maps(self.dev)  # caller-owned maps for the layout-order gradient reduction
        self._ws_B = None
        self._pads = {}
        self.timers = None  # bench.py sets this to {} to collect per-kernel HIP event pairs
        self.timer_every = 8  # ... on every 8th step only: an event pair costs ~5 us of GPU idle time per kernel
        self._timer_tick = 0
        self.timer_names = None  # restrict the event pairs to these launches (None = all)
        # make trainable tensors' .grad views of the flat gradient, so state is inspectable like torch's
        off = 0
        for p in model.trainable():
            p.grad = self.grad[off:off + p.numel()].view_as(p)
            off += p.numel()

    # ------------------------------------------------------------------
    def _pad_cols(self, t, dk, slot):
        """[B, d] -> zero-padded [B, dk] (a persistent buffer per input slot)."""
        key = (slot, t.shape[0], dk, t.dtype)
        buf = self._pads.get(key)
        if buf is None:
            buf = self._pads[key] = torch.zeros(t.shape[0], dk, dtype=t.dtype, device=self.dev)
        buf[:, :t.shape[1]].copy_(t)
        return buf

    def _workspaces(self, B, d):
        if self._ws_B == (B, d):
            return
        dev, Ld = self.dev, self.lay.L
        np_ = 1 if self.vanilla else 2
        self.h1 = [torch.empty(B, H1P, device=dev) for _ in range(np_)]
        self.h2 = [torch.empty(B, H2P, device=dev) for _ in range(np_)]
        # latent statistics live in padded [B][16] workspaces (16-byte vector access in the kernels)
        self.mean = [torch.empty(B, LP, device=dev) for _ in range(np_)]
        self.logvar = [torch.empty(B, LP, device=dev) for _ in range(np_)]
        self.dmean = [torch.empty(B, LP, device=dev) for _ in range(np_)]
        self.dlogvar = [torch.empty(B, LP, device=dev) for _ in range(np_)]
        self.eps_buf = torch.empty(3, B, LP, device=dev)
        self.mask_p_buf = torch.empty(B, d, dtype=torch.uint8, device=dev)
        self._ws_B = (B, d)

    def _step_ws(self, B):
        """Workspace of the whole-step kernel (the packed seeds between its two sweeps)."""
        n = ops.step_workspace_floats(B)
        if getattr(self, "_ws_step", None) is None or self._ws_step.numel() < n:
            self._ws_step = torch.empty(n, device=self.dev)
        return self._ws_step

    def _timed(self, name, fn, *args):
        """Run one launch; with timers enabled bracket it with events on the launch stream."""
        if self.timers is None or self._timer_tick % self.timer_every or \
                (self.timer_names is not None and name not in self.timer_names):
            return fn(*args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*args)
        e1.record()
        self.timers.setdefault(name, []).append((e0, e1))
        return r

    def dominant_launch(self):
        """Name (as in `timers`) of the launch that carries most of the step's FLOPs in the shape last stepped."""
        if getattr(self, "_used_step_small", False):
            return "step_small"
        return "step_fused" if getattr(self, "_used_step_fused", False) else "decoder_fused"

    def coefficients(self, epoch, alpha, beta, beta_annealing):
        """Loss coefficients of the generic form in csrc/vpc_dec.hip (VAE.py:425-446 / 1183-1195)."""
        bw = (epoch / MAX_EPOCH) * beta if beta_annealing else beta
        if self.vanilla:
            return dict(cA=[1.0], cE=[0.0], bq=bw, bp=0.0, cr=0.0, wml=0.0)
        rt = self.model.reg_type
        if rt == "kl_reg":
            return dict(cA=[1.0 - alpha, alpha], cE=[alpha, 0.0], bq=(1.0 - alpha) * bw, bp=alpha * bw, cr=alpha,
                        wml=0.0)
        if rt == "ml_reg":
            return dict(cA=[1.0, 0.0], cE=[0.0, 0.0], bq=bw, bp=0.0, cr=0.0, wml=(epoch / MAX_EPOCH) * alpha)
        print("Not implemented!")  # VAE.py:447-449
        raise NotImplementedError(f"reg_type {rt!r}")

    # ------------------------------------------------------------------
    def step(self, x, mask, mask_p=None, eps_q=None, eps_p=None, eps_ml=None, *, epoch=1, alpha=1.0, beta=1.0,
             beta_annealing=False, p_missingness=30, global_batch=None, row_lo=None, update=True, _state=None):
        """One training step on the local rows `x` [B, d] (fp32, GPU), `mask` [B, d] (bool / uint8 / float).
        mask_p / eps_* are drawn on the device unless injected (parity tests).  Returns nothing: the loss of
        this step is in `self.out9[0]` (device), the running total in `self.accum` (train.py:117).
        Data parallel: `global_batch` rows in all (default B * world_size) of which this rank holds
        [row_lo, row_lo + B) (default rank * B).  The Philox counters of the draws are those of the GLOBAL row
        (SURVEY.md section 8e): with one shared seed every row gets the mask_p / eps it would get in the single-process
        step on the concatenated batch, whatever the world size."""
        L.require_cuda(x)
        self._timer_tick += 1
        lay, m = self.lay, self.model
        B, d, Ld = x.shape[0], lay.d, lay.L
        x = ops._f32c(x)
        mask = as_mask_u8(mask)
        # Row pitch of the kernels' x / mask arrays.  obs_dim % 4 != 0 would take the scalar-load kernel variants (the ones
        # that spill, profiles/r01_kernel_resources.txt): instead the inputs are copied into arrays padded to a multiple of
        # 4 columns with mask 0 in the padding - never observed, so it adds nothing to any sum or gradient, and the weight
        # images already hold zeros for the columns / rows past obs_dim - and the 16-byte-vector variants run on those.
        dk = d if (d % 4 == 0 or lay.mask_augm) else 4 * ((d + 3) // 4)
        if dk != d:
            x, mask = self._pad_cols(x, dk, 0), self._pad_cols(mask, dk, 1)
            if mask_p is not None:
                mask_p = self._pad_cols(as_mask_u8(mask_p), dk, 2)
        self._workspaces(B, dk)
        Bg = global_batch if global_batch is not None else B * self.world_size
        if row_lo is None:
            row_lo = self.rank * B if self.world_size > 1 else 0
        if row_lo < 0 or row_lo + B > Bg:
            raise ValueError(f"rows [{row_lo}, {row_lo + B}) are not inside the global batch of {Bg}")
        co = self.coefficients(epoch, alpha, beta, beta_annealing)
        two = not self.vanilla
        # Small batches run the fp32 N-split kernel (csrc/vpc_small.hip) in EVERY precision: `precision` bounds the rounding a step may
        # use, and at these sizes the fp32 kernel is both the most accurate and the fastest form (B = 64: 41 us against 62 us for the
        # bf16 engine's small-shape kernels).  VPC_STEP_SMALL=0 in the environment keeps the requested engine (tests of its kernels).
        use_small = not lay.mask_augm and B <= ops.step_small_max_rows()  # (<= 16 x CUs rows)
        self._used_step_small = use_small
        use_step = (not use_small) and bool(self.prec) and self._step_ok and ops.step_fused_applicable(B, dk, Ld, 2 if two else 1)
        self._used_step_fused = use_step
        # the model's fp32 images (what the fp32 kernels read and the Adam launches re-pack); the whole-step bf16 kernel has its
        # own image and leaves these stale (invalidate_images below), so it must not ask for them every step
        img = None if use_step else m._images()
        enc_img, dec_img = (None, None) if use_step else (img[:lay.enc_img], img[lay.enc_img:])
        if self.prec and not use_small:  # bf16 images, re-packed from the flat parameters after an optimiser step (lazily, see __init__)
            enc_img, dec_img = self.img_bf[:self.enc_img_bf], self.img_bf[self.enc_img_bf:]
            kind = "step" if use_step else "pair"
            if self._stale[kind]:
                if use_step:
                    ops.step_pack_weights_bf16(m._flat, self.pidx_c, self.img_c)
                else:
                    ops.pack_weights_bf16(m._flat, self.pidx_bf, self.img_bf)
                self._stale[kind] = False
        rng0 = self.rng_offset
        # ---- random draws (mask_p and eps in ONE launch when both are drawn on the device)
        need_ml = two and co["wml"] != 0.0
        draw_eps = eps_q is None or (two and eps_p is None) or (need_ml and eps_ml is None)
        eps_view = self.eps_buf[: (3 if need_ml else 2 if two else 1)]  # only the draws this step consumes
        # counters advance by what the GLOBAL batch consumes, so that all ranks stay on one stream of counters
        n_eps_groups = eps_view.shape[0] * Bg * (LP // 4)
        eps_shard = (B, Bg, row_lo, LP)
        # small batches: the N-split kernel makes the step's draws itself (same counters, one launch less) when ALL of them are
        # drawn on the device
        all_dev = eps_q is None and (not two or eps_p is None) and (not need_ml or eps_ml is None)
        fuse_draw = use_small and all_dev and (not two or mask_p is None) and LP == 16
        draw_args = None
        if fuse_draw:
            off_m = self.rng_offset
            if two:
                self.rng_offset += (Bg * dk + 7) // 8 + 1
                mask_p = self.mask_p_buf
            draw_args = (mask if two else None, 1.0 - p_missingness / 100.0, eps_view, self.seed, off_m, self.rng_offset, _state,
                         row_lo * dk, eps_shard)
            self.rng_offset += n_eps_groups
            draw_eps = False
        elif two and mask_p is None:
            off_m = self.rng_offset
            self.rng_offset += (Bg * dk + 7) // 8 + 1
            if draw_eps:
                self._timed("draw_step", ops.draw_step, mask, self.mask_p_buf, 1.0 - p_missingness / 100.0, eps_view,
                            self.seed, off_m, self.rng_offset, _state, row_lo * dk, eps_shard)
                self.rng_offset += n_eps_groups
                draw_eps = False
            else:
                ops.draw_mask(mask, self.mask_p_buf, 1.0 - p_missingness / 100.0, self.seed, off_m, row_lo * dk)
            mask_p = self.mask_p_buf
        elif two:
            mask_p = as_mask_u8(mask_p)
        if draw_eps:
            # `_state`: under graph replay the offset lives on the device (a frozen host offset would replay one eps)
            ops.fill_normal(eps_view, self.seed, self.rng_offset, _state, eps_shard)
            self.rng_offset += n_eps_groups
        if eps_q is not None:  # injected draws (parity tests) arrive dense [B][L]; pad entries are ignored
            self.eps_buf[0, :, :Ld].copy_(eps_q)
        if two and eps_p is not None:
            self.eps_buf[1, :, :Ld].copy_(eps_p)
        if need_ml and eps_ml is not None:
            self.eps_buf[2, :, :Ld].copy_(eps_ml)
        eq = self.eps_buf[0]
        ep = self.eps_buf[1] if two else None
        eml = self.eps_buf[2] if need_ml else None
        masks = [mask, mask_p] if two else [mask]
        epss = [eq, ep] if two else [eq]
        maskB = [mask_p, None] if (two and co["cE"][0] != 0.0) else [None] * len(masks)
        if use_small:
            # ---- small batch (fp32 arithmetic in every precision): the whole step in ONE launch, 16-row tiles with the feature tiles split over the waves
            if draw_args is not None:
                nbE = nbD = self._timed("step_small", ops.step_small_draw_f32, x, enc_img, dec_img, masks, maskB, co["cA"],
                                        co["cE"], epss, eml, co["bq"], co["bp"], co["cr"], co["wml"], 1.0 / Bg, m._x_logvar_value,
                                        self.partE, self.partD, self.loss_part, dk, Ld, *draw_args)
            else:
                nbE = nbD = self._timed("step_small", ops.step_small_f32, x, enc_img, dec_img, masks, maskB, co["cA"], co["cE"],
                                        epss, eml, co["bq"], co["bp"], co["cr"], co["wml"], 1.0 / Bg, m._x_logvar_value,
                                        self.partE, self.partD, self.loss_part, dk, Ld)
        elif use_step:
            # ---- plain bf16, throughput shape: encoder forward + decoder + loss + all backward in ONE launch
            nbE = nbD = self._timed("step_fused", ops.step_fused_bf16, x, self.img_c, masks, maskB, co["cA"], co["cE"], epss,
                                    eml, co["bq"], co["bp"], co["cr"], co["wml"], 1.0 / Bg, m._x_logvar_value, self.partE,
                                    self.partD, self.loss_part, self._step_ws(B), dk, Ld)
        else:
            # ---- forward (encoder), fused decoder + loss + decoder backward, encoder backward
            self._timed("encoder_fwd", ops.encoder_fwd, x, enc_img, masks, None, self.h1, self.h2, self.mean, self.logvar,
                        None, dk, Ld, LP, lay.mask_augm, self.prec)
            nbD = self._timed("decoder_fused", ops.decoder_fused, x, dec_img, masks, maskB, co["cA"], co["cE"], self.mean,
                              self.logvar, epss, eml, co["bq"], co["bp"], co["cr"], co["wml"], 1.0 / Bg, m._x_logvar_value,
                              self.dmean, self.dlogvar, self.partD, self.loss_part, dk, Ld, LP, self.prec)
            nbE = self._timed("encoder_bwd", ops.encoder_bwd, x, enc_img, masks, self.h1, self.h2, self.dmean,
                              self.dlogvar, self.partE, dk, Ld, LP, lay.mask_augm, self.prec)
        # ---- flat gradient + loss terms (+ Adam when nothing has to happen between them): one launch
        cA1 = co["cA"][1] if two else 0.0
        self.last_blocks = (nbE, nbD)
        if update and not self.dp and _state is None:
            self.step_count += 1
            if use_step:
                # the fused Adam re-packs the whole-step kernel's compact bf16 image itself (no pack launch); the model's fp32
                # images and the pair-slot bf16 image go stale and are re-packed by whoever needs them next
                self._timed("reduce_step", ops.reduce_step_adam, self.partE, nbE, lay.enc_part, self.partD, nbD,
                            lay.dec_part, self.gidx, self.grad, lay.n_enc, self.loss_part, nbD, co["cA"][0], co["cE"][0],
                            cA1, co["bq"], co["bp"], co["cr"], co["wml"], B, Bg, d, self.out9, self.accum, m._flat,
                            self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0], self.betas[1], self.eps,
                            self.step_count, self.pidx_c, self.img_c, self.inv, True)
                m.invalidate_images()
                self._stale = {"pair": True, "step": False}
                return
            self._timed("reduce_step", ops.reduce_step_adam, self.partE, nbE, lay.enc_part, self.partD, nbD,
                        lay.dec_part, self.gidx, self.grad, lay.n_enc, self.loss_part, nbD, co["cA"][0], co["cE"][0],
                        cA1, co["bq"], co["bp"], co["cr"], co["wml"], B, Bg, d, self.out9, self.accum, m._flat,
                        self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0], self.betas[1], self.eps,
                        self.step_count, self.pidx, img, self.inv)
            if self.prec:
                self._stale = {"pair": True, "step": True}
            return
        ops.reduce_step(self.partE, nbE, lay.enc_part, self.partD, nbD, lay.dec_part,
                    self.gidx, self.grad, lay.n_enc, self.loss_part, nbD, co["cA"][0], co["cE"][0], cA1, co["bq"],
                    co["bp"], co["cr"], co["wml"], B, Bg, d, self.out9,
                    None if self.dp else self.accum, _state,
                    self.rng_offset - rng0 if _state is not None else 0, self.inv)
        if self.dp:
            self._allreduce()
        if update:
            self.step_count += 1
            # under data parallelism the Adam launch also adds the all-reduced loss to the epoch accumulator
            if use_step:
                m.invalidate_images()
            ops.adam_step(m._flat, self.grad, self.exp_avg, self.exp_avg_sq, self.step_count,
                        self.lr, self.betas[0], self.betas[1], self.eps, None if use_step else self.pidx, img,
                        None if _state is None else _state[0:1],
                        loss_in=self.out9 if self.dp else None, accum=self.accum if self.dp else None)
            if self.prec:
                self._stale = {"pair": True, "step": True}
            if use_step and _state is not None:
                # graph capture: the lazy re-pack at the start of the next eager step is not part of a replay - the compact
                # image of the whole-step kernel is re-packed here, inside the captured sequence
                ops.step_pack_weights_bf16(m._flat, self.pidx_c, self.img_c)
                self._stale["step"] = False
        elif self.dp:
            self.accum += self.out9[0]

    # ------------------------------------------------------------------ HIP-graph replay of the step
    def step_graph(self, x, mask, *, epoch=1, alpha=1.0, beta=1.0, beta_annealing=False, p_missingness=30):
        """Same step, replayed from a captured HIP graph (torch.cuda.CUDAGraph): one host call per step instead
        of six kernel launches - what matters at the reference's own batch sizes (64 / 128), where the step is
        launch-bound.  Step count and Philox offsets live on the device (`state`), since kernel arguments are
        frozen in a graph.  Draws are always on the device; the first call with a new (shape, coefficients) runs
        one eager step and captures.  Under data parallelism the graph holds reduce_step -> ncclAllReduce -> adam_step
        when the bucket travels through dist.FlatAllReduce (RCCL on the compute stream); with a torch.distributed
        carrier (gloo) it falls back to step()."""
        if self.dp and not isinstance(self._collective(), dp_mod.FlatAllReduce):
            return self.step(x, mask, epoch=epoch, alpha=alpha, beta=beta, beta_annealing=beta_annealing,
                             p_missingness=p_missingness)
        L.require_cuda(x)
        x = ops._f32c(x)
        mask = as_mask_u8(mask)
        co = self.coefficients(epoch, alpha, beta, beta_annealing)
        key = (tuple(x.shape), p_missingness, tuple(co["cA"]), tuple(co["cE"]), co["bq"], co["bp"], co["cr"], co["wml"])
        kw = dict(epoch=epoch, alpha=alpha, beta=beta, beta_annealing=beta_annealing, p_missingness=p_missingness)
        if getattr(self, "_graph_key", None) != key:
            self.step(x, mask, **kw)  # eager warm-up: also sets the LDS attributes, workspaces, packed image
            self._gx, self._gmask = x.clone(), mask.clone()
            self.state = torch.tensor([self.step_count, 0], dtype=torch.int64, device=self.dev)
            timers, self.timers = self.timers, None
            base_rng, base_step = self.rng_offset, self.step_count
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.step(self._gx, self._gmask, _state=self.state, **kw)
            self._graph_rng_inc = self.rng_offset - base_rng
            self.rng_offset, self.step_count = base_rng, base_step  # capture executed nothing
            self._timer_tick -= 1
            self.timers = timers
            self._graph, self._graph_key = g, key
            return
        if x.data_ptr() != self._gx.data_ptr():
            self._gx.copy_(x)
        if mask.data_ptr() != self._gmask.data_ptr():
            self._gmask.copy_(mask)
        self._graph.replay()
        self.step_count += 1
        self.rng_offset += self._graph_rng_inc

    def _collective(self):
        if not self._coll_ready:
            self.collective = dp_mod.make_collective(self.world_size, self.rank, self.dev, self.pg)
            self._coll_ready = True
        return self.collective

    def _allreduce(self):
        """ONE collective per step over the flat bucket [grads | loss terms]: ncclAllReduce (RCCL over xGMI) on the
        compute stream when the process group is NCCL, torch.distributed.all_reduce otherwise (dist.py).  Every term
        is already normalised by the GLOBAL batch, so a plain SUM is the result of the concatenated batch."""
        dp_mod.allreduce_bucket(self.bucket, self.pg, self._collective())

    def loss_value(self) -> float:
        """Loss of the last step (host sync)."""
        return float(self.out9[0].item())

    def epoch_total(self, reset=True) -> float:
        """Sum of train_loss over the steps since the last reset (train.py:117-118; one host sync per epoch)."""
        v = float(self.accum.item())
        if reset:
            self.accum.zero_()
        return v
