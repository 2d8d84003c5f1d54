"""bench.py - training samples/sec of the VAE posterior-consistency step (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = the build's counterpart of src/experiment_main/train.py:53-117 for Reg_VAE(kl_reg, alpha=1,
beta=1): on-device mask_p draw + eps draw -> 2 x encoder + 2 x decoder forward -> ELBO + consistency loss ->
backward -> (one flat all-reduce) -> Adam, on synthetic batches x ~ U(0,1) [B=65 536, d=128], mask ~
Bern(0.7), already resident in HBM.  The timed loop ROTATES over 8 distinct resident batches (336 MB of x + mask,
more than the 256 MiB Infinity Cache), as an epoch streams different rows every step.
Scaling: "weak" (default) - every rank steps its own B rows (global batch N*B); "strong" (--scaling strong) - the
global batch stays B = 65 536 and every rank steps B/N rows of it (north_star's "partition minibatches").  Either
way the per-step gradient all-reduce (RCCL) is inside the timed region, ranks share ONE Philox seed and the draws
are keyed by the global row.
Warm-up: the W requested steps, then "settle" steps in groups of 8 until the dominant kernel's HIP-event time of two
consecutive groups agrees within 1.5 % (clock ramp after idle; at most 40 groups, all untimed; reported as
`settle_steps`), then EXACTLY K timed steps.

Prints ONE JSON line (rank 0) with the contract's fields plus
  roofline     - dominant kernel (vpc_decoder_fused: reparam + decoder fwd + loss + decoder bwd): algorithmic
                 FLOP per launch / HIP-event duration measured live on the launch stream, vs the f32 MFMA peak
  cpu_baseline - the oracle's op-for-op torch port of the reference CPU path (oracle/vae_oracle.py), timed on
                 this box's host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import gc
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2516.6  # same guide: bf16 MFMA = 16 x the fp32 MFMA rate (~2.5 PF dense)
# MFMA peak the algorithmic FLOPs are priced against: a bf16x3 product issues three bf16 MFMAs per fp32 product
PEAK = {"f32": F32_MFMA_PEAK_TFLOPS, "bf16x3": BF16_MFMA_PEAK_TFLOPS / 3.0, "bf16": BF16_MFMA_PEAK_TFLOPS}
HBM_PEAK_TBS = 8.0


def flops_per_sample(d, L, passes=2):
    """Algorithmic FLOP per sample: fwd F, bwd 2F - d*H1 (no dgrad to x)."""
    enc_f = d * 100 + 100 * 50 + 50 * 2 * L
    dec_f = L * 50 + 50 * 100 + 100 * d
    enc = enc_f + (2 * enc_f - d * 100)
    dec = 3 * dec_f
    return dict(total=2 * passes * (enc + dec), decoder_fused=2 * passes * dec, encoder_fwd=2 * passes * enc_f,
                encoder_bwd=2 * passes * (2 * enc_f - d * 100))


def measured_traffic(B, d, L, precision="f32"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/r02_traffic.json:
    FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE).  PMC counters cannot be read from inside this process, so
    the number is the one measured with `rocprofv3 --pmc` on this same command; None for other shapes."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r02_traffic.json")))
        if (B, d, L) == (65536, 128, 10):
            ks = t["precisions"][precision]
            return next(v["hbm_bytes_corrected"] for k, v in ks.items() if "dec8" in k)
    except Exception:
        pass
    return None


def measured_step_traffic(B, d, L, precision="f32"):
    """HBM bytes per STEP: the three MFMA kernels from the committed PMC passes (profiles/r02_traffic.json) plus the
    algorithmic traffic of the two small launches (reduce_step reads 256 partial blocks = 49 MB and writes < 1 MB; draw_step
    reads the mask and writes mask_p + eps = 22 MB).  None for shapes / precisions without a PMC pass."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r02_traffic.json")))
        if (B, d, L) != (65536, 128, 10):
            return None
        ks = t["precisions"]["bf16x3" if precision == "bf16" else precision]  # bf16 moves the same bytes as bf16x3
        return sum(v["hbm_bytes_corrected"] for v in ks.values()) + 49_000_000 + 22_000_000
    except Exception:
        return None


def cpu_baseline(B, d, L, seconds=20.0, threads=None):
    """Reference CPU path (port), fp32: 2 warm-up steps + timed steps, median; then the same with autograd anomaly
    detection ON (the reference's drivers enable it, src/experiment_main/imputation.py:19) as a second row.
    `threads` defaults to the cores this job may use (the GPU box gives a 1-GPU job a 16-core share of a 128-core
    host; torch's default of 128 threads oversubscribes that share 8x and runs the step 6x slower: capped at 16)."""
    from oracle import vae_oracle as O
    if threads is None:
        threads = min(16, len(os.sched_getaffinity(0)))
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(B, d, generator=g)
    mask = torch.rand(B, d, generator=g) < 0.7

    def run(budget, anomaly):
        torch.manual_seed(0)
        tr = O.TorchTrainer(O.init_params(d, L, seed=0), L, reg_type="kl_reg")
        times = []
        t_end = time.perf_counter() + budget
        with torch.autograd.set_detect_anomaly(anomaly):
            for i in range(2 + 30):
                t0 = time.perf_counter()
                tr.step(x, mask, p_missingness=30, alpha=1.0, beta=1.0, epoch=1)
                dt = time.perf_counter() - t0
                if i >= 2:
                    times.append(dt)
                if i >= 4 and time.perf_counter() > t_end:
                    break
        return statistics.median(times), len(times)

    med, n = run(seconds * 0.6, False)
    med_on, n_on = run(seconds * 0.4, True)
    what = (f"steps (median) after 2 warm-up of the same B={B} d={d} Reg_VAE kl_reg step (numpy mask_p draw + forward + "
            f"loss + backward + Adam + .item())")
    return dict(value=B / med, unit="samples/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{n} {what}, anomaly-detect off", ms_per_step=med * 1e3,
                anomaly_on=dict(value=B / med_on, unit="samples/s", ms_per_step=med_on * 1e3,
                                sample=f"{n_on} {what}, torch.autograd.set_detect_anomaly(True) as imputation.py:19"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=65536, help="rows per GPU (weak) / global rows (strong)")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--latent", type=int, default=10)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--precision", choices=["f32", "bf16x3", "bf16"], default="f32",
                    help="f32 = the headline (exact fp32 MFMA); bf16x3 / bf16 = extra lines on v_mfma_f32_16x16x32_bf16")
    ap.add_argument("--batches", type=int, default=8, help="distinct resident batches the loop rotates over")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-settle", action="store_true", help="skip the settle steps after the W warm-up steps")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--cpu-threads", type=int, default=None)
    args = ap.parse_args()

    import vpc_amd as vpc
    rank, world, local = vpc.dp.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    d, Ld = args.dim, args.latent
    if args.scaling == "strong":
        Bg = args.batch
        lo, hi = vpc.dp.shard_rows(Bg, rank, world)
        B = hi - lo
    else:
        B, Bg, lo = args.batch, args.batch * world, rank * args.batch

    torch.manual_seed(0)
    model = vpc.Reg_VAE(d, 500, 10, Ld, {"batch_size": B, "patience": 100}, "bench", "kl_reg").to(dev)
    flat = model.flatten_parameters()
    vpc.dp.broadcast_parameters(flat, model=model)
    # this rank's rows of `--batches` distinct global batches, resident in HBM (8 x 42 MB > the 256 MiB Infinity Cache)
    g = torch.Generator().manual_seed(1234 + rank)
    xs = [torch.rand(B, d, generator=g).to(dev) for _ in range(args.batches)]
    masks = [(torch.rand(B, d, generator=g) < 0.7).to(dev) for _ in range(args.batches)]
    # one shared seed: Philox counters are keyed by the global row, so ranks draw disjoint parts of one stream
    tr = vpc.FusedTrainer(model, lr=1e-3, seed=0, world_size=world, rank=rank, precision=args.precision)
    kw = dict(alpha=1.0, beta=1.0, p_missingness=30, epoch=1, global_batch=Bg, row_lo=lo)
    it = [0]

    def step():
        i = it[0] % args.batches
        it[0] += 1
        tr.step(xs[i], masks[i], **kw)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()

    for _ in range(args.warmup):
        step()
    sync()
    # CPython's generation-2 garbage collection fires once around step 70 of the timed loop (the ctypes argument arrays of
    # ~350 launches) and, with torch's millions of live objects, pauses the host for 20-40 ms - longer than the work
    # queued on the GPU at that point (rocprofv3 trace: one 38 ms gap, nothing else).  Collect here, BEFORE the settle
    # steps: a collection right in front of the timed region leaves the GPU idle for tens of ms and the first timed steps
    # then run on ramped-down clocks (a 20-step region read 0.369 instead of 0.340 ms / step).
    gc.collect()
    gc.disable()
    # settle: until the dominant kernel's event time is stable (all ranks run the same number of groups)
    settle = 0
    if not args.no_settle:
        tr.timers, tr.timer_names, tr.timer_every = {}, {"decoder_fused"}, 1
        prev = None
        for _ in range(40):
            tr.timers.clear()
            for _ in range(8):
                step()
            settle += 8
            torch.cuda.synchronize()
            cur = statistics.mean(a.elapsed_time(b) for a, b in tr.timers["decoder_fused"])
            done = prev is not None and abs(cur - prev) <= 0.015 * prev
            if world > 1:
                flag = torch.tensor([1.0 if done else 0.0], device=dev)
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
                done = bool(flag.item() > 0.5)
            if done:
                break
            prev = cur
        tr.timers, tr.timer_every = None, 8
        sync()
    tr.epoch_total()  # reset the device-side loss accumulator: loss_mean below covers the timed steps only
    # Inside the timed region only the dominant kernel is bracketed by HIP events (every 8th step); the other launches
    # are sampled right after the region.
    tr.timers, tr.timer_names = {}, {"decoder_fused"}
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    gc.enable()
    total = tr.epoch_total()  # one host read per "epoch", as train.py:118 (the timed steps only)
    timers, tr.timers = tr.timers, {}
    tr.timer_names, tr.timer_every = {"encoder_fwd", "encoder_bwd"}, 2
    for _ in range(16):
        step()
    sync()
    timers.update(tr.timers)
    tr.timers = None
    # the reference reads the loss on the host every step (train.py:117 `.item()`): the same steps with that sync
    n_sync = min(args.steps, 50)
    sync()
    t1 = time.perf_counter()
    for _ in range(n_sync):
        step()
        tr.loss_value()
    sync()
    per_step_sync_ms = (time.perf_counter() - t1) / n_sync * 1e3
    if world > 1:
        t = torch.tensor([elapsed, per_step_sync_ms], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed, per_step_sync_ms = float(t[0].item()), float(t[1].item())
        torch.distributed.barrier()
    if rank != 0:
        vpc.dp.shutdown()
        return
    if not (total == total):
        raise SystemExit("loss is NaN")

    ms_step = elapsed / args.steps * 1e3
    value = Bg * args.steps / elapsed
    fl = flops_per_sample(d, Ld)
    kern_ms = {k: statistics.mean(a.elapsed_time(b) for a, b in v) for k, v in timers.items()}
    dom = "decoder_fused"
    achieved = fl[dom] * B / (kern_ms[dom] * 1e-3) / 1e12
    rows = f"B={B} per GPU" if args.scaling == "weak" else f"global B={Bg} sharded {B} rows per GPU"
    out = {
        "metric": "training samples/sec (ELBO+consistency step), B=65536 d=128",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": args.precision,
        "data": "synthetic",
        "config": {"workload": f"Reg_VAE kl_reg alpha=1 beta=1 training step, synthetic tabular {rows}, "
                               f"d={d}, L={Ld}, MCAR mask 0.7, p_missingness=30, Adam lr=1e-3, "
                               f"{args.batches} resident batches rotated",
                   "global_batch": Bg, "parallelism": f"dp{world}"},
        "roofline": {"bound": "mfma",
                     "kernel": ("vpc::dec_kernel<8,true,FUSED,1> (small-batch shape)" if tr.last_blocks[1] > (B + 127) // 128
                                else "vpc::dec8_kernel<8,true>") + " (vpc_decoder_fused)",
                     "achieved": achieved, "peak": PEAK[args.precision], "unit": "TFLOP/s",
                     "frac": achieved / PEAK[args.precision],
                     "traffic": measured_traffic(B, d, Ld, args.precision),
                     "flop_per_launch": fl[dom] * B, "avg_launch_ms": kern_ms[dom]},
        "kernels_ms": kern_ms,
        "step_hbm": (lambda tb: None if tb is None else {
            "bytes_per_step": tb, "achieved_TBps": tb / (ms_step * 1e-3) / 1e12, "peak_TBps": HBM_PEAK_TBS,
            "frac": tb / (ms_step * 1e-3) / 1e12 / HBM_PEAK_TBS,
            "compulsory_bytes_per_step": 768 * B})(measured_step_traffic(B, d, Ld, args.precision)),
        "step_tflops_algorithmic": fl["total"] * B * world / (ms_step * 1e-3) / 1e12,
        "settle_steps": settle,
        "per_step_sync_ms": per_step_sync_ms,
        "loss_mean": total / args.steps,
    }
    if world == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline(B, d, Ld, args.cpu_seconds, args.cpu_threads)
        out["cpu_baseline"] = cb
        out["speedup_vs_cpu"] = value / cb["value"]
    print(json.dumps(out))
    vpc.dp.shutdown()


if __name__ == "__main__":
    main()
