"""One JSON line per BASELINE.json config that runs on one GPU (configs 2, 3, 5; config 1 is the CPU plumbing case and is
timed on the host; config 4 needs the 8-GPU node: bench.py --gpus 8).  Synthetic stand-ins of the right shape (no datasets
ship with the reference): "UCI gas" = 13 910 rows x 128 features, MCAR 30 %.

    python tools/bench_configs.py > profiles/r02_configs.jsonl
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vpc_amd as vpc  # noqa: E402
from oracle import vae_oracle as O  # noqa: E402

dev = torch.device("cuda:0")
TP = {"batch_size": 64, "patience": 100}


def epoch_time(trainer_step, batches, repeats=3):
    for _ in range(3):  # warm-up incl. the one-time costs of a kernel's first launches (module load, LDS attribute)
        for b in batches[:40]:
            trainer_step(*b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(repeats):
        for b in batches:
            trainer_step(*b)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / repeats


def config1():
    """UCI Boston (506 x 14), vanilla_VAE, batch 256, the reference's CPU path (the oracle's torch port)."""
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    g = torch.Generator().manual_seed(0)
    x = torch.rand(506, 14, generator=g); m = torch.rand(506, 14, generator=g) < 0.7
    tr = O.TorchTrainer(O.init_params(14, 10, seed=0), 10, vanilla=True)
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 3.0:
        for i in range(0, 506, 256):
            tr.step(x[i:i + 256], m[i:i + 256])
        n += 1
    dt = (time.perf_counter() - t0) / n
    return dict(config=1, workload="UCI Boston stand-in 506x14, vanilla_VAE, batch 256, CPU port (plumbing)", epoch_ms=dt * 1e3,
                samples_per_s=506 / dt, device="host CPU", cores=torch.get_num_threads())


def config2(batch, precision):
    N, d = 13910, 128
    g = torch.Generator().manual_seed(1)
    x = torch.rand(N, d, generator=g).to(dev); m = (torch.rand(N, d, generator=g) < 0.7).to(dev)
    model = vpc.Reg_VAE(d, 500, 10, 10, TP, "bench", "kl_reg").to(dev)
    tr = vpc.FusedTrainer(model, precision=precision)
    batches = [(x[i:i + batch], m[i:i + batch]) for i in range(0, N, batch)]
    dt = epoch_time(lambda xb, mb: tr.step(xb, mb, alpha=1.0, p_missingness=30), batches)
    return dict(config=2, workload=f"UCI gas stand-in {N}x{d}, Reg_VAE kl_reg, batch {batch}, fused step", dtype=precision,
                epoch_ms=dt * 1e3, samples_per_s=N / dt, steps_per_epoch=len(batches), us_per_step=dt / len(batches) * 1e6)


def config3(batch, precision):
    from vpc_amd import notmiwae as nm
    N, d, K = 13910, 128, 20
    g = torch.Generator().manual_seed(2)
    x = torch.rand(N, d, generator=g).to(dev); m = (torch.rand(N, d, generator=g) < 0.7).float().to(dev)
    model = nm.REG_notMIWAE_v2(d, 128, 10, 10, {"batch_size": batch, "patience": 1}, K, 1).to(dev)
    tr = nm.NMTrainer(model, precision=precision)
    batches = [(x[i:i + batch], m[i:i + batch]) for i in range(0, N - batch + 1, batch)]
    dt = epoch_time(lambda xb, mb: tr.step(xb, mb, alpha=0.5, p_missingness=50), batches, repeats=2)
    n = len(batches) * batch
    return dict(config=3, workload=f"UCI gas stand-in (MNAR), REG_notMIWAE_v2 K={K}, batch {batch}, p_missingness 50", dtype=precision,
                epoch_ms=dt * 1e3, samples_per_s=n / dt, steps_per_epoch=len(batches), us_per_step=dt / len(batches) * 1e6)


def config5():
    """active_learning_func on a (briefly) trained Reg_VAE: d = 128, n_test = 256, M = 50 (imputation_args.json), one repeat."""
    d, n, M = 128, 256, 50
    g = torch.Generator().manual_seed(3)
    base = torch.rand(n + 2048, 4, generator=g); mix = torch.rand(4, d, generator=g)
    data = torch.sigmoid(3.0 * (base @ mix / mix.sum(0) - 0.5)) + 0.05 * torch.rand(n + 2048, d, generator=g)
    data = (data - data.min(0).values) / (data.max(0).values - data.min(0).values)
    model = vpc.Reg_VAE(d, 500, 10, 10, TP, "bench", "kl_reg").to(dev)
    tr = vpc.FusedTrainer(model)
    xt, mt = data[n:].to(dev), (torch.rand(2048, d, generator=g) < 0.7).to(dev)
    for i in range(300):
        tr.step(xt, mt, alpha=1.0, epoch=i + 1)
    x, tm = data[:n], torch.rand(n, d, generator=g) < 0.7
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = vpc.active_learning_func(None, x, tm, 30, d, 500, 10, M, 10, "toy", TP, "exp", "reg_vae1", 100, 1, 1, alpha=1.0,
                                   p_missingness=30, reg_type="kl_reg", Repeat=1, model=model, save=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    curve = out["information_curve_CHAI"][0, 0]
    return dict(config=5, workload=f"active_learning_func, Reg_VAE d={d}, n_test={n}, M={M}, {d - 1} acquisition steps", seconds=dt,
                ms_per_acquisition_step=dt / (d - 1) * 1e3, encoder_evaluations_replaced=4 * (d - 1) * (d - 1) * M * n,
                target_mse_start=float(curve[0]), target_mse_end=float(curve[-1]))


if __name__ == "__main__":
    # every line in its own process: trainers of different shapes / precisions in one process perturb each other's timings
    # (allocator state, lazily loaded code objects)
    import subprocess
    if len(sys.argv) > 1:
        fn, args = sys.argv[1], sys.argv[2:]
        r = {"config1": config1, "config2": config2, "config3": config3, "config5": config5}[fn](
            *[int(a) if a.isdigit() else a for a in args])
        print(json.dumps(r), flush=True)
        sys.exit(0)
    jobs = [["config1"]] + [["config2", "64", p] for p in ("f32", "bf16x3", "bf16")] + \
           [["config3", "128", p] for p in ("f32", "bf16x3", "bf16")] + [["config5"]]
    for j in jobs:
        out = subprocess.run([sys.executable, os.path.abspath(__file__)] + j, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        print(line[-1] if line else json.dumps({"job": j, "error": out.stderr[-300:]}), flush=True)
