"""Small-batch / strong-scaling timings of the fused Reg_VAE step.

    python tools/bench_small.py [--batches 64,256,1024,8192,16384,32768] [--steps 300] [--graph]

For every batch size: us / step of FusedTrainer.step with the 16-row N-split kernel forced (VPC_TILE=16: csrc/vpc_small.hip,
one launch for the whole step), with the throughput shape forced (VPC_TILE=128: 128-row tiles, 8
waves, both passes looped in the workgroup) and with the small-batch shape forced (VPC_TILE=64: 64-row tiles, one wave
per SIMD, passes spread over blockIdx.y), device-side draws and Adam included, plus per-kernel HIP-event times.
Prints one JSON line per (B, shape)."""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vpc_amd as vpc  # noqa: E402


def run(B, d, tile, steps, graph):
    os.environ["VPC_TILE"] = tile
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = vpc.Reg_VAE(d, 500, 10, 10, {"batch_size": B, "patience": 1}, "bench", "kl_reg").to(dev)
    tr = vpc.FusedTrainer(m, seed=1)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(B, d, generator=g).to(dev)
    mask = (torch.rand(B, d, generator=g) < 0.7).to(dev)
    fn = tr.step_graph if graph else tr.step
    for _ in range(60):
        fn(x, mask, alpha=1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn(x, mask, alpha=1.0)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / steps * 1e6
    kern = {}
    if not graph:
        tr.timers, tr.timer_every = {}, 1
        for _ in range(20):
            tr.step(x, mask, alpha=1.0)
        torch.cuda.synchronize()
        kern = {k: round(statistics.median(a.elapsed_time(b) for a, b in v) * 1e3, 1) for k, v in tr.timers.items()}
    return dict(B=B, d=d, tile=int(tile), graph=graph, us_per_step=round(us, 1), blocks=tr.last_blocks, kernels_us=kern,
                loss=tr.loss_value())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="64,256,1024,8192,16384,32768")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--graph", action="store_true")
    a = ap.parse_args()
    for B in [int(v) for v in a.batches.split(",")]:
        for tile in ("128", "64", "16"):  # 16 = the N-split kernel (csrc/vpc_small.hip)
            print(json.dumps(run(B, a.dim, tile, a.steps, False)), flush=True)
            if a.graph:
                print(json.dumps(run(B, a.dim, tile, a.steps, True)), flush=True)


if __name__ == "__main__":
    main()
