"""Throughput of the MNAR training step (BASELINE config 3: REG_notMIWAE_v2, d = 128, K = 20, p_missingness = 50) on
one MI355X, with the per-kernel HIP-event breakdown, the GEMM roofline fraction and the CPU port timed beside it.

    python tools/bench_mnar.py [--batch 128] [--steps 50] [--warmup 10] [--vanilla] [--no-cpu] [--timers]

Prints one JSON line.  `value` = data rows (not K-replicated rows) per second, inputs resident in HBM.
FLOP model (algorithmic, fp32 MAC = 2 FLOP): forward per encoder row 2*(d*128 + 128*128 + 128*2L), per decoder row
2*(L*128 + 128*128 + 128*2d); backward = 2x forward (dgrad + wgrad) minus the first-layer dgrad of the encoder.
"""
import argparse
import gc
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpc_amd  # noqa: E402
from vpc_amd import notmiwae as nm  # noqa: E402

PEAK_F32_MFMA = 157.3  # TFLOP/s dense fp32 MFMA (MI355X_MICROARCH.md)


def flops_per_step(B, K, d, L, passes):
    enc_f = 2 * (d * 128 + 128 * 128 + 128 * 2 * L)
    dec_f = 2 * (L * 128 + 128 * 128 + 128 * 2 * d)
    enc = enc_f * 3 - 2 * d * 128          # no dgrad into x
    dec = dec_f * 3
    return passes * B * (enc + K * dec)


PEAK_MULT = {"f32": 1.0, "bf16x3": 16.0 / 3.0, "bf16": 16.0}  # bf16 MFMA = 16 x the fp32 MFMA rate; bf16x3 issues 3 per product


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--latent", type=int, default=10)
    ap.add_argument("--vanilla", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--timers", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured HIP graph")
    ap.add_argument("--precision", choices=["f32", "bf16x3", "bf16"], default="f32")
    a = ap.parse_args()
    B, d, K, L = a.batch, a.d, a.k, a.latent
    torch.manual_seed(0)
    cls = nm.notMIWAE_myversion if a.vanilla else nm.REG_notMIWAE_v2
    model = cls(d, 128, 10, L, {"batch_size": B, "patience": 1}, K, 1).cuda()
    tr = nm.NMTrainer(model, lr=1e-3, seed=0, precision=a.precision)
    x = torch.rand(B, d, device="cuda")
    m = (torch.rand(B, d, device="cuda") < 0.5).float()
    stepfn = tr.step_graph if a.graph else tr.step
    for _ in range(a.warmup):
        stepfn(x, m, alpha=0.5, p_missingness=50)
    torch.cuda.synchronize()
    if a.timers:
        tr.timers = {}
    gc.collect()
    gc.disable()  # a gen-2 collection inside the loop stalls the host for tens of ms (profiles/r01_notes.md)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        stepfn(x, m, alpha=0.5, p_missingness=50)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    gc.enable()
    passes = 1 if a.vanilla else 2
    fl = flops_per_step(B, K, d, L, passes)
    out = {"metric": "MNAR training samples/sec (REG_notMIWAE_v2 step, K=20)" if not a.vanilla else
           "MNAR training samples/sec (notMIWAE_myversion step, K=20)",
           "value": B / dt, "unit": "samples/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": dt * 1e3, "dtype": a.precision, "data": "synthetic",
           "config": {"workload": f"config 3: B={B} d={d} K={K} L={L} p_missingness=50 alpha=0.5", "graph": bool(a.graph)},
           "loss": tr.loss_value(),
           "roofline": {"bound": "mfma", "achieved": fl / dt / 1e12, "peak": PEAK_F32_MFMA * PEAK_MULT[a.precision],
                        "unit": "TFLOP/s", "frac": fl / dt / 1e12 / (PEAK_F32_MFMA * PEAK_MULT[a.precision]),
                        "scope": "whole step (all launches)"}}
    if a.timers:
        out["kernels_ms"] = {k: sum(e0.elapsed_time(e1) for e0, e1 in v) / a.steps for k, v in tr.timers.items()}
    if not a.no_cpu:
        from oracle import notmiwae_oracle as O
        nthr = min(16, os.cpu_count() or 1)
        torch.set_num_threads(nthr)
        Bc = min(B, 2048)
        p = {k: v.clone().requires_grad_(True) for k, v in O.nm_init_params(d, L, 0).items()}
        port = O.NMTorchPort(p, L, K, not a.vanilla)
        opt = torch.optim.Adam(list(p.values()), lr=1e-3)
        xc, mc = x[:Bc].cpu(), m[:Bc].cpu()
        def cpu_step():
            if a.vanilla:
                loss = port.van_loss(xc, port.van_forward(xc, mc, torch.randn(Bc, K, L)), mc, torch.randn(Bc, K, L))
            else:
                mp = mc * (torch.rand(Bc, d) < 0.5).float()
                outs = port.reg_forward(xc, mc, mp, torch.randn(Bc, K, L), torch.randn(Bc, K, L))
                loss = port.reg_loss(xc, outs, mc, mp, alpha=0.5)
            opt.zero_grad()
            loss.backward()
            opt.step()
        cpu_step()
        n, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < 10.0:
            cpu_step()
            n += 1
        cdt = (time.perf_counter() - t1) / n
        out["cpu_baseline"] = {"value": Bc / cdt, "unit": "samples/s", "cores": nthr, "kind": "port",
                               "sample": f"{n} steps of the torch port at B={Bc} (same d, K, L), ~10 s"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
