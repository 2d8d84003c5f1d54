"""Step time of the EDDI family on the API path: model.forward -> model.loss -> backward ->
torch.optim.Adam on one MI355X, the front-end kernels timed separately, the CPU port timed beside it.

    python tools/bench_eddi.py [--batch 64] [--d 128] [--k 20] [--vanilla] [--no-cpu]
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
from vpc_amd import eddi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--latent", type=int, default=10)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--vanilla", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--api", action="store_true", help="time the API path instead of the fused EDDITrainer step")
    a = ap.parse_args()
    B, d, K, L = a.batch, a.d, a.k, a.latent
    torch.manual_seed(0)
    tp = {"batch_size": B, "patience": 1}
    model = (eddi.vanilla_EDDI(d, 500, K, L, tp, "bench") if a.vanilla else
             eddi.Reg_EDDI(d, 500, K, L, tp, "bench", "kl_reg")).cuda()
    model.flatten_parameters()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    x = torch.rand(B, d, device="cuda")
    m = torch.rand(B, d, device="cuda") < 0.7

    tr = None if a.api else eddi.EDDITrainer(model, seed=0)

    def step():
        if tr is not None:
            tr.step(x, m, epoch=1, alpha=0.5, p_missingness=30)
            return tr.out9[0]
        if a.vanilla:
            o = model.forward(x, m)
            _, tl = model.loss(x, o[2], o[3], o[0], o[1], 1, m)
        else:
            mp = vpc_amd.create_missing_uci(x.shape, 30) * m
            o = model.forward(x, m, mp, "train")
            _, tl = model.loss(x, o[2], o[3], o[0], o[1], o[6], o[7], o[4], o[5], m, mp, 1, alpha=0.5)
        opt.zero_grad()
        tl.backward()
        opt.step()
        return tl

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()  # a gen-2 collection inside the loop stalls the host for tens of ms (profiles/r01_notes.md)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        tl = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    gc.enable()
    # front-end kernels alone (HBM-bound: x fp32 + mask u8 in, agg out)
    mu8 = m.view(torch.uint8)
    AC = torch.empty(2, K, d, device="cuda")
    t = model.trainable()
    eddi.eddi_fold(t[12], t[13], t[14], t[15], AC, d, K)
    agg = torch.empty(B, K, device="cuda")
    dagg = torch.randn(B, K, device="cuda")
    g = [torch.empty_like(p) for p in t[12:]]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for _ in range(3):
        eddi.eddi_front_fwd(x, mu8, AC, agg, B, d, K)
        eddi.eddi_front_bwd(x, mu8, AC, dagg, t[12], t[13], t[14], *g, B, d, K)
    ev[0].record()
    for _ in range(20):
        eddi.eddi_front_fwd(x, mu8, AC, agg, B, d, K)
    ev[1].record()
    for _ in range(20):
        eddi.eddi_front_bwd(x, mu8, AC, dagg, t[12], t[13], t[14], *g, B, d, K)
    ev[2].record()
    torch.cuda.synchronize()
    f_ms, b_ms = ev[0].elapsed_time(ev[1]) / 20, ev[1].elapsed_time(ev[2]) / 20
    out = {"metric": "EDDI training samples/sec (" + ("API path" if a.api else "fused EDDITrainer step") + ")", "value": B / dt,
           "unit": "samples/s", "ms_per_step": dt * 1e3, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"{'vanilla_EDDI' if a.vanilla else 'Reg_EDDI kl_reg'} B={B} d={d} K={K} L={L}"},
           "loss": float(tl),
           "front_end": {"fwd_ms": f_ms, "bwd_ms": b_ms, "fwd_GBps": B * d * 5 / f_ms / 1e6,
                         "bytes_model": "x fp32 + mask u8 per (row, feature); agg / dagg B*K*4"}}
    if not a.no_cpu:
        from oracle import eddi_oracle as O
        nthr = min(16, os.cpu_count() or 1)
        torch.set_num_threads(nthr)
        Bc = min(B, 4096)
        sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items() if k in O.EDDI_KEYS}
        port = O.EDDIPort(sd, L, "kl_reg")
        copt = torch.optim.Adam(list(sd.values()), lr=1e-3)
        xc, mc = x[:Bc].cpu(), m[:Bc].cpu()

        def cstep():
            if a.vanilla:
                mf = mc.float()
                o = port.vanilla_forward(xc, mf)
                _, l = port.vanilla_loss(xc, o[2], o[3], o[0], o[1], 1, mf)
            else:
                mp = mc & (torch.rand(Bc, d) < 0.7)
                o = port.reg_forward(xc, mc, mp)
                _, l = port.reg_loss(xc, o[2], o[3], o[0], o[1], o[6], o[7], o[4], o[5], mc, mp, 1, alpha=0.5)
            copt.zero_grad()
            l.backward()
            copt.step()
        cstep()
        n, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < 8.0:
            cstep()
            n += 1
        cdt = (time.perf_counter() - t1) / n
        out["cpu_baseline"] = {"value": Bc / cdt, "unit": "samples/s", "cores": nthr, "kind": "port",
                               "sample": f"{n} steps of the torch port at B={Bc}, ~8 s"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
