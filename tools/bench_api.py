"""API-path timings (model.forward / model.loss / backward / torch Adam) and the stand-alone fused loss
kernel K4 (vpc_loss_fwd_bwd) at B=65536, d=128: ms per step, K4 HBM GB/s against the 8 TB/s peak.
Algorithmic bytes of K4: 3 216 B/sample = read x, xhat_q, xhat_p (12d) + masks (2d) + 4x4L,
write dxhat_q, dxhat_p (8d) + 4x4L."""
import json, os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpc_amd as vpc

B, d, L = (int(sys.argv[1]) if len(sys.argv) > 1 else 65536), 128, 10
dev = torch.device("cuda")
torch.manual_seed(0)
m = vpc.Reg_VAE(d, 500, 10, L, {"batch_size": B, "patience": 1}, "bench", "kl_reg").to(dev)
m.flatten_parameters()
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
x = torch.rand(B, d, device=dev)
mask = torch.rand(B, d, device=dev) < 0.7

def step():
    mask_p = vpc.create_missing_uci(x.shape, 30, device=dev) * mask
    o = m.forward(x, mask, mask_p, stage="train")
    _, tl = m.loss(x, o[2], o[3], o[0], o[1], o[6], o[7], o[4], o[5], mask, mask_p, 1, beta=1.0, alpha=1.0)
    opt.zero_grad(); tl.backward(); opt.step()
    return tl

for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): step()
torch.cuda.synchronize(); api_ms = (time.perf_counter() - t0) / 30 * 1e3

# K4 alone
with torch.no_grad():
    mask_p = vpc.create_missing_uci(x.shape, 30, device=dev) * mask
    o = m.forward(x, mask, mask_p, stage="train")
xq = o[6].detach().requires_grad_(True); xp = o[2].detach().requires_grad_(True)
mq, lq, mp, lp = [t.detach().requires_grad_(True) for t in (o[4], o[5], o[0], o[1])]
ev = []
for i in range(25):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _, tl = m.loss(x, xp, o[3], mp, lp, xq, o[7], mq, lq, mask, mask_p, 1, beta=1.0, alpha=1.0)
    e1.record(); ev.append((e0, e1))
torch.cuda.synchronize()
k4_ms = statistics.median(a.elapsed_time(b) for a, b in ev[5:])
alg = 3216 * B
print(json.dumps({"B": B, "api_path_ms_per_step": api_ms, "api_path_samples_per_s": B / api_ms * 1e3,
                  "k4_loss_call_ms": k4_ms, "k4_algorithmic_bytes": alg,
                  "k4_GBps_incl_host_glue": alg / (k4_ms * 1e-3) / 1e9, "hbm_peak_GBps": 8000}))
