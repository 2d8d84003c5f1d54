"""Config 5 (active variable selection): one acquisition step's reward matrix R[n, d-1] at n_test=256, d=128, M=50
(Data/imputation_args.json: M=50) - GPU (vpc_reward_matrix) vs the oracle's op-for-op CPU port of the
reference loop (evaluate.py:424-433, 514-634) on a bounded sample of candidates."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpc_amd as vpc
from oracle import vae_oracle as O

n, d, M, L = 256, 128, 50, 10
params = O.init_params(d, L, seed=0)
g = torch.Generator().manual_seed(0)
x = torch.rand(n, d, generator=g); mask = (torch.rand(n, d, generator=g) < 0.5).float(); mask[:, -1] = 0
im = torch.rand(M, n, d, generator=g)
m = vpc.Reg_VAE(d, 500, 10, L, {"batch_size": n, "patience": 1}, "b", "kl_reg")
sd = m.state_dict(); sd.update({k: v.clone() for k, v in params.items()}); m.load_state_dict(sd); m.cuda()
xd, md, imd = x.cuda(), mask.cuda(), im.cuda()
for _ in range(3): R = vpc.reward_matrix(m, xd, md, imd)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): R = vpc.reward_matrix(m, xd, md, imd)
torch.cuda.synchronize(); gpu_ms = (time.perf_counter() - t0) / 20 * 1e3
# CPU port on a bounded sample: 4 of the 127 candidates, scaled
torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
port = O.TorchPort(params, L)
import numpy as np
t0 = time.perf_counter()
with torch.no_grad():
    for u in (0, 31, 77, 126):
        loc = np.where(mask[:, u].numpy() == 0)[0]
        O.R_lindley_chain(port, u, x, mask, M, im, loc)
cpu_ms = (time.perf_counter() - t0) / 4 * (d - 1) * 1e3
evals = n * M * (2 + 2 * (d - 1)) * 0.5  # ~half of the candidates are unobserved
print(json.dumps({"workload": f"reward matrix n={n} d={d} M={M} (one acquisition step)", "gpu_ms": gpu_ms,
                  "cpu_port_ms_extrapolated_from_4_candidates": cpu_ms, "cpu_threads": torch.get_num_threads(),
                  "speedup": cpu_ms / gpu_ms, "encoder_evaluations_gpu": evals}))
