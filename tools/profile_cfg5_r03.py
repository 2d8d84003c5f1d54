"""One acquisition step of active_learning_func (config 5: Reg_VAE d = 128, n_test = 256, M = 50) in a loop, for rocprofv3:
    rocprofv3 --kernel-trace --stats -d out -o cfg5 -- python3 tools/profile_cfg5_r03.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpc_amd as vpc  # noqa: E402
from vpc_amd import active as act  # noqa: E402

d, Ld, n, M = 128, 10, 256, 50
torch.manual_seed(0)
model = vpc.Reg_VAE(d, 500, 10, Ld, {"batch_size": 64, "patience": 100}, "bench", "kl_reg").cuda()
x = torch.rand(n, d, device="cuda")
tm = (torch.rand(n, d, device="cuda") < 0.9)
cur = (torch.rand(n, d, generator=torch.Generator().manual_seed(5)) < 0.3).float().cuda()
cur[:, -1] = 0
with torch.no_grad():
    def step():
        im = act.mc_forward(model, x, cur, tm, M)
        R = act.reward_matrix(model, x, cur, im)
        return R.argmax(1)
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        step()
    torch.cuda.synchronize()
    print("ms per acquisition step: %.3f" % ((time.perf_counter() - t0) * 10))
