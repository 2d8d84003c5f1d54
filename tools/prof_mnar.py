"""Host-side profile (cProfile) of the fused MNAR step at the reference's batch size: where the Python time of a step goes."""
import cProfile, pstats, sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpc_amd  # noqa
from vpc_amd import notmiwae as nm
B, d, K = 128, 128, 20
dev = torch.device("cuda")
model = nm.REG_notMIWAE_v2(d, 128, 10, 10, {"batch_size": B, "patience": 1}, K, 1).to(dev)
tr = nm.NMTrainer(model)
x = torch.rand(B, d, device=dev); m = (torch.rand(B, d, device=dev) < 0.5).float()
for _ in range(50): tr.step(x, m, alpha=0.5, p_missingness=50)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): tr.step(x, m, alpha=0.5, p_missingness=50)
t1 = time.perf_counter()  # host time to ENQUEUE 300 steps
torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"enqueue {1e6 * (t1 - t0) / 300:.1f} us/step, complete {1e6 * (t2 - t0) / 300:.1f} us/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(200): tr.step(x, m, alpha=0.5, p_missingness=50)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
