"""cProfile of the host side of FusedTrainer.step at a launch-bound batch size (B = 64)."""
import cProfile, pstats, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpc_amd as vpc
B, d, L = 64, 128, 10
dev = torch.device("cuda")
m = vpc.Reg_VAE(d, 500, 10, L, {"batch_size": B, "patience": 1}, "bench", "kl_reg").to(dev)
tr = vpc.FusedTrainer(m, seed=1)
x = torch.rand(B, d, device=dev); mask = torch.rand(B, d, device=dev) < 0.7
for _ in range(20): tr.step(x, mask, alpha=1.0)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): tr.step(x, mask, alpha=1.0)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
