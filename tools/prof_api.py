import cProfile, pstats, sys, os, torch
sys.path.insert(0, os.getcwd())
import vpc_amd as vpc
B, d, L = 64, 128, 10
dev = torch.device("cuda")
m = vpc.Reg_VAE(d, 500, 10, L, {"batch_size": B, "patience": 1}, "bench", "kl_reg").to(dev)
m.flatten_parameters()
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
x = torch.rand(B, d, device=dev); mask = torch.rand(B, d, device=dev) < 0.7
def step():
    mask_p = vpc.create_missing_uci(x.shape, 30, device=dev) * mask
    o = m.forward(x, mask, mask_p, stage="train")
    _, tl = m.loss(x, o[2], o[3], o[0], o[1], o[6], o[7], o[4], o[5], mask, mask_p, 1, beta=1.0, alpha=1.0)
    opt.zero_grad(); tl.backward(); opt.step()
for _ in range(10): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(100): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
