import sys, os, time, torch
sys.path.insert(0, os.getcwd())
import vpc_amd as vpc
from vpc_amd import ops
B, d, L = 64, 128, 10
dev = torch.device("cuda")
m = vpc.Reg_VAE(d, 500, 10, L, {"batch_size": B, "patience": 1}, "bench", "kl_reg").to(dev)
tr = vpc.FusedTrainer(m, seed=1)
x = torch.rand(B, d, device=dev); mask = torch.rand(B, d, device=dev) < 0.7
for tile in ("64", "128"):
    os.environ["VPC_TILE"] = tile
    for _ in range(10): tr.step(x, mask, alpha=1.0)
    torch.cuda.synchronize()
    import types
    names = ["draw_step", "encoder_fwd", "decoder_fused", "encoder_bwd", "reduce_step_adam"]
    acc = {n: 0.0 for n in names}
    orig = {n: getattr(ops, n) for n in names}
    def wrap(n):
        f = orig[n]
        def g(*a, **k):
            t0 = time.perf_counter(); r = f(*a, **k); acc[n] += time.perf_counter() - t0; return r
        return g
    for n in names: setattr(ops, n, wrap(n))
    N = 300
    t0 = time.perf_counter()
    for _ in range(N): tr.step(x, mask, alpha=1.0)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    for n in names: setattr(ops, n, orig[n])
    print("tile", tile, "host us/step", round(t_host / N * 1e6, 1), "incl sync", round(t_all / N * 1e6, 1), {n: round(v / N * 1e6, 1) for n, v in acc.items()})
