"""us / step of the fused step by batch size and precision (bf16 variants exist in the throughput shape only)."""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpc_amd as vpc
dev = torch.device("cuda:0")
for B in (64, 1024, 8192, 65536):
    for prec in ("f32", "bf16x3", "bf16"):
        torch.manual_seed(0)
        m = vpc.Reg_VAE(128, 500, 10, 10, {"batch_size": B, "patience": 1}, "b", "kl_reg").to(dev)
        tr = vpc.FusedTrainer(m, seed=1, precision=prec)
        x = torch.rand(B, 128, device=dev); mask = torch.rand(B, 128, device=dev) < 0.7
        for _ in range(80): tr.step(x, mask, alpha=1.0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(300): tr.step(x, mask, alpha=1.0)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 300 * 1e6
        tr.timers, tr.timer_every = {}, 1
        for _ in range(20): tr.step(x, mask, alpha=1.0)
        torch.cuda.synchronize()
        import statistics
        k = {n: round(statistics.median(a.elapsed_time(b) for a, b in v) * 1e3, 1) for n, v in tr.timers.items()}
        print(json.dumps(dict(B=B, precision=prec, us_per_step=round(us, 1), kernels_us=k, loss=round(tr.loss_value(), 3))), flush=True)
