import os, sys, torch
sys.path.insert(0, '.')
from vpc_amd import notmiwae as nm
d, L, K, B = 128, 10, 20, 256
g = torch.Generator().manual_seed(9)
X = torch.rand(4096, d, generator=g).cuda(); M = (torch.rand(4096, d, generator=g) < 0.7).float().cuda()
res = {}
for form in ("fused", "gemm"):
    if form == "gemm": os.environ["VPC_NMDEC"] = "0"
    torch.manual_seed(3)
    model = nm.REG_notMIWAE_v2(d, 128, 10, L, {"batch_size": B, "patience": 1}, K, 1).cuda()
    tr = nm.NMTrainer(model, precision="bf16", seed=11)
    curve = []
    for ep in range(25):
        for i in range(0, 4096, B):
            tr.step(X[i:i + B], M[i:i + B], alpha=0.5, p_missingness=50)
        curve.append(tr.epoch_total() / 16)
    res[form] = curve
    print(form, tr.use_nmdec, ["%.3f" % c for c in curve[::4]])
print("max rel diff of epoch means:", max(abs(a - b) / abs(b) for a, b in zip(res["fused"], res["gemm"])))
