import sys, time, torch
sys.path.insert(0, '.')
import vpc_amd
from vpc_amd import eddi
d, B = 128, 64
torch.manual_seed(0)
model = eddi.vanilla_EDDI(d, 500, 10, 10, {"batch_size": B, "patience": 1}, "exp").cuda()
tr = eddi.EDDITrainer(model, lr=1e-3)
x = torch.rand(B, d, device="cuda"); m = (torch.rand(B, d, device="cuda") < 0.7)
def step():
    mk = m.to(torch.float32) * vpc_amd.create_missing_uci_drop_eddi(x.shape, device="cuda")
    tr.step(x, mk, epoch=1)
for _ in range(50): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(300): step()
torch.cuda.synchronize(); print("with_drop fused EDDI step B=64 d=128: %.1f us" % ((time.perf_counter() - t0) / 300 * 1e6))
