"""Phase stamps (diagnostic build only: VPC_LIB=.../libvpc_hip_ablate.so VPC_DEBUG=64 VPC_DEBUG_ENC=64) of one fused step."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vpc_amd as vpc
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
PREC = sys.argv[2] if len(sys.argv) > 2 else "f32"
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = vpc.Reg_VAE(128, 500, 10, 10, {"batch_size": B, "patience": 1}, "bench", "kl_reg").to(dev)
tr = vpc.FusedTrainer(m, seed=1, precision=PREC)
x = torch.rand(B, 128, device=dev); mask = torch.rand(B, 128, device=dev) < 0.7
for i in range(3):
    tr.step(x, mask, alpha=1.0)
    torch.cuda.synchronize()
    print("---- step", i, flush=True)
