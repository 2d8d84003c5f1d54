"""Phase stamps of the whole-step bf16 kernel (diagnostic build: python vae-posterior-consistency_amd/csrc/build.py --ablate,
then VPC_LIB=.../libvpc_hip_ablate.so VPC_DEBUG=64 python tools/stamps_step.py [B])."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vpc_amd as vpc  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = vpc.Reg_VAE(128, 500, 10, 10, {"batch_size": B, "patience": 100}, "bench", "kl_reg").to(dev)
tr = vpc.FusedTrainer(m, precision=prec)
x = torch.rand(B, 128, device=dev)
mk = torch.rand(B, 128, device=dev) < 0.7
for i in range(3):
    tr.step(x, mk, alpha=1.0)
torch.cuda.synchronize()
