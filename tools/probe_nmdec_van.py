import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_nmdec as T
from vpc_amd import notmiwae as nm
def run(B, K, L, reg, seed=5):
    p, x, m, mp, eps = T._problem(128, L, K, B, seed)
    cls = nm.REG_notMIWAE_v2 if reg else nm.notMIWAE_myversion
    model = cls(128, 128, 10, L, {"batch_size": B, "patience": 1}, K, 1)
    model.load_state_dict({k: v.float() for k, v in p.items()}, strict=False)
    model = model.cuda()
    tr = nm.NMTrainer(model, precision="bf16")
    tr.step(x.cuda(), m.cuda(), mask_p=mp.cuda() if reg else None, eps=eps.cuda(), alpha=0.5 if reg else 0.0, p_missingness=50)
    ref, gref = T._oracle_step(p, x, m, mp, eps, L, K, 0.5 if reg else 0.0, tr.use_nmdec, reg)
    errs = {k: T.rel(prm.grad.cpu().numpy(), gref[k]) for k, prm in model.named_parameters() if k in gref}
    w = max(errs, key=errs.get)
    print(f"B={B} K={K} reg={reg} seed={seed} loss_rel={abs(tr.loss_value()-ref)/abs(ref):.2e} worst={w} {errs[w]:.2e}")
for B in (36, 37, 38, 39, 64, 128):
    run(B, 20, 10, False)
for s in (6, 7, 8):
    run(37, 20, 10, False, s)
run(37, 20, 10, True)
