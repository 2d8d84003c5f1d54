"""Golden vectors for the PNP / EDDI encoder family, produced by running the REFERENCE.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_eddi.py

Authoring container only: imports /root/reference (never copied, never shipped) and stores DATA only.

  eddi_reg_d{14,40}.npz   Reg_EDDI (VAE.py:670-853): state_dict, inputs (x, bool masks), eps of the two rsample()
                          calls, the 8 forward outputs, loss (kl_reg alpha in {0.5, 1.0}; ml_reg + its eps) with all
                          parameter grads, the evaluate / llh_eval branch
  eddi_van_d{14,40}.npz   vanilla_EDDI (VAE.py:856-992): same (float mask as train.py:58,97 passes it)
  eddi_traj_{reg,van}_d14.npz  5 Adam steps exactly as train.py:87-117 runs them
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
tv = types.ModuleType("torchvision")
tv.datasets = types.ModuleType("torchvision.datasets")
tv.transforms = types.ModuleType("torchvision.transforms")
sys.modules["torchvision"] = tv
sys.modules["torchvision.datasets"] = tv.datasets
sys.modules["torchvision.transforms"] = tv.transforms

from src.models.VAE import Reg_EDDI, vanilla_EDDI  # noqa: E402

TP = {"batch_size": 64, "patience": 100}


def peek_normals(shapes):
    st = torch.get_rng_state()
    eps = [torch.empty(s).normal_() for s in shapes]
    torch.set_rng_state(st)
    return eps


def sd_np(model):
    return {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}


def grads_np(model, tag):
    return {f"grad.{tag}.{k}": p.grad.detach().numpy().copy() for k, p in model.named_parameters()
            if p.grad is not None}


def make_inputs(B, d, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, d, generator=g)
    mask = torch.rand(B, d, generator=g) < 0.7
    mask_p = mask & (torch.rand(B, d, generator=g) < 0.7)
    return x, mask, mask_p


def gen_reg(d, K, L, B, seed):
    torch.manual_seed(seed)
    model = Reg_EDDI(d, 500, K, L, TP, "exp", "kl_reg")
    x, mask, mask_p = make_inputs(B, d, seed + 1)
    out = {"param." + k: v for k, v in sd_np(model).items()}
    out.update(x=x.numpy(), mask=mask.numpy(), mask_p=mask_p.numpy(), K=np.int64(K), L=np.int64(L))
    eps_q, eps_p, eps_ml = peek_normals([(B, L), (B, L), (B, L)])
    names = ["mean_p", "logvar_p", "x_mean_p", "x_logvar_p", "mean_q", "logvar_q", "x_mean_q", "x_logvar_q"]
    for tag, reg_type, alpha in (("kl0.5", "kl_reg", 0.5), ("kl1.0", "kl_reg", 1.0), ("ml0.8", "ml_reg", 0.8)):
        st = torch.get_rng_state()
        model.reg_type = reg_type
        model.zero_grad()
        outs = model.forward(x, mask, mask_p, "train")
        o = outs
        pl, tl = model.loss(x, o[2], o[3], o[0], o[1], o[6], o[7], o[4], o[5], mask, mask_p, 1400, beta=0.9,
                            alpha=alpha, beta_annealing=(tag == "kl1.0"))
        tl.backward()
        out[f"loss.{tag}"] = np.float64(tl.item())
        out.update(grads_np(model, tag))
        torch.set_rng_state(st)
    model.reg_type = "kl_reg"
    for n, t in zip(names, outs):
        out["fwd." + n] = t.detach().numpy()
    out.update(eps_q=eps_q.numpy(), eps_p=eps_p.numpy(), eps_ml=eps_ml.numpy())
    with torch.no_grad():
        r = model.loss(x, o[2], o[3], o[0], o[1], o[6], o[7], o[4], o[5], mask, mask_p, 7, llh_eval=True,
                       stage="evaluate")
    out.update(eval_loss=np.float64(r[1].item()), eval_re=np.float64(r[2].item()), eval_re_imp=np.float64(r[3].item()))
    np.savez_compressed(os.path.join(OUT, f"eddi_reg_d{d}.npz"), **out)
    print("eddi_reg", d, {k: float(v) for k, v in out.items() if k.startswith("loss.")})


def gen_van(d, K, L, B, seed):
    torch.manual_seed(seed)
    model = vanilla_EDDI(d, 500, K, L, TP, "exp")
    x, mask, _ = make_inputs(B, d, seed + 1)
    maskf = mask * torch.ones(mask.shape)  # train.py:58,97
    out = {"param." + k: v for k, v in sd_np(model).items()}
    out.update(x=x.numpy(), mask=mask.numpy(), K=np.int64(K), L=np.int64(L))
    (eps_q,) = peek_normals([(B, L)])
    model.zero_grad()
    o = model.forward(x, maskf)
    r = model.loss(x, o[2], o[3], o[0], o[1], 3, maskf, beta=0.8, llh_eval=True)
    r[1].backward()
    out.update(loss=np.float64(r[1].item()), re=np.float64(r[2].item()), re_imp=np.float64(r[3].item()))
    out.update(grads_np(model, "v"))
    for n, t in zip(["mean", "logvar", "x_mean", "x_logvar"], o):
        out["fwd." + n] = t.detach().numpy()
    out.update(eps_q=eps_q.numpy())
    np.savez_compressed(os.path.join(OUT, f"eddi_van_d{d}.npz"), **out)
    print("eddi_van", d, out["loss"])


def gen_traj(kind, d=14, K=10, L=10, B=32, steps=5, seed=909):
    torch.manual_seed(seed)
    model = Reg_EDDI(d, 500, K, L, TP, "exp", "kl_reg") if kind == "reg" else vanilla_EDDI(d, 500, K, L, TP, "exp")
    opt = torch.optim.Adam(model.parameters(), lr=0.001)
    x, mask, _ = make_inputs(B, d, seed + 1)
    out = {"param0." + k: v for k, v in sd_np(model).items()}
    out.update(x=x.numpy(), mask=mask.numpy(), K=np.int64(K), L=np.int64(L))
    g = torch.Generator().manual_seed(seed + 2)
    losses, eps_all, mp_all = [], [], []
    for s in range(steps):
        if kind == "reg":
            mask_p = mask & (torch.rand(B, d, generator=g) < 0.7)
            mp_all.append(mask_p.numpy())
            eps = peek_normals([(B, L), (B, L)])
            o = model.forward(x, mask, mask_p, stage="train")
            _, tl = model.loss(x, o[2], o[3], o[0], o[1], o[6], o[7], o[4], o[5], mask, mask_p, s + 1,
                               beta_annealing=False, beta=1.0, alpha=0.5, alpha_annealing=True, stage="train")
        else:
            eps = peek_normals([(B, L)])
            mf = mask * torch.ones(mask.shape)
            o = model.forward(x, mf)
            _, tl = model.loss(x, o[2], o[3], o[0], o[1], s + 1, mf, beta_annealing=False, beta=1.0, stage="train")
        eps_all.append(np.stack([e.numpy() for e in eps]))
        opt.zero_grad()
        tl.backward()
        opt.step()
        losses.append(tl.item())
    out.update({"param5." + k: v for k, v in sd_np(model).items()})
    out.update(losses=np.array(losses, dtype=np.float64), eps=np.stack(eps_all))
    if mp_all:
        out["mask_p"] = np.stack(mp_all)
    np.savez_compressed(os.path.join(OUT, f"eddi_traj_{kind}_d{d}.npz"), **out)
    print("eddi_traj", kind, losses)


if __name__ == "__main__":
    gen_reg(14, 10, 10, 32, 71)
    gen_reg(40, 20, 6, 48, 72)
    gen_van(14, 20, 10, 32, 81)
    gen_van(40, 10, 6, 48, 82)
    gen_traj("reg")
    gen_traj("van")
