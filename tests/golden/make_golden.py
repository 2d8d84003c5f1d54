"""Generate golden vectors by running the REFERENCE itself (authoring container only).

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Imports /root/reference (never copied into this repo, never shipped to the GPU
box) and stores only DATA - inputs and the reference's outputs - as .npz under
tests/golden/.  `torchvision` is absent in this image and is imported (never
used) by src/utils/loaders.py:10, so an empty stub module is registered first.

What is pinned:
  reg_d{14,128}.npz      Reg_VAE forward (8 outputs), loss for a grid of
                         (reg_type, alpha, beta, beta_annealing, epoch, stage,
                         llh_eval, MI), and all 12 parameter grads for 3 configs
  vanilla_d{14,128}.npz  vanilla_VAE forward / loss / grads
  traj_reg_d14.npz       5 optimiser steps (model.forward/loss + optim.Adam as
  traj_vanilla_d14.npz   train.py:87-117 does), eps / mask_p recorded
  train_e2e_d14.npz      the reference's own train() run end to end (2 epochs,
                         3 batches) with numpy / torch RNG seeded
eps are recorded exactly: the RNG state is saved, the same normal_() draws
Normal.rsample() makes are taken, and the state is restored before the call.
"""
import os
import sys
import types
import tempfile

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

from src.models.VAE import Reg_VAE, vanilla_VAE  # noqa: E402

TP = {"batch_size": 64, "patience": 100}
L = 10


def peek_normals(shapes):
    """Return the eps tensors the next rsample() calls will draw, leaving the RNG untouched."""
    st = torch.get_rng_state()
    eps = [torch.empty(s).normal_() for s in shapes]
    torch.set_rng_state(st)
    return eps


def sd_np(model):
    return {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}


def grads_np(model):
    return {"grad." + k: p.grad.detach().numpy().copy() for k, p in model.named_parameters() if p.grad is not None}


def make_inputs(B, d, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, d, generator=g)
    mask = torch.rand(B, d, generator=g) < 0.7
    mask_p = mask & (torch.rand(B, d, generator=g) < 0.7)
    return x, mask, mask_p


def gen_reg(d, B=64, seed=1234):
    torch.manual_seed(seed)
    model = Reg_VAE(d, 500, 10, L, TP, "exp", "kl_reg")
    x, mask, mask_p = make_inputs(B, d, seed + 1)
    out = {"param." + k: v for k, v in sd_np(model).items()}
    out.update(x=x.numpy(), mask=mask.numpy(), mask_p=mask_p.numpy())
    eps_q, eps_p = peek_normals([(B, L), (B, L)])
    outs = model.forward(x, mask, mask_p, "train")
    mean_p, logvar_p, x_mean_p, x_logvar_p, mean_q, logvar_q, x_mean_q, x_logvar_q = outs
    # sanity: injected-eps identity (rsample == loc + eps * scale)
    assert torch.equal(model.encoder(x, mask, sample=False)[1], mean_q)
    out.update(eps_q=eps_q.numpy(), eps_p=eps_p.numpy(),
               mean_q=mean_q.detach().numpy(), logvar_q=logvar_q.detach().numpy(),
               x_mean_q=x_mean_q.detach().numpy(), mean_p=mean_p.detach().numpy(),
               logvar_p=logvar_p.detach().numpy(), x_mean_p=x_mean_p.detach().numpy(),
               x_logvar=x_logvar_q.detach().numpy())
    args = (x, x_mean_p, x_logvar_p, mean_p, logvar_p, x_mean_q, x_logvar_q, mean_q, logvar_q, mask, mask_p)
    # ---- loss grid (train stage, kl_reg)
    rows, vals = [], []
    for alpha in (1.0, 0.5, 0.8):
        for beta in (1.0, 0.7):
            for ann in (False, True):
                for epoch in (1, 1400):
                    _, tl = model.loss(*args, epoch, beta_annealing=ann, beta=beta, alpha=alpha, stage="train")
                    rows.append([alpha, beta, float(ann), epoch])
                    vals.append(tl.item())
    out["klreg_grid_cfg"] = np.array(rows, np.float64)
    out["klreg_grid_loss"] = np.array(vals, np.float64)
    # ---- evaluate stage extras
    pl, tl, re_q, re_imp = model.loss(*args, 1, llh_eval=True, beta=1.0, alpha=1.0, stage="evaluate")
    out["eval_llh"] = np.array([tl.item(), re_q.item(), re_imp.item()], np.float64)
    pl, tl, mi, klq = model.loss(*args, 1, MI=True, beta=1.0, alpha=1.0, stage="evaluate")
    out["eval_MI"] = np.array([tl.item(), mi.item(), klq.item()], np.float64)
    pl, tl, re_q, re_imp = model.loss(*args, 7, llh_eval=True, beta=0.7, alpha=0.5, stage="train",
                                      beta_annealing=True)
    out["train_llh"] = np.array([tl.item(), re_q.item(), float(re_imp)], np.float64)
    # ---- grads, kl_reg
    for tag, kw in (("a1", dict(alpha=1.0, beta=1.0, beta_annealing=False, epoch=1)),
                    ("a05", dict(alpha=0.5, beta=0.7, beta_annealing=True, epoch=1400))):
        model.zero_grad()
        # fresh graph with the SAME eps: the model's own encoder/decoder with sample=False plus
        # eps * std (== Normal.rsample() by definition); checked bit-equal to forward() above
        _, mq, lq = model.encoder(x, mask, sample=False)
        xq, xlv = model.decoder(mq + eps_q * torch.exp(lq / 2))
        _, mp, lp = model.encoder(x, mask_p, sample=False)
        xp, _ = model.decoder(mp + eps_p * torch.exp(lp / 2))
        assert torch.equal(xq, x_mean_q) and torch.equal(xp, x_mean_p)
        ep = kw.pop("epoch")
        _, tl = model.loss(x, xp, xlv, mp, lp, xq, xlv, mq, lq, mask, mask_p, ep, stage="train", **kw)
        tl.backward()
        out["loss_" + tag] = np.array(tl.item(), np.float64)
        for k, v in grads_np(model).items():
            out[tag + "." + k] = v
    # ---- ml_reg (extra rsample inside loss, VAE.py:435-440)
    model.reg_type = "ml_reg"
    model.zero_grad()
    _, mq, lq = model.encoder(x, mask, sample=False)
    xq, xlv = model.decoder(mq + eps_q * torch.exp(lq / 2))
    _, mp, lp = model.encoder(x, mask_p, sample=False)
    xp, _ = model.decoder(mp + eps_p * torch.exp(lp / 2))
    (eps_ml,) = peek_normals([(B, L)])
    _, tl = model.loss(x, xp, xlv, mp, lp, xq, xlv, mq, lq, mask, mask_p, 1400, beta=1.0, alpha=0.8,
                       stage="train")
    tl.backward()
    out["eps_ml"] = eps_ml.numpy()
    out["loss_ml"] = np.array(tl.item(), np.float64)
    for k, v in grads_np(model).items():
        out["ml." + k] = v
    model.reg_type = "kl_reg"
    np.savez_compressed(os.path.join(OUT, f"reg_d{d}.npz"), **out)
    print("reg", d, "loss a1", out["loss_a1"], "ml", out["loss_ml"])


def gen_vanilla(d, B=64, seed=4321):
    torch.manual_seed(seed)
    model = vanilla_VAE(d, 500, 10, L, TP, "exp")
    x, mask, _ = make_inputs(B, d, seed + 1)
    out = {"param." + k: v for k, v in sd_np(model).items()}
    out.update(x=x.numpy(), mask=mask.numpy())
    (eps_q,) = peek_normals([(B, L)])
    mean_q, logvar_q, x_mean_q, x_logvar_q = model.forward(x, mask)
    out.update(eps_q=eps_q.numpy(), mean_q=mean_q.detach().numpy(), logvar_q=logvar_q.detach().numpy(),
               x_mean_q=x_mean_q.detach().numpy())
    mask_f = mask * torch.ones(x.shape)  # train.py:58,97
    rows, vals = [], []
    for beta in (1.0, 0.7):
        for ann in (False, True):
            for epoch in (1, 1400):
                _, tl = model.loss(x, x_mean_q, x_logvar_q, mean_q, logvar_q, epoch, mask_f,
                                   beta_annealing=ann, beta=beta, stage="train")
                rows.append([beta, float(ann), epoch])
                vals.append(tl.item())
    out["grid_cfg"] = np.array(rows, np.float64)
    out["grid_loss"] = np.array(vals, np.float64)
    pl, tl, re_q, re_imp = model.loss(x, x_mean_q, x_logvar_q, mean_q, logvar_q, 1, mask_f, llh_eval=True,
                                      stage="evaluate")
    out["eval_llh"] = np.array([tl.item(), re_q.item(), re_imp.item()], np.float64)
    model.zero_grad()
    _, tl = model.loss(x, x_mean_q, x_logvar_q, mean_q, logvar_q, 1, mask_f, stage="train")
    tl.backward()
    out["loss_b1"] = np.array(tl.item(), np.float64)
    for k, v in grads_np(model).items():
        out["b1." + k] = v
    np.savez_compressed(os.path.join(OUT, f"vanilla_d{d}.npz"), **out)
    print("vanilla", d, "loss", out["loss_b1"])


def gen_traj(kind, d=14, B=64, steps=5, seed=777):
    torch.manual_seed(seed)
    if kind == "reg":
        model = Reg_VAE(d, 500, 10, L, TP, "exp", "kl_reg")
    else:
        model = vanilla_VAE(d, 500, 10, L, TP, "exp")
    opt = torch.optim.Adam(model.parameters(), lr=0.001)  # train.py:21
    out = {"param0." + k: v for k, v in sd_np(model).items()}
    g = torch.Generator().manual_seed(seed + 5)
    x = torch.rand(B, d, generator=g)
    mask = torch.rand(B, d, generator=g) < 0.7
    out.update(x=x.numpy(), mask=mask.numpy())
    losses, mps, eqs, eps_ = [], [], [], []
    for i in range(steps):
        if kind == "reg":
            mask_p = mask & (torch.rand(B, d, generator=g) < 0.7)
            eq, ep = peek_normals([(B, L), (B, L)])
            o = model.forward(x, mask, mask_p, stage="train")
            _, tl = model.loss(x, o[2], o[3], o[0], o[1], o[6], o[7], o[4], o[5], mask, mask_p, i + 1,
                               beta_annealing=False, beta=1.0, alpha=1.0, alpha_annealing=True, stage="train")
            mps.append(mask_p.numpy())
            eps_.append(ep.numpy())
        else:
            mask_f = mask * torch.ones(x.shape)
            (eq,) = peek_normals([(B, L)])
            o = model.forward(x, mask_f)
            _, tl = model.loss(x, o[2], o[3], o[0], o[1], i + 1, mask_f, beta_annealing=False, beta=1.0,
                               stage="train")
        eqs.append(eq.numpy())
        opt.zero_grad()
        tl.backward()
        opt.step()
        losses.append(tl.item())
    out["loss"] = np.array(losses, np.float64)
    out["eps_q"] = np.stack(eqs)
    if kind == "reg":
        out["eps_p"] = np.stack(eps_)
        out["mask_p"] = np.stack(mps)
    out.update({"paramT." + k: v for k, v in sd_np(model).items()})
    np.savez_compressed(os.path.join(OUT, f"traj_{kind}_d{d}.npz"), **out)
    print("traj", kind, losses)


def gen_train_e2e(d=14, N=96, B=32, epochs=2, seed=99):
    """Run the reference's train() itself (train.py:13-133) on a synthetic DataLoader."""
    from src.experiment_main.train import train
    from src.utils.loaders import ConcatDataset
    from torch.utils.data import DataLoader
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(N, d, generator=g)
    mask = torch.rand(N, d, generator=g) < 0.7
    loader = DataLoader(ConcatDataset(x, mask), batch_size=B, shuffle=False)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            os.makedirs("experiments/exp/synth/checkpoints/reg_vae", exist_ok=True)
            torch.manual_seed(seed)
            np.random.seed(seed)
            init = Reg_VAE(d, 500, 10, L, {"batch_size": B, "patience": 100}, "exp", "kl_reg")
            init_sd = sd_np(init)
            torch.manual_seed(seed)  # train() builds its model first thing -> identical init
            np.random.seed(seed)
            train((loader, None), 30, d, 500, 10, 1, L, "synth", {"batch_size": B, "patience": 100}, "exp",
                  "reg_vae1", 20, 10, epochs, torch.device("cpu"), alpha=1.0, p_missingness=30,
                  reg_type="kl_reg")
            path = "experiments/exp/synth/checkpoints/reg_vae/checkpoint_reg_vae1_1.0_30_kl_reg_30_missing_rate_full_reg_test.pt"
            final = torch.load(path, weights_only=True)
        finally:
            os.chdir(cwd)
    out = {"param0." + k: v for k, v in init_sd.items()}
    out.update({"paramT." + k: v.numpy() for k, v in final.items()})
    out.update(x=x.numpy(), mask=mask.numpy(), meta=np.array([N, B, epochs, seed, 30], np.int64))
    out["ckpt_relpath"] = np.array(path)
    np.savez_compressed(os.path.join(OUT, f"train_e2e_d{d}.npz"), **out)
    print("train e2e done")


if __name__ == "__main__" and "--reward" not in sys.argv and "--mask-augm" not in sys.argv:
    for d in (14, 128):
        gen_reg(d)
        gen_vanilla(d)
    gen_traj("reg")
    gen_traj("vanilla")
    gen_train_e2e()


def gen_reward(d=14, n=24, M=5, seed=2024):
    """R_lindley_chain / chaini_I / chaini_II of the reference (evaluate.py:514-634) on a Reg_VAE, float 0/1 masks
    (as active_learning.py passes them), one case with the target column partly observed (exercises the
    temp_x[loc,-1] carry-over between MC samples)."""
    from src.experiment_main.evaluate import R_lindley_chain, chaini_I, chaini_II
    torch.manual_seed(seed)
    model = Reg_VAE(d, 500, 10, L, TP, "exp", "kl_reg")
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.rand(n, d, generator=g)
    out = {"param." + k: v for k, v in sd_np(model).items()}
    im = torch.rand(M, n, d, generator=g)
    for tag, target_obs in (("t0", 0.0), ("t1", 0.5)):
        mask = (torch.rand(n, d, generator=g) < 0.6).float()
        mask[:, -1] = (torch.rand(n, generator=g) < target_obs).float()
        R = -1e4 * torch.ones(n, d - 1)
        with torch.no_grad():
            for u in range(d - 1):
                loc = np.where(mask[:, u] == 0)[0]
                if len(loc):
                    R[loc, u] = R_lindley_chain(u, x, mask, M, model, im, loc).float()
            k1 = chaini_I(x, mask, 3, model)
            k2 = chaini_II(x, mask, 3, model)
        out.update({f"mask_{tag}": mask.numpy(), f"R_{tag}": R.numpy(), f"kl1_{tag}": k1.numpy(), f"kl2_{tag}": k2.numpy()})
    out.update(x=x.numpy(), im=im.numpy())
    np.savez_compressed(os.path.join(OUT, f"reward_d{d}.npz"), **out)
    print("reward golden done", float(out["R_t0"].max()))


if __name__ == "__main__" and "--reward" in sys.argv:
    gen_reward()
    gen_reward(d=40, n=20, M=19, seed=77)


def gen_mask_augm(d=14, B=48, seed=555):
    """Reg_VAE_mask / vanilla_VAE_mask (VAE.py:510-667, 995-1116): forward, one loss, all grads."""
    from src.models.VAE import Reg_VAE_mask, vanilla_VAE_mask
    out = {}
    for kind, cls in (("reg", Reg_VAE_mask), ("vanilla", vanilla_VAE_mask)):
        torch.manual_seed(seed)
        model = cls(d, 500, 10, L, TP, "exp", "kl_reg") if kind == "reg" else cls(d, 500, 10, L, TP, "exp")
        x, mask, mask_p = make_inputs(B, d, seed + 1)
        out.update({f"{kind}.param." + k: v for k, v in sd_np(model).items()})
        if kind == "reg":
            eps_q, eps_p = peek_normals([(B, L), (B, L)])
            o = model.forward(x, mask, mask_p, "train")
            _, tl = model.loss(x, o[2], o[3], o[0], o[1], o[6], o[7], o[4], o[5], mask, mask_p, 1, beta=0.9, alpha=0.7,
                               stage="train")
            out.update({"reg.eps_p": eps_p.numpy(), "reg.mean_p": o[0].detach().numpy(), "reg.x_mean_p": o[2].detach().numpy(),
                        "reg.mean_q": o[4].detach().numpy(), "reg.logvar_q": o[5].detach().numpy(),
                        "reg.x_mean_q": o[6].detach().numpy()})
        else:
            (eps_q,) = peek_normals([(B, L)])
            mf = mask * torch.ones(x.shape)
            o = model.forward(x, mf)
            _, tl = model.loss(x, o[2], o[3], o[0], o[1], 1, mf, beta=0.9, stage="train")
            out.update({"vanilla.mean_q": o[0].detach().numpy(), "vanilla.x_mean_q": o[2].detach().numpy()})
        tl.backward()
        out[f"{kind}.eps_q"] = eps_q.numpy()
        out[f"{kind}.loss"] = np.array(tl.item(), np.float64)
        for k, v in grads_np(model).items():
            out[f"{kind}." + k] = v
    out.update(x=x.numpy(), mask=mask.numpy(), mask_p=mask_p.numpy())
    np.savez_compressed(os.path.join(OUT, f"maskaugm_d{d}.npz"), **out)
    print("mask_augm golden done", out["reg.loss"], out["vanilla.loss"])


if __name__ == "__main__" and "--mask-augm" in sys.argv:
    gen_mask_augm()
