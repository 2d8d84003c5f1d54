"""Golden vectors for the MNAR path, produced by running the REFERENCE itself.

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_notmiwae.py

Authoring container only: imports /root/reference (never copied, never shipped) and stores DATA only.

  nm_reg_d{14,40,128}.npz  REG_notMIWAE_v2 (VAE.py:2327-2505): state_dict, inputs (x, float masks), the eps the two
                           rsample() calls drew, the 8 forward outputs, loss for alpha in {1.0, 0.5, 0.0} with all
                           parameter grads, the llh_eval branch (xm, RE_q.mean())
  nm_van_d{14,40,128}.npz  notMIWAE_myversion (VAE.py:2691-2847): same, incl. the fresh eps drawn inside loss()
  nm_traj_{reg,van}_d14.npz  5 Adam steps exactly as train.py:87-117 runs them
  nm_eval_{reg,van}_d14.npz  (--eval) the reference's eval_vae_mnar (evaluate.py:13-69) on a checkpoint in its own
                           naming scheme: parameters, test rows and the RMSE it wrote
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

from src.models.VAE import REG_notMIWAE_v2, notMIWAE_myversion  # noqa: E402

TP = {"batch_size": 128, "patience": 100}


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
    mask = (torch.rand(B, d, generator=g) < 0.7).float()
    mask_p = mask * (torch.rand(B, d, generator=g) < 0.5).float()
    return x, mask, mask_p


def gen_reg(d, L, K, B, seed, alphas=(1.0, 0.5, 0.0)):
    torch.manual_seed(seed)
    model = REG_notMIWAE_v2(d, 500, 10, L, TP, K, 1)
    x, mask, mask_p = make_inputs(B, d, seed + 1)
    out = {"param." + k: v for k, v in sd_np(model).items()}
    out.update(x=x.numpy(), mask=mask.numpy(), mask_p=mask_p.numpy(), K=np.int64(K), L=np.int64(L))
    eps_q, eps_p = peek_normals([(B, K, L), (B, K, L)])
    names = ["mean_p", "logvar_p", "x_mean_p", "x_logvar_p", "mean_q", "logvar_q", "x_mean_q", "x_logvar_q"]
    for alpha in alphas:
        st = torch.get_rng_state()
        model.zero_grad()
        outs = model.forward(x, mask, mask_p, "train")
        mean_p, logvar_p, x_mean_p, x_logvar_p, mean_q, logvar_q, x_mean_q, x_logvar_q = outs
        pl, tl = model.loss(x, x_mean_p, x_logvar_p, mean_p, logvar_p, x_mean_q, x_logvar_q, mean_q, logvar_q, mask,
                            mask_p, 7, alpha=alpha)
        tl.backward()
        out[f"loss.a{alpha}"] = np.float64(tl.item())
        out.update(grads_np(model, f"a{alpha}"))
        torch.set_rng_state(st)
    for n, t in zip(names, outs):
        out["fwd." + n] = t.detach().numpy()
    out.update(eps_q=eps_q.numpy(), eps_p=eps_p.numpy())
    with torch.no_grad():
        xm, tl, re = model.loss(x, x_mean_p, x_logvar_p, mean_p, logvar_p, x_mean_q, x_logvar_q, mean_q, logvar_q,
                                mask, mask_p, 7, alpha=0.5, llh_eval=True)
    out.update(llh_xm=xm.numpy(), llh_loss=np.float64(tl.item()), llh_re=np.float64(re.item()))
    np.savez_compressed(os.path.join(OUT, f"nm_reg_d{d}.npz"), **out)
    print("nm_reg", d, {k: float(v) for k, v in out.items() if k.startswith("loss.")})


def gen_van(d, L, K, B, seed):
    torch.manual_seed(seed)
    model = notMIWAE_myversion(d, 500, 10, L, TP, K, 1)
    x, mask, _ = make_inputs(B, d, seed + 1)
    out = {"param." + k: v for k, v in sd_np(model).items()}
    out.update(x=x.numpy(), mask=mask.numpy(), K=np.int64(K), L=np.int64(L))
    eps_q, eps_kl = peek_normals([(B, K, L), (B, K, L)])
    model.zero_grad()
    mean, logvar, x_mean, x_logvar = model.forward(x, mask)
    pl, tl = model.loss(x, x_mean, x_logvar, mean, logvar, 3, mask)
    tl.backward()
    out["loss"] = np.float64(tl.item())
    out.update(grads_np(model, "v"))
    for n, t in zip(["mean", "logvar", "x_mean", "x_logvar"], (mean, logvar, x_mean, x_logvar)):
        out["fwd." + n] = t.detach().numpy()
    out.update(eps_q=eps_q.numpy(), eps_kl=eps_kl.numpy())
    (eps_llh,) = peek_normals([(B, K, L)])
    with torch.no_grad():
        xm, tl2, re = model.loss(x, x_mean, x_logvar, mean, logvar, 3, mask, llh_eval=True)
    out.update(llh_xm=xm.numpy(), llh_loss=np.float64(tl2.item()), llh_re=np.float64(re.item()),
               eps_llh=eps_llh.numpy())
    np.savez_compressed(os.path.join(OUT, f"nm_van_d{d}.npz"), **out)
    print("nm_van", d, out["loss"])


def gen_traj(kind, d=14, L=10, K=20, B=16, steps=5, seed=4242):
    torch.manual_seed(seed)
    model = (REG_notMIWAE_v2 if kind == "reg" else notMIWAE_myversion)(d, 500, 10, L, TP, K, 1)
    opt = torch.optim.Adam(model.parameters(), lr=0.001)  # train.py:21
    x, mask, _ = make_inputs(B, d, seed + 1)
    out = {"param0." + k: v for k, v in sd_np(model).items()}
    out.update(x=x.numpy(), mask=mask.numpy(), K=np.int64(K), L=np.int64(L))
    g = torch.Generator().manual_seed(seed + 2)
    losses, eps_all, mp_all = [], [], []
    for s in range(steps):
        if kind == "reg":
            mask_p = mask * (torch.rand(B, d, generator=g) < 0.5).float()
            mp_all.append(mask_p.numpy())
            eps = peek_normals([(B, K, L), (B, K, L)])
            o = model.forward(x, mask, mask_p, stage="train")
            _, tl = model.loss(x, o[2], o[3], o[0], o[1], o[6], o[7], o[4], o[5], mask, mask_p, s + 1,
                               beta_annealing=False, beta=1.0, alpha=0.5, alpha_annealing=True, stage="train")
        else:
            eps = peek_normals([(B, K, L), (B, K, L)])
            o = model.forward(x, mask * torch.ones_like(mask))
            _, tl = model.loss(x, o[2], o[3], o[0], o[1], s + 1, mask * torch.ones_like(mask),
                               beta_annealing=False, beta=1.0, stage="train")
        eps_all.append(np.stack([e.numpy() for e in eps]))
        opt.zero_grad()
        tl.backward()
        opt.step()
        losses.append(tl.item())
    out.update({"param5." + k: v for k, v in sd_np(model).items()})
    out.update(losses=np.array(losses, dtype=np.float64), eps=np.stack(eps_all))
    if mp_all:
        out["mask_p"] = np.stack(mp_all)
    np.savez_compressed(os.path.join(OUT, f"nm_traj_{kind}_d{d}.npz"), **out)
    print("nm_traj", kind, losses)


def gen_eval_mnar(kind, d=14, L=10, N=24, valid_k=4000, M=3, seed=606):
    """The reference's own eval_vae_mnar (evaluate.py:13-69) on a checkpoint written in its naming scheme: pins
    the checkpoint interop and the importance-weighted imputation RMSE (an MC estimate: compare within noise)."""
    import tempfile
    from src.experiment_main.evaluate import eval_vae_mnar
    torch.manual_seed(seed)
    np.random.seed(seed)
    vae_type = "reg_notMIWAE1" if kind == "reg" else "vanilla_notMIWAE1"
    model = (REG_notMIWAE_v2 if kind == "reg" else notMIWAE_myversion)(d, 500, 10, L, TP, 20, 1)
    opt = torch.optim.Adam(model.parameters(), lr=0.003)
    x, mask, _ = make_inputs(N, d, seed + 1)
    for s in range(60):  # a few steps so that the imputations are not trivial
        if kind == "reg":
            mp = mask * (torch.rand(N, d) < 0.5).float()
            o = model.forward(x, mask, mp)
            _, tl = model.loss(x, o[2], o[3], o[0], o[1], o[6], o[7], o[4], o[5], mask, mp, s + 1, alpha=0.5)
        else:
            o = model.forward(x, mask)
            _, tl = model.loss(x, o[2], o[3], o[0], o[1], s + 1, mask)
        opt.zero_grad(); tl.backward(); opt.step()
    out = {"param." + k: v for k, v in sd_np(model).items()}
    fam = "".join(c for c in vae_type if not c.isdigit())
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            for sub in ("checkpoints", "rest"):
                os.makedirs(os.path.join("experiments", "exp", "toy", sub, fam))
            if kind == "reg":
                ck = f"experiments/exp/toy/checkpoints/{fam}/checkpoint_{vae_type}_0.5_50_kl_reg_50_missing_rate_full_reg_test.pt"
            else:
                ck = f"experiments/exp/toy/checkpoints/{fam}/checkpoint_{vae_type}_50_missing_rate_test.pt"
            torch.save(model.state_dict(), ck)
            eval_vae_mnar(x, mask, 50, d, 500, 10, M, L, "toy", TP, "exp", vae_type, 100, valid_k, 1, alpha=0.5,
                          p_missingness=50, reg_type="kl_reg")
            files = os.listdir(f"experiments/exp/toy/rest/{fam}")
            assert len(files) == 1, files
            rmse = torch.load(os.path.join(f"experiments/exp/toy/rest/{fam}", files[0]))
            out["result_file"] = np.array(files[0])
        finally:
            os.chdir(cwd)
    out.update(x=x.numpy(), mask=mask.numpy(), rmse=np.float64(rmse.item()), valid_k=np.int64(valid_k), M=np.int64(M),
               L=np.int64(L), K=np.int64(20))
    np.savez_compressed(os.path.join(OUT, f"nm_eval_{kind}_d{d}.npz"), **out)
    print("nm_eval", kind, rmse.item(), files[0])


if __name__ == "__main__" and "--eval" in sys.argv:
    gen_eval_mnar("reg")
    gen_eval_mnar("van")
    sys.exit(0)

if __name__ == "__main__":
    gen_reg(14, 10, 20, 16, 31)
    gen_reg(40, 6, 5, 24, 32)
    gen_van(14, 10, 20, 16, 41)
    gen_van(40, 6, 5, 24, 42)
    gen_traj("reg")
    gen_traj("van")
    # config 3's own model shape (UCI gas, d = 128, K = train_k = 20, imputation_args_mnar.json), B kept small: the
    # fixture stores [B, K, d] outputs and every gradient
    gen_reg(128, 10, 20, 8, 33, alphas=(0.5,))
    gen_van(128, 10, 20, 8, 43)
