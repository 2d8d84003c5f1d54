"""Golden vectors for the active-variable-selection loop (BASELINE config 5), produced
by running the REFERENCE's own `active_learning_func` (src/experiment_main/evaluate.py:300-511) end to end.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_active.py

Authoring container only: imports /root/reference (never copied, never shipped) and stores DATA only.

The loop draws eps inside every model.forward() from torch's global RNG (M calls before each reward evaluation, M after
each acquisition, and more inside R_lindley_chain), so its trajectory cannot be replayed from a seed by another
implementation.  What makes it comparable is the forward outputs themselves: this script wraps the loaded model's
forward (a Python attribute of the instance - the reference source is not modified) and records the x_mean_q of every
call in order.  active_{d}.npz then holds
    param.*            the checkpoint the reference loaded (written in its own naming scheme)
    x, test_mask       the test rows
    fwd_xmean          [n_calls][n][d]   x_mean_q of every forward call, in call order
    im                 [steps][M][n][d]  im_CHAI (the MC imputations each reward step consumed)
    R_hist             [steps][n][d-1]   R_hist_CHAI
    action             [n][steps]        action_CHAI (chosen feature per row and step)
    info_curve         [steps+1]         information_curve_CHAI[0, 0, :] (target MSE before / after each acquisition)
    files              names of the four result files it wrote
"""
import os
import sys
import tempfile
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

from src.models.VAE import Reg_VAE, vanilla_VAE  # noqa: E402
import src.experiment_main.evaluate as EV  # noqa: E402

TP = {"batch_size": 64, "patience": 100}


def gen(kind, d=14, n=24, M=5, L=10, seed=909):
    torch.manual_seed(seed)
    np.random.seed(seed)
    vae_type = "reg_vae1" if kind == "reg" else "vanilla_vae1"
    model = Reg_VAE(d, 500, 10, L, TP, "exp", "kl_reg") if kind == "reg" else vanilla_VAE(d, 500, 10, L, TP, "exp")
    g = torch.Generator().manual_seed(seed + 1)
    # correlated columns, so that revealing features carries information about the target (last column)
    base = torch.rand(n + 256, 3, generator=g)
    mix = torch.rand(3, d, generator=g)
    data = torch.sigmoid(3.0 * (base @ mix / mix.sum(0) - 0.5)) + 0.05 * torch.rand(n + 256, d, generator=g)
    data = (data - data.min(0).values) / (data.max(0).values - data.min(0).values)
    xtr, x = data[n:], data[:n].clone()
    mtr = torch.rand(256, d, generator=g) < 0.7
    opt = torch.optim.Adam(model.parameters(), lr=3e-3)
    for s in range(150):  # a short training run: rewards and imputations are not those of a random network
        if kind == "reg":
            mp = mtr & (torch.rand(256, d) < 0.7)
            o = model.forward(xtr, mtr, mp, "train")
            _, tl = model.loss(xtr, o[2], o[3], o[0], o[1], o[6], o[7], o[4], o[5], mtr, mp, s + 1, alpha=1.0)
        else:
            mf = mtr * torch.ones(256, d)
            o = model.forward(xtr, mf)
            _, tl = model.loss(xtr, o[2], o[3], o[0], o[1], s + 1, mf)
        opt.zero_grad(); tl.backward(); opt.step()
    test_mask = torch.rand(n, d, generator=g) < 0.7
    out = {"param." + k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    calls = []
    orig_loader = EV.model_loader

    def loader(*a, **kw):
        m = orig_loader(*a, **kw)
        fwd = m.forward

        def rec(*fa, **fk):
            r = fwd(*fa, **fk)
            calls.append((r[6] if kind == "reg" else r[2]).detach().numpy().copy())  # x_mean_q
            return r

        m.forward = rec
        return m

    fam = "".join(c for c in vae_type if not c.isdigit())
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            for sub in ("checkpoints", "rest"):
                os.makedirs(os.path.join("experiments", "exp", "toy", sub, fam))
            if kind == "reg":
                ck = f"experiments/exp/toy/checkpoints/{fam}/checkpoint_{vae_type}_1.0_30_kl_reg_30_missing_rate_full_reg_test.pt"
            else:
                ck = f"experiments/exp/toy/checkpoints/{fam}/checkpoint_{vae_type}_30_missing_rate_test.pt"
            torch.save(model.state_dict(), ck)
            EV.model_loader = loader
            torch.manual_seed(seed + 7)
            EV.active_learning_func(None, x, test_mask, 30, d, 500, 10, M, L, "toy", TP, "exp", vae_type, 100, 1, 1,
                                    alpha=1.0, p_missingness=30, reg_type="kl_reg", Repeat=1)
            rest = f"experiments/exp/toy/rest/{fam}"
            files = sorted(os.listdir(rest))
            res = {}
            for f in files:
                key = [k for k in ("information_curve_CHAI", "action_CHAI", "R_hist_CHAI", "im_CHAI") if k in f][0]
                res[key] = torch.load(os.path.join(rest, f))
        finally:
            EV.model_loader = orig_loader
            os.chdir(cwd)
    steps = d - 1
    assert len(calls) == M + steps * 2 * M
    out.update(x=x.numpy(), test_mask=test_mask.numpy(), fwd_xmean=np.stack(calls).astype(np.float32),
               im=res["im_CHAI"][0].numpy().astype(np.float32), R_hist=res["R_hist_CHAI"][0].numpy(),
               action=res["action_CHAI"][0].numpy(), info_curve=res["information_curve_CHAI"][0, 0].numpy(),
               files=np.array(files), M=np.int64(M), L=np.int64(L))
    np.savez_compressed(os.path.join(OUT, f"active_{kind}_d{d}.npz"), **out)
    print("active", kind, "info curve", out["info_curve"][:4], "...", out["info_curve"][-1], "first actions", out["action"][:3, :4])


if __name__ == "__main__":
    gen("reg")
    gen("van")
