"""Active variable selection (BASELINE config 5): the information reward of
src/experiment_main/evaluate.py:514-634 on the GPU.

`reward_matrix` replaces the whole candidate loop of active_learning_func (evaluate.py:424-433) by three kernel
launches (vpc_reward_matrix); `R_lindley_chain`, `chaini_I`, `chaini_II` keep the reference's signatures so that
evaluate.py can call them unchanged.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib as L
from ._lib import check, lib, ptr, stream_ptr
from .ops import _f32c, as_mask_u8


def reward_matrix(vae, x, mask, im):
    """R [n, d-1]: reward of revealing feature u for row n (-1e4 where already observed).
    x [n, d]; mask [n, d] (bool / float 0-1 / uint8); im [M, n, d] MC imputations; target = last column."""
    L.require_cuda(x, im)
    if vae.mask_augm:
        raise NotImplementedError("reward_matrix: mask-augmented encoders are not supported")
    n, d = x.shape
    M = im.shape[0]
    lay = vae._lay()
    dev = x.device
    vae._images()
    w1, b1 = vae.trainable()[0], vae.trainable()[1]
    sizes = [C.c_long() for _ in range(3)]
    check(lib().vpc_reward_scratch(n, d, M, *[C.byref(s) for s in sizes]), "vpc_reward_scratch")
    pre = torch.empty(sizes[0].value, device=dev)
    stat = torch.empty(sizes[1].value, device=dev)
    w1t = torch.empty(sizes[2].value, device=dev)
    R = torch.empty(n, d - 1, device=dev)
    check(lib().vpc_reward_matrix(ptr(_f32c(x)), ptr(as_mask_u8(mask.to(dev))), ptr(_f32c(im)), ptr(w1.data), ptr(b1.data),
                                  ptr(vae._enc_img()), ptr(pre), ptr(stat), ptr(w1t), ptr(R), n, d, lay.L, M,
                                  stream_ptr()), "vpc_reward_matrix")
    return R


def R_lindley_chain(i, x, mask, M, vae, im, loc):
    """Same signature / result as evaluate.py:514-542 (rows `loc`, candidate `i`).  Prefer reward_matrix: it
    returns every candidate of every row for the price of this one call."""
    loc_t = torch.as_tensor(loc, device=x.device, dtype=torch.long)
    R = reward_matrix(vae, x[loc_t], mask[loc_t], im[:M][:, loc_t])
    return R[:, i]


def _kl(mean, logvar, mean_i, logvar_i):
    # evaluate.py:582-583: first term divided by v = exp(logvar / 2), as in the reference
    return 0.5 * torch.sum(torch.square(mean_i - mean) / torch.exp(logvar / 2) + torch.exp(logvar_i - logvar) - 1.0
                           - logvar_i + logvar, 1)


def chaini_I(x, mask, i, vae):
    """evaluate.py:546-586 on the encoder kernels (API path)."""
    tm = mask.clone()
    with torch.no_grad():
        _, mean, logvar = vae.encoder(x, tm, sample=False)
        tm[:, i] = 1
        _, mean_i, logvar_i = vae.encoder(x, tm, sample=False)
    return _kl(mean, logvar, mean_i, logvar_i)


def chaini_II(x, mask, i, vae):
    """evaluate.py:590-634."""
    tm = mask.clone()
    tm[:, -1] = 1
    with torch.no_grad():
        _, mean, logvar = vae.encoder(x, tm, sample=False)
        tm[:, i] = 1
        _, mean_i, logvar_i = vae.encoder(x, tm, sample=False)
    return _kl(mean, logvar, mean_i, logvar_i)


# ------------------------------------------------------------------------------------------------ the acquisition loop
def active_result_paths(experiment_type, data_type, vae_type, missing_rate, alpha=1.0, p_missingness=30, reg_type="ml_reg"):
    """The four files active_learning_func writes (evaluate.py:457-511), reference naming."""
    fam = "".join(c for c in "_".join(vae_type.split("_")[:2]) if not c.isdigit())
    rest = os.path.join("experiments", experiment_type, data_type, "rest", fam)
    if "vanilla" in vae_type:
        mk = lambda key, sep: os.path.join(rest, f"{vae_type}_{missing_rate}_missing_rate{sep}UCI_{key}_default_test.pt")  # noqa: E731
        return dict(information_curve_CHAI=mk("information_curve_CHAI", "_"), action_CHAI=mk("action_CHAI", "__"),
                    R_hist_CHAI=mk("R_hist_CHAI", "__"), im_CHAI=mk("im_CHAI", "__"))
    suf = f"_{alpha}_{p_missingness}_{reg_type}_{missing_rate}_missing_rate_default_full_reg_test.pt"
    return {k: os.path.join(rest, f"{vae_type}_UCI_{k}{suf}")
            for k in ("information_curve_CHAI", "action_CHAI", "R_hist_CHAI", "im_CHAI")}


def mc_forward(model, x, mask, mask_p, M, stage="evaluate"):
    """x_mean_q of M independent forward passes as ONE batched pass over M stacked copies of the rows (the eps of
    Normal.rsample are i.i.d. per row, so M calls of model.forward on n rows == one call on M n rows): [M, n, d]."""
    n, d = x.shape
    xr, mr = x.repeat(M, 1), mask.repeat(M, 1)
    if hasattr(model, "reg_type"):  # Reg_VAE family: forward(data, mask, mask_p, stage) -> (..p.., mean_q, logvar_q, x_mean_q, ..)
        x_mean = model.forward(xr, mr, mask_p.repeat(M, 1), stage)[6]
    else:
        x_mean = model.forward(xr, mr)[2]
    return x_mean.reshape(M, n, d)


def active_learning_func(data_loader_train, test_data, test_mask, missing_rate, obs_dim, hid_dim, K, M, latent_dim,
                         data_type, training_parameters, experiment_type, vae_type, max_epochs, valid_k, num_estimates,
                         device=None, alpha=1.0, stage="evaluate", p_missingness=30, reg_type="ml_reg", beta=1.0,
                         beta_annealing=False, alpha_annealing=True, Repeat=5, model=None, save=True, verbose=False,
                         _forward=None, max_steps=None):
    """Active variable selection, src/experiment_main/evaluate.py:300-511 (same positional signature; `model`, `save`,
    `verbose`, `_forward`, `max_steps` (stop after that many acquisitions; the rest of the outputs stays zero) are additions).  Per repeat: all features start unobserved (the target - last column - stays
    unobserved throughout); at each of the obs_dim - 1 steps M Monte-Carlo forward passes impute the rows (`im`), the
    information reward of revealing each candidate feature is evaluated for every row - ONE vpc_reward_matrix call instead
    of the reference's (obs_dim - 1) R_lindley_chain calls with 4 M encoder passes each - the best candidate per row is
    revealed (argmax, evaluate.py:435-440), and the target MSE of M further passes goes to the information curve.
    The M passes run as one batched forward (mc_forward).  `_forward(mask) -> x_mean_q [n, d]` replaces a single forward
    pass (tests replay the outputs recorded from the reference, whose eps come from the global RNG).
    Returns dict(information_curve_CHAI [Repeat, n, d], action_CHAI [Repeat, n, d-1], R_hist_CHAI [Repeat, d-1, n, d-1],
    im_CHAI [Repeat, d-1, M, n, d]) and (save=True) writes the reference's four files."""
    from .harness import create_missing_uci, model_loader
    dev = torch.device(device) if device is not None else torch.device("cuda")
    n_test, d = test_data.shape[0], obs_dim
    info = torch.zeros(Repeat, n_test, d)
    action = torch.zeros(Repeat, n_test, d - 1)
    R_hist = torch.zeros(Repeat, d - 1, n_test, d - 1)
    im_hist = torch.zeros(Repeat, d - 1, M, n_test, d)
    x = test_data.reshape(-1, d).float().to(dev)
    tmask = test_mask.to(dev)
    eye = torch.eye(d, device=dev)
    with torch.no_grad():
        for r in range(Repeat):
            if model is None or r > 0:
                mdl = model_loader("test", obs_dim, hid_dim, K, latent_dim, missing_rate, data_type, training_parameters,
                                   max_epochs, valid_k, num_estimates, experiment_type, reg_type, vae_type, alpha=alpha,
                                   p_missingness=p_missingness, alpha_annealing=alpha_annealing).to(dev)
            else:
                mdl = model.to(dev)
            mask_p = tmask * create_missing_uci(tuple(test_data.shape), p_missingness, device=dev)  # evaluate.py:349-350
            mask = torch.zeros(n_test, d, device=dev)

            def passes(cur_mask):
                if _forward is not None:
                    return torch.stack([_forward(cur_mask).to(dev) for _ in range(M)], 0)
                return mc_forward(mdl, x, cur_mask, mask_p, M, stage)

            def target_mse(xm):  # mean over the M passes of F.mse_loss on the target column (evaluate.py:390-392)
                return ((xm[:, :, -1] - x[None, :, -1]) ** 2).mean(1).mean()

            info[r, :, 0] = target_mse(passes(mask)).cpu()
            for t in range(d - 1 if max_steps is None else min(max_steps, d - 1)):
                if verbose:
                    print("Repeat = {:.1f}".format(r)); print("Strategy = {:.1f}".format(2)); print("Step = {:.1f}".format(t))
                im = passes(mask)
                R = reward_matrix(mdl, x, mask, im)
                i_opt = R.argmax(1)
                mask = mask + eye[i_opt]
                info[r, :, t + 1] = target_mse(passes(mask)).cpu()
                action[r, :, t] = i_opt.cpu().float()
                R_hist[r, t] = R.cpu()
                im_hist[r, t] = im.cpu()
    out = dict(information_curve_CHAI=info, action_CHAI=action, R_hist_CHAI=R_hist, im_CHAI=im_hist)
    if save:
        paths = active_result_paths(experiment_type, data_type, vae_type, missing_rate, alpha, p_missingness, reg_type)
        for k, pth in paths.items():
            os.makedirs(os.path.dirname(pth), exist_ok=True)
            torch.save(out[k], pth)
    return out
