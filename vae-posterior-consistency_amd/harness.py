"""Harness around the model classes: the pieces of the reference's L2/L3 layers that sit on the hot path.

  create_missing_uci   src/utils/utils.py:36-39      per-step Bernoulli keep-mask (device Philox instead of numpy)
  model_loader         src/utils/loaders.py:13-246   vae_type substring dispatch + checkpoint naming
  checkpoint_path      src/experiment_main/train.py:120-131
  train                src/experiment_main/train.py:13-133   epoch / batch loop, Adam(lr=1e-3), save at end
  eval_vae             src/experiment_main/evaluate.py:136-297  M MC passes: imputation RMSE on ~mask, ELBO, NLL
  eval_vae_mnar        src/experiment_main/evaluate.py:13-69    importance-weighted imputation RMSE (MNAR path)

Only the classes named by the hot path are built (Reg_VAE, vanilla_VAE and their *_mask variants,
REG_notMIWAE_v2, notMIWAE_myversion); other vae_type families raise NotImplementedError (out of scope,
SURVEY.md section 8).
"""
from __future__ import annotations

import os

import torch

from . import _lib as L
from . import ops
from .fused import FusedTrainer
from .models import Reg_VAE, Reg_VAE_mask, vanilla_VAE, vanilla_VAE_mask
from .notmiwae import NMTrainer, REG_notMIWAE_v2, notMIWAE_myversion
from .eddi import EDDITrainer, Reg_EDDI, vanilla_EDDI

_seed_counter = [0]


def create_missing_uci(shape, missing_rate, device="cuda", seed=None, offset=0):
    """Keep-mask (True = keep) with P(keep) = 1 - missing_rate/100, drawn on the device (utils.py:36-39).
    The reference draws from numpy's global RNG on the host; only the distribution is reproducible."""
    out = torch.empty(tuple(shape), dtype=torch.uint8, device=device)
    if seed is None:
        _seed_counter[0] += 1
        seed, offset = 0x5EED, _seed_counter[0] * (1 << 32)
    ops.draw_mask(None, out, 1.0 - missing_rate / 100.0, seed, offset)
    return out.view(torch.bool)


def create_missing_uci_drop_eddi(shape, device="cuda", generator=None):
    """Per-element keep-mask of the 'with_drop' variants (utils.py:42-45): keep ~ Bernoulli(1 - min(U, 0.99)) with its own U per
    element, drawn on the device as a float 0 / 1 tensor (the reference draws numpy / scipy variates on the host; only the
    distribution is reproducible - marginally P(keep) = 1 - E[min(U, .99)] = 0.50005)."""
    keep_p = 1.0 - torch.rand(tuple(shape), device=device, generator=generator).clamp_(max=0.99)
    return (torch.rand(tuple(shape), device=device, generator=generator) < keep_p).float()


def _family(vae_type: str) -> str:
    # train.py:122-124: first two '_' tokens of vae_type with the digits removed
    return "".join(ch for ch in "_".join(vae_type.split("_")[:2]) if not ch.isdigit())


def checkpoint_path(experiment_type, data_type, vae_type, missing_rate, alpha=1.0, p_missingness=30, reg_type="kl_reg"):
    """File name used by train() to save and by model_loader(stage != 'train') to load (train.py:120-131)."""
    base = os.path.join("experiments", experiment_type, data_type, "checkpoints", _family(vae_type))
    if "vanilla" in vae_type:
        return os.path.join(base, f"checkpoint_{vae_type}_{missing_rate}_missing_rate_test.pt")
    return os.path.join(base, f"checkpoint_{vae_type}_{alpha}_{p_missingness}_{reg_type}_{missing_rate}"
                              "_missing_rate_full_reg_test.pt")


def model_loader(stage, obs_dim, hid_dim, K, latent_dim, missing_rate, data_type, training_parameters, max_epochs,
                 num_samples, num_estimates, experiment_type, reg_type, vae_type="vae", alpha=1.0, p_missingness=30,
                 beta=0.5, beta_annealing=True, alpha_annealing=True, not_miwae_type="changed"):
    """Same positional signature and substring dispatch as loaders.py:13-246 for the in-scope families."""
    if "flow" in vae_type or ("MIWAE" in vae_type and "notMIWAE" not in vae_type) or \
            ("EDDI" in vae_type and data_type == "mnist"):
        raise NotImplementedError(f"vae_type {vae_type!r}: only reg_vae* / vanilla_vae* / *_notMIWAE* / *_EDDI* (UCI) "
                                  "are on the accelerated path")
    augm = "mask_augm" in vae_type  # loaders.py:47, 143
    if "reg_notMIWAE" in vae_type:  # loaders.py:89-103
        model = REG_notMIWAE_v2(obs_dim, hid_dim, K, latent_dim, training_parameters, num_samples, num_estimates)
    elif "vanilla_notMIWAE" in vae_type:  # loaders.py:219-233
        model = notMIWAE_myversion(obs_dim, hid_dim, K, latent_dim, training_parameters, num_samples, num_estimates)
    elif "reg_EDDI" in vae_type:  # loaders.py:104-131 (UCI branch)
        model = Reg_EDDI(obs_dim, hid_dim, K, latent_dim, training_parameters, experiment_type, reg_type, num_samples,
                         num_estimates)
    elif "vanilla_EDDI" in vae_type:  # loaders.py:196-218
        model = vanilla_EDDI(obs_dim, hid_dim, K, latent_dim, training_parameters, experiment_type, num_samples,
                             num_estimates)
    elif "reg_vae" in vae_type:
        model = (Reg_VAE_mask if augm else Reg_VAE)(obs_dim, hid_dim, K, latent_dim, training_parameters,
                                                    experiment_type, reg_type, num_samples, num_estimates)
    elif "vanilla_vae" in vae_type:
        model = (vanilla_VAE_mask if augm else vanilla_VAE)(obs_dim, hid_dim, K, latent_dim, training_parameters,
                                                            experiment_type, num_samples, num_estimates)
    else:
        raise NotImplementedError(f"vae_type {vae_type!r}")
    if stage == "train":
        print("Initializing fresh model")
    else:
        print("Loading saved model")
        path = checkpoint_path(experiment_type, data_type, vae_type, missing_rate, alpha, p_missingness, reg_type)
        model.load_state_dict(torch.load(path, map_location=torch.device("cpu"), weights_only=True))
    return model


def train(data_loader_train, missing_rate, obs_dim, hid_dim, K, M, latent_dim, data_type, training_parameters,
          experiment_type, vae_type, train_k, num_estimates, max_epochs=1000, device=torch.device("cuda"), alpha=1.0,
          stage="train", p_missingness=30, reg_type="ml_reg", beta=1.0, beta_annealing=False, alpha_annealing=True,
          not_miwae_type="changed", fused=True, seed=0, save=True, verbose=True):
    """train.py:13-133 for reg_vae* / vanilla_vae* / reg_notMIWAE* / vanilla_notMIWAE*.  With fused=True every batch is one FusedTrainer.step (no
    per-step host sync: the epoch total is read once per epoch, as the reference only prints it per epoch);
    with fused=False it is the reference's own sequence model.forward -> model.loss -> backward -> optim.Adam
    on the API path.  Returns the trained model."""
    model = model_loader("train", obs_dim, hid_dim, K, latent_dim, missing_rate, data_type, training_parameters,
                         max_epochs, train_k, num_estimates, experiment_type, reg_type, vae_type, alpha=alpha,
                         p_missingness=p_missingness)
    model.to(device)
    nm = "notMIWAE" in vae_type
    loader = data_loader_train if nm else data_loader_train[0]  # train.py:22-25
    is_reg = "reg" in vae_type
    eddi = "EDDI" in vae_type
    # 'with_drop' variants (train.py:32-37, 50-51; vanilla classes only - the reference's regularised branch never draws mask_p
    # beside it): the model sees mask * mask_drop, the keep-mask of create_missing_uci_drop_eddi (utils.py:42-45)
    drop = "with_drop" in vae_type and not is_reg
    if fused:
        if getattr(model, "_wide", False):  # encoder input > 128 columns: the generic-GEMM step (wide.py)
            from .wide import WideTrainer
            trainer = WideTrainer(model, lr=0.001, seed=seed)
        else:
            trainer = (NMTrainer if nm else EDDITrainer if eddi else FusedTrainer)(model, lr=0.001, seed=seed)
    else:
        model.flatten_parameters()
        optimizer = torch.optim.Adam(model.parameters(), lr=0.001)  # train.py:21
    for i in range(max_epochs):
        total_loss = 0.0
        for data_sample, mask in loader:
            data_sample = data_sample.to(device)
            mask = mask.to(device)
            if fused and drop:  # the fused vanilla step on the thinned mask (one more elementwise launch, no host sync)
                mask = mask.to(torch.float32) * create_missing_uci_drop_eddi(data_sample.shape, device=device)
            if fused and nm:
                trainer.step(data_sample, mask, alpha=alpha, p_missingness=p_missingness)
                continue
            if fused:
                trainer.step(data_sample, mask, epoch=i + 1, alpha=alpha, beta=beta, beta_annealing=beta_annealing,
                             p_missingness=p_missingness)
                continue
            if is_reg:  # train.py:53-56, 87-94
                mask_p = create_missing_uci(data_sample.shape, p_missingness, device=device) * mask
                if nm:
                    mask_p = mask_p.float()
                o = model.forward(data_sample, mask, mask_p, stage=stage)
                _, train_loss = model.loss(data_sample, o[2], o[3], o[0], o[1], o[6], o[7], o[4], o[5], mask, mask_p,
                                           i + 1, beta_annealing=beta_annealing, beta=beta, alpha=alpha,
                                           alpha_annealing=alpha_annealing, stage=stage)
            else:  # train.py:50-51, 58, 95-101
                if drop:
                    mask_drop = create_missing_uci_drop_eddi(data_sample.shape, device=device)
                else:
                    mask_drop = torch.ones(data_sample.shape, device=device)
                o = model.forward(data_sample, mask * mask_drop)
                _, train_loss = model.loss(data_sample, o[2], o[3], o[0], o[1], i + 1, mask * mask_drop,
                                           beta_annealing=beta_annealing, beta=beta, stage=stage)
            optimizer.zero_grad()
            train_loss.backward()
            optimizer.step()
            total_loss += train_loss.item()
        if fused:
            total_loss = trainer.epoch_total()
        if verbose:
            print("Epoch: [{}/{}], Total Loss: {}".format(i, max_epochs, total_loss))
    if save:
        path = checkpoint_path(experiment_type, data_type, vae_type, missing_rate, alpha, p_missingness, reg_type)
        os.makedirs(os.path.dirname(path), exist_ok=True)  # the reference never creates it
        torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, path)
    print("Training is over!")
    return model


def result_paths(experiment_type, data_type, vae_type, loader_stage, missing_rate, alpha=0.5, p_missingness=30,
                 reg_type="ml_reg"):
    """File names eval_vae writes (evaluate.py:247-297): rmse, vae_elbo, negative_llh(_q), negative_llh(_q)_imputed."""
    fam = _family(vae_type)
    rest = os.path.join("experiments", experiment_type, data_type, "rest", fam)
    elbo = os.path.join("experiments", experiment_type, data_type, "elbos", fam)
    pre = f"{loader_stage}_{vae_type}"
    if "vanilla" in vae_type:
        suf = f"_{missing_rate}_missing_rate_test.pt"
        return dict(rmse=os.path.join(rest, pre + "_rmse" + suf), elbo=os.path.join(elbo, pre + "_vae_elbo" + suf),
                    negll=os.path.join(rest, pre + "_negative_llh" + suf),
                    negll_imp=os.path.join(rest, pre + "_negative_llh_imputed" + suf))
    suf = f"_{alpha}_{p_missingness}_{reg_type}_{missing_rate}_missing_rate_full_reg_test.pt"
    return dict(rmse=os.path.join(rest, pre + "_rmse" + suf), elbo=os.path.join(elbo, pre + "_vae_elbo" + suf),
                negll=os.path.join(rest, pre + "_negative_llh_q" + suf),
                negll_imp=os.path.join(rest, pre + "_negative_llh_q_imputed" + suf))


def eval_vae(list_loaders, missing_rate, obs_dim, hid_dim, K, M, latent_dim, data_type, training_parameters,
             experiment_type, vae_type, max_epochs, valid_k, num_estimates, device=torch.device("cuda"), alpha=0.5,
             stage="evaluate", p_missingness=30, reg_type="ml_reg", beta=1.0, beta_annealing=False,
             alpha_annealing=True, model=None, save=True):
    """evaluate.py:136-297 for reg_vae* / vanilla_vae*: reload the checkpoint (or use `model`), and for every
    (loader, loader_stage): M Monte-Carlo passes of forward + loss(llh_eval=True, stage) per batch; RMSE of the
    imputations on the UNobserved entries, mean ELBO, NLL on observed and on imputed entries.  Returns
    {loader_stage: dict(rmse, elbo, negll, negll_imp)} and (save=True) writes the reference's four files."""
    out = {}
    with torch.no_grad():
        if model is None:
            model = model_loader("test", obs_dim, hid_dim, K, latent_dim, missing_rate, data_type, training_parameters,
                                 max_epochs, valid_k, num_estimates, experiment_type, reg_type, vae_type, alpha=alpha,
                                 p_missingness=p_missingness)
        model.to(device)
        opt_epoch = max_epochs
        is_reg = "reg_vae" in vae_type or "reg_EDDI" in vae_type
        for loader, loader_stage in list_loaders:
            recon, res, res_negll, res_negll_imp = [], [], [], []
            for _ in range(M):
                elbos, negls, negls_imp, temp_recon = [], [], [], []
                for data_sample, mask in loader:
                    data_sample, mask = data_sample.to(device), mask.to(device)
                    if is_reg:  # evaluate.py:172-173, 210-216
                        mask_p = create_missing_uci(data_sample.shape, p_missingness, device=device) * mask
                        o = model.forward(data_sample, mask, mask_p, stage=stage)
                        _, train_loss, negl, negl_imp = model.loss(
                            data_sample, o[2], o[3], o[0], o[1], o[6], o[7], o[4], o[5], mask, mask_p, opt_epoch,
                            llh_eval=True, beta_annealing=beta_annealing, beta=beta, alpha=alpha,
                            alpha_annealing=alpha_annealing, stage=stage)
                        x_mean = o[6]
                    else:  # evaluate.py:219-227
                        o = model.forward(data_sample, mask)
                        _, train_loss, negl, negl_imp = model.loss(data_sample, o[2], o[3], o[0], o[1], opt_epoch, mask,
                                                                   llh_eval=True, beta_annealing=beta_annealing,
                                                                   beta=beta, stage=stage)
                        x_mean = o[2]
                    mask = mask.reshape(-1, obs_dim)
                    inv = ~mask if mask.dtype == torch.bool else (mask == 0)
                    temp_recon.append(torch.sqrt(torch.sum(torch.square(
                        torch.squeeze(x_mean) * inv - data_sample.view(-1, obs_dim) * inv)) / torch.sum(inv)))
                    elbos.append(train_loss)
                    negls.append(torch.as_tensor(negl, device=device, dtype=torch.float32))
                    negls_imp.append(torch.as_tensor(negl_imp, device=device, dtype=torch.float32))
                recon.append(torch.stack(temp_recon).mean())
                res.append(torch.mean(torch.stack(elbos)))
                res_negll.append(torch.mean(torch.stack(negls)))
                res_negll_imp.append(torch.mean(torch.stack(negls_imp)))
            r = dict(rmse=torch.stack(recon).mean(), elbo=torch.stack(res).mean(), negll=torch.stack(res_negll).mean(),
                     negll_imp=torch.stack(res_negll_imp).mean())
            out[loader_stage] = {k: v.cpu() for k, v in r.items()}
            if save:
                paths = result_paths(experiment_type, data_type, vae_type, loader_stage, missing_rate, alpha,
                                     p_missingness, reg_type)
                for k, pth in paths.items():
                    os.makedirs(os.path.dirname(pth), exist_ok=True)
                    torch.save(out[loader_stage][k], pth)
    return out


def mnar_result_path(experiment_type, data_type, vae_type, alpha=0.5, p_missingness=30, reg_type="ml_reg",
                     not_miwae_type="changed"):
    """File eval_vae_mnar writes (evaluate.py:56-69)."""
    rest = os.path.join("experiments", experiment_type, data_type, "rest", "".join(c for c in vae_type if not c.isdigit()))
    if "vanilla" in vae_type:
        return os.path.join(rest, f"{vae_type}_rmse_{not_miwae_type}_large_batch_test.pt")
    return os.path.join(rest, f"{vae_type}_rmse_{alpha}_{p_missingness}_{reg_type}_full_reg_large_batch_v2_test.pt")


def eval_vae_mnar(data_test, mask_test, missing_rate, obs_dim, hid_dim, K, M, latent_dim, data_type,
                  training_parameters, experiment_type, vae_type, max_epochs, valid_k, num_estimates,
                  device=torch.device("cuda"), alpha=0.5, stage="evaluate", p_missingness=30, reg_type="ml_reg",
                  beta=1.0, beta_annealing=False, alpha_annealing=True, not_miwae_type="changed", model=None,
                  save=True, max_decoder_rows=1 << 20):
    """evaluate.py:13-69: M repetitions of the self-normalised importance-weighted imputation with valid_k samples
    per row, RMSE on the missing entries.  The reference walks the test set ONE ROW per forward (and redraws a
    full-size mask for each row); here rows are processed in chunks of max_decoder_rows / valid_k per launch
    sequence, one mask_p draw per repetition - the same estimator, row-independent, so the result has the same
    distribution.  Returns the RMSE (0-dim CPU tensor) and, with save=True, writes the reference's result file."""
    with torch.no_grad():
        if model is None:
            model = model_loader("test", obs_dim, hid_dim, K, latent_dim, missing_rate, data_type, training_parameters,
                                 max_epochs, valid_k, num_estimates, experiment_type, reg_type, vae_type, alpha=alpha,
                                 p_missingness=p_missingness, not_miwae_type=not_miwae_type)
        model.to(device)
        is_reg = "reg_notMIWAE" in vae_type
        data_test = data_test.to(device).float().reshape(-1, obs_dim)
        mask_test = mask_test.to(device).float().reshape(-1, obs_dim)
        N = data_test.shape[0]
        rows = max(1, max_decoder_rows // max(1, model.num_samples))
        temp_recon = []
        for _ in range(M):
            XM = torch.zeros_like(data_test)
            mask_p = create_missing_uci(data_test.shape, p_missingness, device=device).float() * mask_test
            for lo in range(0, N, rows):
                x, m, mp = data_test[lo:lo + rows], mask_test[lo:lo + rows], mask_p[lo:lo + rows]
                if is_reg:
                    o = model.forward(x, m, mp, stage=stage)
                    xm, _, _ = model.loss(x, o[2], o[3], o[0], o[1], o[6], o[7], o[4], o[5], m, mp, max_epochs,
                                          llh_eval=True, beta_annealing=beta_annealing, beta=beta, alpha=alpha,
                                          alpha_annealing=alpha_annealing, stage=stage)
                else:
                    o = model.forward(x, m)
                    xm, _, _ = model.loss(x, o[2], o[3], o[0], o[1], max_epochs, m, llh_eval=True,
                                          beta_annealing=beta_annealing, beta=beta, stage=stage)
                XM[lo:lo + rows] = xm
            inv = 1 - mask_test
            temp_recon.append(torch.sqrt(torch.sum(torch.square(XM * inv - data_test * inv)) / torch.sum(inv)))
        recon = torch.stack(temp_recon).mean().cpu()
        if save:
            pth = mnar_result_path(experiment_type, data_type, vae_type, alpha, p_missingness, reg_type, not_miwae_type)
            os.makedirs(os.path.dirname(pth), exist_ok=True)
            torch.save(recon, pth)
    return recon
