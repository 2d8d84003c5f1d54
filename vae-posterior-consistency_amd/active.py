"""Active variable selection (BASELINE config 5): the information reward of
src/experiment_main/evaluate.py:514-634 on the GPU.

`reward_matrix` replaces the whole candidate loop of active_learning_func (evaluate.py:424-433) by three kernel
launches (vpc_reward_matrix); `R_lindley_chain`, `chaini_I`, `chaini_II` keep the reference's signatures so that
evaluate.py can call them unchanged.
"""
from __future__ import annotations

import ctypes as C

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
