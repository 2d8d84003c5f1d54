"""CPU oracle for the PNP / EDDI encoder family: Reg_EDDI / vanilla_EDDI.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (only tests/ and bench baselines may import it).

``EDDIPort`` restates the reference classes ``Reg_EDDI`` (src/models/VAE.py:670-853) and ``vanilla_EDDI``
(:856-992) on stock PyTorch CPU: the point-net front-end (per feature j: ``relu(W [x_j, x_j E_j, b_j] + c)``,
mask-weighted sum over the features, VAE.py:719-733) followed by pnp_encoder2 (K->100->50->2L); decoder and loss are
those of Reg_VAE / vanilla_VAE, inherited from ``vae_oracle.TorchPort``.  ``front_closed_form`` is an independent
float64 numpy statement of the front-end in the folded form the HIP kernel uses (A_j = w_x + W_E E_j,
C_j = w_b b_j + c) with hand-derived gradients.

Pinned by tests/test_eddi_oracle.py against tests/golden/eddi_*.npz (tests/golden/make_golden_eddi.py runs the
reference itself).
"""
from __future__ import annotations

import numpy as np
import torch

from .vae_oracle import TorchPort

EDDI_KEYS = (
    "type_pars1", "type_bias1",
    "pnp_encoder1.0.weight", "pnp_encoder1.0.bias",
    "pnp_encoder2.0.weight", "pnp_encoder2.0.bias", "pnp_encoder2.2.weight", "pnp_encoder2.2.bias",
    "pnp_encoder2.4.weight", "pnp_encoder2.4.bias",
    "seq_decoder.0.weight", "seq_decoder.0.bias", "seq_decoder.2.weight", "seq_decoder.2.bias",
    "seq_decoder.4.weight", "seq_decoder.4.bias",
)


class EDDIPort(TorchPort):
    def __init__(self, params, latent_dim, reg_type="kl_reg"):
        super().__init__(params, latent_dim, reg_type)

    def front(self, x, mask):  # VAE.py:719-733
        p = self.p
        B, d = x.shape
        E, tb = p["type_pars1"], p["type_bias1"]
        xf = x.reshape(-1, 1)
        feat = torch.cat([xf, xf * E.repeat(B, 1), tb.repeat(B, 1)], 1).to(E.dtype)
        h = torch.relu(torch.nn.functional.linear(feat, p["pnp_encoder1.0.weight"], p["pnp_encoder1.0.bias"]))
        K = E.shape[1]
        return (mask.reshape(B, d, 1).to(E.dtype) * h.reshape(B, d, K)).sum(1)

    def encoder(self, x, mask, eps=None, sample=True):
        p = self.p
        h = self.front(x, mask)
        h = torch.relu(torch.nn.functional.linear(h, p["pnp_encoder2.0.weight"], p["pnp_encoder2.0.bias"]))
        h = torch.relu(torch.nn.functional.linear(h, p["pnp_encoder2.2.weight"], p["pnp_encoder2.2.bias"]))
        h = torch.nn.functional.linear(h, p["pnp_encoder2.4.weight"], p["pnp_encoder2.4.bias"])
        mean, logvar = h.chunk(2, dim=1)
        if not sample:
            return mean, mean, logvar
        std = torch.exp(logvar / 2)
        z = mean + (torch.randn_like(std) if eps is None else eps) * std
        return z, mean, logvar


    def vanilla_loss(self, *a, **kw):
        # vanilla_EDDI.loss (VAE.py:935-950) computes RE_q_imputed in EVERY stage; otherwise it is vanilla_VAE.loss
        kw["stage"] = "evaluate"
        return super().vanilla_loss(*a, **kw)


def front_closed_form(x, mask, E, tb, Wp, cp, dagg=None):
    """agg [B,K] (and, given dagg, the gradients of E, tb, Wp, cp) in float64, folded form."""
    x, m = x.astype(np.float64), mask.astype(np.float64)
    E, tb, Wp, cp = (t.astype(np.float64) for t in (E, tb, Wp, cp))
    K = E.shape[1]
    wx, WE, wb = Wp[:, 0], Wp[:, 1:1 + K], Wp[:, 1 + K]
    A = wx[None, :] + E @ WE.T                      # [d, K]
    C = tb * wb[None, :] + cp[None, :]              # [d, K]   (tb is [d, 1])
    pre = x[:, :, None] * A[None] + C[None]         # [B, d, K]
    agg = (m[:, :, None] * np.maximum(pre, 0)).sum(1)
    if dagg is None:
        return agg
    g = m[:, :, None] * (pre > 0) * dagg.astype(np.float64)[:, None, :]   # d pre
    dA = (g * x[:, :, None]).sum(0)                 # [d, K]
    dC = g.sum(0)
    grads = {
        "type_pars1": dA @ WE,                      # dE_j = W_E^T dA_j
        "type_bias1": (dC * wb[None, :]).sum(1, keepdims=True),
        "pnp_encoder1.0.weight": np.concatenate([dA.sum(0)[:, None], dA.T @ E, (dC * tb).sum(0)[:, None]], 1),
        "pnp_encoder1.0.bias": dC.sum(0),
    }
    return agg, grads
