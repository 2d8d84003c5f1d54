"""CPU checks of the host logic that needs no GPU:
  * libvpc_hip.so loads and exports every symbol declared in include/vpc.h;
  * pack_idx / img_template: a numpy emulation of the packed-image MLP chain (swizzled images, "ones trick"
    bias chain, mean|logvar tile split) reproduces the oracle's forward;
  * grad_idx: packed-space gradients dW~ = dY X~^T scattered into a partial block in MFMA C layout and
    gathered through grad_idx reproduce the oracle's parameter gradients.
No kernel is launched.
"""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import vpc_amd as vpc
from oracle import vae_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "vpc.h")).read()
    declared = set(re.findall(r"\b(?:int|long)\s+(vpc_\w+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(vpc._lib.exported_symbols())
    h = ctypes.CDLL(vpc.LIB_PATH)
    for name in declared:
        assert hasattr(h, name), name


_CTYPE = {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double,
          "long long": ctypes.c_longlong, "unsigned long long": ctypes.c_ulonglong}


def _ctype_of(param: str):
    """C parameter declaration -> the ctypes class _lib._PROTOS must use for it."""
    param = re.sub(r"/\*.*?\*/", "", param).strip()
    if "*" in param:
        if param.count("*") == 2:
            return ctypes.POINTER(ctypes.c_void_p)
        base = re.sub(r"\bconst\b", "", param.split("*")[0]).strip()
        return {"int": ctypes.POINTER(ctypes.c_int), "long": ctypes.POINTER(ctypes.c_long)}.get(base, ctypes.c_void_p)
    words = re.sub(r"\bconst\b", "", param).split()
    return _CTYPE[" ".join(words[:-1])]  # drop the parameter name


def test_ctypes_prototypes_match_the_header():
    """Argument count AND type of every entry of _lib._PROTOS against the declaration in include/vpc.h (a drifted
    int / long / float would pass the symbol test and corrupt arguments at run time).  The .hip definitions are checked
    against the same header by the compiler: every translation unit includes it (vpc_abi_internal.h)."""
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "vpc.h")).read(), flags=re.S)
    decls = re.findall(r"\b(int|long)\s+(vpc_\w+)\s*\(([^;]*?)\)\s*;", hdr, flags=re.S)
    assert len(decls) == len(vpc._lib._PROTOS)
    for ret, name, params in decls:
        params = " ".join(params.split())
        want = [] if params in ("", "void") else [_ctype_of(p) for p in params.split(",")]
        got = vpc._lib._PROTOS[name]
        assert len(got) == len(want), (name, len(got), len(want))
        for i, (g, w) in enumerate(zip(got, want)):
            ok = g is w or (w is ctypes.c_void_p and g in (ctypes.c_void_p,)) or \
                (g is ctypes.POINTER(ctypes.c_float) and w is ctypes.c_void_p)  # host float arrays (cA / cE)
            if w in (ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_long)) and g is ctypes.c_void_p:
                ok = True  # int / long arrays passed as raw pointers
            assert ok, (name, i, g, w)
        assert (ret == "long") == (name in vpc._lib._RESTYPE_LONG), name


def test_bad_arguments_are_rejected_without_gpu():
    l = vpc._lib.lib()
    assert l.vpc_layout_sizes(129, 10, 0, *[None] * 8) == 2
    assert l.vpc_layout_sizes(65, 10, 1, *[None] * 8) == 2 and l.vpc_layout_sizes(64, 10, 1, *[None] * 8) == 0
    assert l.vpc_layout_sizes(14, 16, 0, *[None] * 8) == 2
    assert l.vpc_layout_sizes(14, 10, 0, *[None] * 8) == 0
    assert l.vpc_encoder_fwd(None, None, 1, None, None, None, None, None, None, None, 10, 0, 0, 4, 14, 10, None) == 1
    assert l.vpc_adam_step(None, None, None, None, 0, 1e-3, 0.9, 0.999, 1e-8, 1, None, None, None, None, None, None) == 1
    assert l.vpc_build_inverse_maps(None, 1, 2, 4, 4, None, None) == 1
    assert l.vpc_draw_mask(None, None, 8, 0.5, 0, 0, 0, None) == 1
    assert l.vpc_fill_normal(None, 8, 0, 0, None, 0, 0, 0, 4, None) == 1
    # MNAR path: null pointers / bad shapes are rejected before anything touches the device
    assert l.vpc_linear_fwd(None, 4, None, None, None, 4, 8, 4, 4, 0, 0, 0, None) == 1
    assert l.vpc_linear_dgrad(None, 4, None, 4, 0, 0, None, None, 4, 0, None, 4, 8, 4, 4, 0, None) == 1
    assert l.vpc_linear_wgrad(None, 4, None, 4, 0, 0, None, 4, None, None, None, 0, 8, 4, 4, 0, 0, None) == 1
    assert l.vpc_nm_sample(None, 20, None, None, 10, 4, 2, 10, None) == 1
    assert l.vpc_linear_wgrad_scratch(0, 4, 4) == 0 and l.vpc_nm_loss_scratch(0, 4) == 0


def swz(col, row, S=64):
    return (((col >> 2) ^ (row & 15 & (S // 4 - 1))) << 2) | (col & 3)


def unswizzle(img, off, rows, S):
    W = np.zeros((rows, S), np.float64)
    for r in range(rows):
        for c in range(S):
            W[r, c] = img[off + r * S + swz(c, r, S)]
    return W


def images(lay, flat):
    img = lay.img_template.copy()
    img[lay.pack_idx] = flat
    DT = {True: 1}.get(lay.d <= 16, 2 if lay.d <= 32 else 4 if lay.d <= 64 else 8)
    S1 = 128 if DT > 4 else 64
    o = 0
    W1 = unswizzle(img, o, 112, S1); o += 112 * S1
    b1 = img[o:o + 128].astype(np.float64); o += 128
    W2 = unswizzle(img, o, 64, 128); o += 64 * 128
    W3 = unswizzle(img, o, 32, 64); o += 32 * 64
    assert o == lay.enc_img
    W4 = np.zeros((64, 64)); W4[:, :16] = unswizzle(img, o, 64, 16); o += 64 * 16
    W5 = unswizzle(img, o, 112, 64); o += 112 * 64
    W6 = unswizzle(img, o, 16 * DT, 128); o += 16 * DT * 128
    assert o == lay.enc_img + lay.dec_img
    return DT, S1, W1, b1, W2, W3, W4, W5, W6


def flat_of(params):
    return np.concatenate([params[k].numpy().reshape(-1) for k in O.PARAM_KEYS]).astype(np.float32)


@pytest.mark.parametrize("d,Ld", [(14, 10), (128, 10), (40, 6), (64, 15)])
def test_packed_chain_matches_oracle(d, Ld):
    lay = vpc._lib.layout(d, Ld)
    assert len(np.unique(lay.pack_idx)) == lay.n_params
    assert len(np.unique(lay.grad_idx[:lay.n_enc])) == lay.n_enc
    assert len(np.unique(lay.grad_idx[lay.n_enc:])) == lay.n_params - lay.n_enc
    params = O.init_params(d, Ld, seed=3)
    flat = flat_of(params)
    assert flat.size == lay.n_params
    DT, S1, W1, b1, W2, W3, W4, W5, W6 = images(lay, flat)
    rng = np.random.default_rng(0)
    B = 8
    x = rng.random((B, d))
    m = (rng.random((B, d)) < 0.7).astype(np.float64)
    mp = m * (rng.random((B, d)) < 0.7)
    eq, ep = rng.standard_normal((B, Ld)), rng.standard_normal((B, Ld))
    P = O._np(params)
    cq = O.closed_form_pass(P, x, m, eq, Ld)

    def fwd(mask, eps):
        xin = np.zeros((B, S1)); xin[:, :d] = x * mask
        h1 = np.maximum(xin @ W1.T + b1[:112], 0)            # [B,112], constant unit == 1
        h1p = np.zeros((B, 128)); h1p[:, :112] = h1
        h2 = np.maximum(h1p @ W2.T, 0)                       # [B,64],  constant unit == 1
        o = h2 @ W3.T                                        # [B,32]   mean tile | logvar tile
        mu, lv = o[:, :Ld], o[:, 16:16 + Ld]
        z = np.zeros((B, 64)); z[:, :Ld] = mu + eps * np.exp(lv / 2); z[:, Ld] = 1.0
        g1 = np.maximum(z @ W4.T, 0)                         # [B,64],  constant unit == 1
        g2 = np.maximum(g1 @ W5.T, 0)                        # [B,112], constant unit == 1
        g2p = np.zeros((B, 128)); g2p[:, :112] = g2
        xh = 1 / (1 + np.exp(-(g2p @ W6.T)))                 # [B,16*DT]
        return dict(xin=xin, h1=h1, h1p=h1p, h2=h2, o=o, mu=mu, lv=lv, z=z, g1=g1, g2=g2, g2p=g2p, xh=xh)

    fq = fwd(m, eq)
    from vpc_amd._lib import HIDDEN_POS1 as P1, HIDDEN_POS2 as P2  # unit -> position in the padded width
    assert np.allclose(fq["h1"][:, P1[100]], 1) and np.allclose(fq["h2"][:, P2[50]], 1)
    assert np.allclose(fq["g1"][:, P2[50]], 1) and np.allclose(fq["g2"][:, P1[100]], 1)
    assert np.allclose(fq["mu"], cq.mean, atol=1e-6) and np.allclose(fq["lv"], cq.logvar, atol=1e-6)
    assert np.allclose(fq["xh"][:, :d], cq.xhat, atol=1e-6)
    assert np.allclose(fq["h1"][:, P1[:100]], cq.h1, atol=1e-6) and np.allclose(fq["g2"][:, P1[:100]], cq.g2, atol=1e-6)

    # ---- packed-space backward for the kl_reg loss, scattered into fake partial blocks, gathered by grad_idx
    _, grads, (cq, cp), _ = O.closed_form_reg_step(params, Ld, x, m, mp, eq, ep, alpha=0.7, beta=0.9)
    fp = fwd(mp, ep)
    s2 = np.exp(O.X_LOGVAR)
    alpha, bw = 0.7, 0.9
    E = m * (1 - mp)
    diff = cq.mean - cp.mean; eip = np.exp(-cp.logvar); r = np.exp(cq.logvar - cp.logvar)
    seeds = {
        "q": (((1 - alpha) * m + alpha * E) * (cq.xhat - x) / s2, (1 - alpha) * bw * cq.mean + alpha * diff * eip,
              (1 - alpha) * bw * 0.5 * (np.exp(cq.logvar) - 1) + alpha * 0.5 * (r - 1), eq),
        "p": (alpha * mp * (cp.xhat - x) / s2, alpha * bw * cp.mean - alpha * diff * eip,
              alpha * bw * 0.5 * (np.exp(cp.logvar) - 1) + alpha * 0.5 * (1 - r - diff ** 2 * eip), ep),
    }
    encp = np.zeros(lay.enc_part); decp = np.zeros(lay.dec_part)

    def scatter(block, dW, owner_reg, gregs=48):
        # dW [out_pad][in_pad]; element (o, i) -> wave/reg given by owner_reg(mt, nt), C layout inside the tile
        for o in range(dW.shape[0]):
            for i in range(dW.shape[1]):
                wave, reg = owner_reg(o >> 4, i >> 4)
                if wave is None:
                    continue
                ro, ci = o & 15, i & 15
                block[(wave * gregs + reg + (ro & 3)) * 64 + (ro >> 2) * 16 + ci] += dW[o, i]

    for tag, f in (("q", fq), ("p", fp)):
        dxh, dmk, dlk, eps = seeds[tag]
        dxh = dxh / B; dmk = dmk / B; dlk = dlk / B
        dpre = np.zeros((B, 16 * DT)); dpre[:, :d] = dxh * f["xh"][:, :d] * (1 - f["xh"][:, :d])
        # decoder kernel: 4 waves x 92 regs; wave = mt & 3, second out tile (mt >> 2) in the upper half
        scatter(decp, dpre.T @ f["g2"], lambda mt, nt: (mt & 3, 28 * (mt >> 2) + 4 * nt), 92)
        dg2 = (dpre @ W6)[:, :112] * (f["g2"] > 0)
        scatter(decp, dg2.T @ f["g1"], lambda mt, nt: (mt & 3, 56 + 16 * (mt >> 2) + 4 * nt), 92)
        dg1 = (dg2 @ W5) * (f["g1"] > 0)
        scatter(decp, dg1.T @ f["z"][:, :16], lambda mt, nt: (mt, 88), 92)
        dz = (dg1 @ W4)[:, :Ld]
        dml = np.zeros((B, 32))
        dml[:, :Ld] = dmk + dz
        dml[:, 16:16 + Ld] = dlk + dz * eps * 0.5 * np.exp(f["lv"] / 2)
        scatter(encp, dml.T @ f["h2"], lambda mt, nt: (mt * 4 + nt, 44))
        dh2 = (dml @ W3) * (f["h2"] > 0)
        scatter(encp, dh2.T @ f["h1"], lambda mt, nt: (nt, 28 + 4 * mt))
        dh1 = (dh2 @ W2)[:, :112] * (f["h1"] > 0)
        scatter(encp, dh1.T @ f["xin"][:, :16 * DT], lambda mt, nt: (nt, 4 * mt))
        encp[8 * 48 * 64: 8 * 48 * 64 + 112] += dh1.sum(0)
    got = np.concatenate([encp[lay.grad_idx[:lay.n_enc]], decp[lay.grad_idx[lay.n_enc:]]])
    want = np.concatenate([grads[k].reshape(-1) for k in O.PARAM_KEYS])
    assert np.max(np.abs(got - want)) <= 1e-9 * max(1.0, np.max(np.abs(want)))


def test_model_state_dict_keys_and_flat_views():
    m = vpc.Reg_VAE(14, 500, 10, 10, {"batch_size": 64, "patience": 100}, "exp", "kl_reg")
    keys = list(m.state_dict().keys())
    assert keys == ["prior_mean", "prior_std"] + list(O.PARAM_KEYS)
    assert m.x_logvar.shape == (1,) and abs(float(m.x_logvar) - O.X_LOGVAR) < 1e-6
    flat = m.flatten_parameters()
    assert flat.numel() == 201 * 14 + 11820  # 201 d + 11 820 trainable scalars (37 548 at d = 128)
    w = m.seq_encoder[0].weight
    w.data.add_(1.0)
    assert torch.equal(flat[: w.numel()].view_as(w), w.data)
    assert m.flatten_parameters() is flat  # idempotent
    with pytest.raises(vpc.VpcError):
        m.forward(torch.zeros(4, 14), torch.ones(4, 14, dtype=torch.bool), torch.ones(4, 14, dtype=torch.bool))
    # reference checkpoints (plain state_dict) load
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m2 = vpc.vanilla_VAE(14, 500, 10, 10, {"batch_size": 64, "patience": 100}, "exp")
    m2.load_state_dict(sd)
    assert torch.equal(m2.seq_decoder[4].bias, m.seq_decoder[4].bias)


def test_checkpoint_path_format():
    p = vpc.checkpoint_path("exp", "synth", "reg_vae1", 30, 1.0, 30, "kl_reg")
    assert p == "experiments/exp/synth/checkpoints/reg_vae/checkpoint_reg_vae1_1.0_30_kl_reg_30_missing_rate_full_reg_test.pt"
    p = vpc.checkpoint_path("exp", "synth", "vanilla_vae2", 50)
    assert p == "experiments/exp/synth/checkpoints/vanilla_vae/checkpoint_vanilla_vae2_50_missing_rate_test.pt"
