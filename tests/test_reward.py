"""Active-variable-selection reward (BASELINE config 5).

CPU: the oracle's restatement of R_lindley_chain / chaini_I / chaini_II (evaluate.py:514-634) against golden
vectors produced by the reference itself (tests/golden/make_golden.py --reward).
GPU: vpc_reward_matrix (three launches for all rows x candidates x MC samples) against the same vectors and
against the oracle on larger shapes.  The KL terms are differences of O(1) quantities that cancel to ~1e-4, so
fp32 round-off is ~1e-7 ABSOLUTE in both implementations: the tolerance is absolute (5e-7 + 1e-3 * |R|max).
"""
import numpy as np
import pytest
import torch

import vpc_amd as vpc
from conftest import golden_params, load_golden
from oracle import vae_oracle as O

L = 10


def _t(a, dev="cpu"):
    return torch.from_numpy(np.array(a)).to(dev)


@pytest.mark.parametrize("d", [14, 40])
@pytest.mark.parametrize("tag", ["t0", "t1"])
def test_oracle_reward_matches_reference(d, tag):
    g = load_golden(f"reward_d{d}.npz")
    port = O.TorchPort(golden_params(g), L)
    x, mask, im = _t(g["x"]), _t(g[f"mask_{tag}"]), _t(g["im"])
    with torch.no_grad():
        R = O.reward_matrix(port, x, mask, im.shape[0], im)
        k1, k2 = O.chaini_I(port, x, mask, 3), O.chaini_II(port, x, mask, 3)
    want = g[f"R_{tag}"]
    assert np.array_equal(R.numpy() == -1e4, want == -1e4)
    assert np.max(np.abs(R.numpy() - want)) <= 2e-7
    assert np.max(np.abs(k1.numpy() - g[f"kl1_{tag}"])) <= 2e-7 and np.max(np.abs(k2.numpy() - g[f"kl2_{tag}"])) <= 2e-7


def _model(d, params, dev):
    m = vpc.Reg_VAE(d, 500, 10, L, {"batch_size": 8, "patience": 1}, "exp", "kl_reg")
    sd = m.state_dict(); sd.update({k: v.clone() for k, v in params.items()}); m.load_state_dict(sd)
    return m.to(dev)


@pytest.mark.gpu
@pytest.mark.parametrize("d", [14, 40])
@pytest.mark.parametrize("tag", ["t0", "t1"])
def test_gpu_reward_matches_reference(d, tag):
    g = load_golden(f"reward_d{d}.npz")
    m = _model(d, golden_params(g), "cuda")
    x, mask, im = _t(g["x"], "cuda"), _t(g[f"mask_{tag}"], "cuda"), _t(g["im"], "cuda")
    R = vpc.reward_matrix(m, x, mask, im).cpu().numpy()
    want = g[f"R_{tag}"]
    assert np.array_equal(R == -1e4, want == -1e4)
    live = want != -1e4
    tol = 5e-7 + 1e-3 * np.max(np.abs(want[live]))
    assert np.max(np.abs(R[live] - want[live])) <= tol
    # drop-in single-candidate call with the reference's signature
    u = 3
    loc = np.where(g[f"mask_{tag}"][:, u] == 0)[0]
    r1 = vpc.R_lindley_chain(u, x, mask, im.shape[0], m, im, loc).cpu().numpy()
    assert np.max(np.abs(r1 - want[loc, u])) <= tol
    k1 = vpc.chaini_I(x, mask, 3, m).cpu().numpy()
    k2 = vpc.chaini_II(x, mask, 3, m).cpu().numpy()
    assert np.max(np.abs(k1 - g[f"kl1_{tag}"])) <= 1e-6 and np.max(np.abs(k2 - g[f"kl2_{tag}"])) <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("d,n,M", [(128, 64, 50), (100, 33, 7), (13, 5, 17)])
def test_gpu_reward_vs_oracle(d, n, M):
    params = O.init_params(d, L, seed=11)
    params = {k: v * (2.5 if "weight" in k else 1.0) for k, v in params.items()}  # larger weights -> larger rewards
    g = torch.Generator().manual_seed(d + n)
    x = torch.rand(n, d, generator=g)
    mask = (torch.rand(n, d, generator=g) < 0.5).float()
    mask[:, -1] = (torch.rand(n, generator=g) < 0.3).float()
    im = torch.rand(M, n, d, generator=g)
    with torch.no_grad():
        want = O.reward_matrix(O.TorchPort(params, L), x, mask, M, im).numpy()
    m = _model(d, params, "cuda")
    R = vpc.reward_matrix(m, x.cuda(), mask.cuda(), im.cuda()).cpu().numpy()
    assert np.array_equal(R == -1e4, want == -1e4)
    live = want != -1e4
    assert np.max(np.abs(R[live] - want[live])) <= 2e-6 + 1e-3 * np.max(np.abs(want[live]))
    # the acquisition decision (argmax per row) is what the loop consumes (evaluate.py:435)
    top = np.argmax(want, 1)
    got = R[np.arange(n), top]
    assert np.all(np.max(R, 1) - got <= 2e-6 + 1e-3 * np.max(np.abs(want[live])))
