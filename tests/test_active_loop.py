"""Active-variable-selection LOOP (BASELINE config 5; reference src/experiment_main/evaluate.py:300-511,
active_learning.py:23-74) against vectors recorded from the reference's own `active_learning_func`
(tests/golden/active_{reg,van}_d14.npz, produced by tests/golden/make_golden_active.py).

The reference draws eps from torch's global RNG inside every model.forward, so the forward outputs (x_mean_q of every
call, in order) are part of the fixture and are replayed; everything downstream - the reward matrix of every step, the
argmax acquisition order per row, the mask bookkeeping and the target-MSE information curve - is then deterministic and is
compared: CPU = the oracle's restatement, GPU = vpc.active_learning_func (one vpc_reward_matrix launch set per step).
"""
import os

import numpy as np
import pytest
import torch

import vpc_amd as vpc
from conftest import golden_params, load_golden
from oracle import vae_oracle as O

KINDS = ["reg", "van"]


def _replayer(g):
    it = iter(g["fwd_xmean"])
    return lambda mask: torch.from_numpy(next(it).copy())


@pytest.mark.parametrize("kind", KINDS)
def test_oracle_loop_matches_reference(kind):
    g = load_golden(f"active_{kind}_d14.npz")
    L, M = int(g["L"]), int(g["M"])
    port = O.TorchPort(golden_params(g), L)
    x = torch.from_numpy(g["x"])
    with torch.no_grad():
        out = O.active_learning_loop(x, M, _replayer(g), lambda xx, m, im: O.reward_matrix(port, xx, m, M, im))
    assert np.array_equal(out["action"].numpy(), g["action"])
    assert np.array_equal(out["im"].numpy(), g["im"])
    assert np.max(np.abs(out["R_hist"].numpy() - g["R_hist"])) <= 3e-7
    assert np.allclose(out["info_curve"].numpy(), g["info_curve"], rtol=1e-6, atol=1e-8)
    # the acquisitions help: the target MSE after the last acquisition is far below the all-unobserved start
    assert g["info_curve"][-1] < 0.5 * g["info_curve"][0]


def _model(kind, g, dev):
    d, L = g["x"].shape[1], int(g["L"])
    tp = {"batch_size": 64, "patience": 100}
    m = vpc.Reg_VAE(d, 500, 10, L, tp, "exp", "kl_reg") if kind == "reg" else vpc.vanilla_VAE(d, 500, 10, L, tp, "exp")
    sd = m.state_dict(); sd.update({k: v.clone() for k, v in golden_params(g).items()}); m.load_state_dict(sd)
    return m.to(dev)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
def test_gpu_loop_matches_reference(kind, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    g = load_golden(f"active_{kind}_d14.npz")
    d, L, M = g["x"].shape[1], int(g["L"]), int(g["M"])
    m = _model(kind, g, "cuda")
    x, tmask = torch.from_numpy(g["x"]), torch.from_numpy(g["test_mask"])
    vae_type = "reg_vae1" if kind == "reg" else "vanilla_vae1"
    out = vpc.active_learning_func(None, x, tmask, 30, d, 500, 10, M, L, "toy", {"batch_size": 64, "patience": 100}, "exp",
                                   vae_type, 100, 1, 1, alpha=1.0, p_missingness=30, reg_type="kl_reg", Repeat=1, model=m,
                                   _forward=_replayer(g))
    R, want = out["R_hist_CHAI"][0].numpy(), g["R_hist"]
    act, act_ref = out["action_CHAI"][0].numpy(), g["action"]
    # rewards are differences of O(1) KL terms that cancel to ~1e-4: fp32 round-off is absolute (tests/test_reward.py)
    tol = 5e-7 + 1e-3 * np.max(np.abs(want[want != -1e4]))
    n = x.shape[0]
    for t in range(d - 1):
        same = np.all(act[:, :t] == act_ref[:, :t], axis=1)  # rows whose mask history equals the reference's so far
        assert same.mean() > 0.9
        assert np.array_equal(R[t][same] == -1e4, want[t][same] == -1e4), t
        assert np.max(np.abs(R[t][same] - want[t][same])) <= tol, t
        # the acquisition: the same feature, or one whose reference reward is within the round-off of the best
        pick = act[same, t].astype(int)
        gap = want[t][same].max(1) - want[t][same][np.arange(same.sum()), pick]
        assert np.all(gap <= tol), (t, gap.max())
    assert (act == act_ref).mean() > 0.98
    assert np.array_equal(out["im_CHAI"][0].numpy(), g["im"])
    assert np.allclose(out["information_curve_CHAI"][0, 0].numpy(), g["info_curve"], rtol=1e-5, atol=1e-7)
    # the four result files carry the reference's names
    fam = "reg_vae" if kind == "reg" else "vanilla_vae"
    assert sorted(os.listdir(os.path.join("experiments", "exp", "toy", "rest", fam))) == sorted(str(f) for f in g["files"])


@pytest.mark.gpu
def test_gpu_loop_with_device_side_draws(tmp_path, monkeypatch):
    """The product path proper: model.forward on the GPU (M passes as one batched pass), checkpoint found through
    model_loader('test') in the reference's naming scheme.  RNG-dependent, so properties only: every row acquires d - 1
    distinct features, the information curve ends far below its start and close to the reference's final value."""
    monkeypatch.chdir(tmp_path)
    g = load_golden("active_reg_d14.npz")
    d, L = g["x"].shape[1], int(g["L"])
    m = _model("reg", g, "cpu")
    ck = vpc.checkpoint_path("exp", "toy", "reg_vae1", 30, alpha=1.0, p_missingness=30, reg_type="kl_reg")
    os.makedirs(os.path.dirname(ck), exist_ok=True)
    torch.save(m.state_dict(), ck)
    x, tmask = torch.from_numpy(g["x"]), torch.from_numpy(g["test_mask"])
    out = vpc.active_learning_func(None, x, tmask, 30, d, 500, 10, 25, L, "toy", {"batch_size": 64, "patience": 100}, "exp",
                                   "reg_vae1", 100, 1, 1, alpha=1.0, p_missingness=30, reg_type="kl_reg", Repeat=2)
    act = out["action_CHAI"].numpy()
    assert act.shape == (2, x.shape[0], d - 1)
    for r in range(2):
        for row in act[r]:
            assert sorted(row.astype(int)) == list(range(d - 1))
    curve = out["information_curve_CHAI"][:, 0].numpy()
    assert np.all(curve[:, -1] < 0.5 * curve[:, 0])
    assert np.all(np.abs(curve[:, -1] - g["info_curve"][-1]) < 0.3 * g["info_curve"][-1] + 5e-3)


@pytest.mark.gpu
def test_gpu_loop_at_full_width_vs_oracle(tmp_path, monkeypatch):
    """Config 5 at the width BASELINE names (d = 128): the first 3 acquisition steps of active_learning_func with n = 32 rows
    and M = 8 Monte-Carlo passes, the forward passes REPLAYED into both loops (one seeded stream of decoder outputs), GPU
    loop (vpc_reward_matrix: rank-1 first-layer updates on MFMA) against the oracle's restatement of
    evaluate.py:394-440, 514-634 (pinned to the reference's own run at d = 14 above): every reward within 1e-4 of the
    reward scale + fp32 round-off, identical acquisitions."""
    monkeypatch.chdir(tmp_path)
    d, L, n, M, steps = 128, 10, 32, 8, 3
    g = torch.Generator().manual_seed(11)
    base, mix = torch.rand(n + 4096, 4, generator=g), torch.rand(4, d, generator=g)
    data = torch.sigmoid(3.0 * (base @ mix / mix.sum(0) - 0.5)) + 0.05 * torch.rand(n + 4096, d, generator=g)
    data = (data - data.min(0).values) / (data.max(0).values - data.min(0).values)
    torch.manual_seed(3)
    m = vpc.Reg_VAE(d, 500, 10, L, {"batch_size": 64, "patience": 100}, "exp", "kl_reg").cuda()
    tr = vpc.FusedTrainer(m)
    xt, mt = data[n:].cuda(), (torch.rand(4096, d, generator=g) < 0.7).cuda()
    for i in range(200):  # a briefly trained model: rewards of a random-init encoder are all round-off
        tr.step(xt, mt, alpha=1.0, epoch=i + 1)
    params = {k: v.detach().cpu().clone() for k, v in m.state_dict().items() if "prior" not in k}
    port = O.TorchPort(params, L)
    x, tmask = data[:n].clone(), torch.rand(n, d, generator=g) < 0.7

    def replay(seed):  # one model.forward: x_mean_q of the oracle's port under a seeded eps stream
        gg = torch.Generator().manual_seed(seed)

        def fwd(mask):
            eps = torch.randn(n, L, generator=gg)
            with torch.no_grad():
                z, _, _ = port.encoder(x, mask.cpu() > 0.5, eps=eps)
                return port.decoder(z)[0]
        return fwd

    with torch.no_grad():
        ref = O.active_learning_loop(x, M, replay(5), lambda xx, mm, im: O.reward_matrix(port, xx, mm, M, im), max_steps=steps)
    out = vpc.active_learning_func(None, x, tmask, 30, d, 500, 10, M, L, "toy", {"batch_size": 64, "patience": 100}, "exp",
                                   "reg_vae1", 100, 1, 1, alpha=1.0, p_missingness=30, reg_type="kl_reg", Repeat=1, model=m,
                                   _forward=replay(5), max_steps=steps, save=False)
    R, want = out["R_hist_CHAI"][0, :steps].numpy(), ref["R_hist"].numpy()
    live = want != -1e4
    scale = np.max(np.abs(want[live]))
    assert scale > 1e-3  # the trained encoder does discriminate between candidates
    assert np.array_equal(R == -1e4, ~live)
    assert np.max(np.abs(R[live] - want[live])) <= 1e-4 * scale + 5e-7, (np.max(np.abs(R[live] - want[live])), scale)
    assert np.array_equal(out["action_CHAI"][0, :, :steps].numpy(), ref["action"].numpy())
    assert np.array_equal(out["im_CHAI"][0, :steps].numpy(), ref["im"].numpy())
    assert np.allclose(out["information_curve_CHAI"][0, 0, :steps + 1].numpy(), ref["info_curve"].numpy(), rtol=1e-5, atol=1e-7)
