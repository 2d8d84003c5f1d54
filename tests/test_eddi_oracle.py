"""CPU: pin the EDDI oracle (oracle/eddi_oracle.py) to vectors captured from the reference itself
(tests/golden/eddi_*.npz, tests/golden/make_golden_eddi.py).  SURVEY.md section 8 row f-3."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import eddi_oracle as E


def _params(g, prefix="param.", dtype=torch.float32):
    return {k: torch.from_numpy(g[prefix + k]).to(dtype).clone().requires_grad_(True) for k in E.EDDI_KEYS}


def _t(g, k):
    return torch.from_numpy(g[k])


@pytest.mark.parametrize("d", [14, 40])
def test_reg_eddi(d):
    g = load_golden(f"eddi_reg_d{d}.npz")
    L = int(g["L"])
    x, m, mp = _t(g, "x"), _t(g, "mask"), _t(g, "mask_p")
    names = ["mean_p", "logvar_p", "x_mean_p", "x_logvar_p", "mean_q", "logvar_q", "x_mean_q", "x_logvar_q"]
    for tag, rt, alpha in (("kl0.5", "kl_reg", 0.5), ("kl1.0", "kl_reg", 1.0), ("ml0.8", "ml_reg", 0.8)):
        p = _params(g)
        port = E.EDDIPort(p, L, rt)
        o = port.reg_forward(x, m, mp, _t(g, "eps_q"), _t(g, "eps_p"))
        for n, t in zip(names, o):
            np.testing.assert_allclose(t.detach().numpy(), g["fwd." + n], rtol=1e-5, atol=1e-6)
        _, tl = port.reg_loss(x, o[2], o[3], o[0], o[1], o[6], o[7], o[4], o[5], m, mp, 1400, beta=0.9, alpha=alpha,
                              beta_annealing=(tag == "kl1.0"), eps_ml=_t(g, "eps_ml"))
        assert abs(tl.item() - float(g[f"loss.{tag}"])) <= 3e-6 * abs(float(g[f"loss.{tag}"]))
        tl.backward()
        for k in E.EDDI_KEYS:
            ref = g[f"grad.{tag}.{k}"]
            np.testing.assert_allclose(p[k].grad.numpy(), ref, rtol=3e-4, atol=3e-6 * max(1.0, np.abs(ref).max()))
    with torch.no_grad():
        port = E.EDDIPort(_params(g), L, "kl_reg")
        o = port.reg_forward(x, m, mp, _t(g, "eps_q"), _t(g, "eps_p"))
        r = port.reg_loss(x, o[2], o[3], o[0], o[1], o[6], o[7], o[4], o[5], m, mp, 7, llh_eval=True, stage="evaluate")
    for got, key in zip(r[1:], ("eval_loss", "eval_re", "eval_re_imp")):
        assert abs(float(got) - float(g[key])) <= 3e-6 * abs(float(g[key]))


@pytest.mark.parametrize("d", [14, 40])
def test_vanilla_eddi(d):
    g = load_golden(f"eddi_van_d{d}.npz")
    L = int(g["L"])
    x, m = _t(g, "x"), _t(g, "mask").float()
    p = _params(g)
    port = E.EDDIPort(p, L)
    o = port.vanilla_forward(x, m, _t(g, "eps_q"))
    for n, t in zip(["mean", "logvar", "x_mean", "x_logvar"], o):
        np.testing.assert_allclose(t.detach().numpy(), g["fwd." + n], rtol=1e-5, atol=1e-6)
    r = port.vanilla_loss(x, o[2], o[3], o[0], o[1], 3, m, beta=0.8, llh_eval=True)
    assert abs(r[1].item() - float(g["loss"])) <= 3e-6 * abs(float(g["loss"]))
    assert abs(float(r[2]) - float(g["re"])) <= 3e-6 * abs(float(g["re"]))
    assert abs(float(r[3]) - float(g["re_imp"])) <= 3e-6 * abs(float(g["re_imp"]))
    r[1].backward()
    for k in E.EDDI_KEYS:
        ref = g[f"grad.v.{k}"]
        np.testing.assert_allclose(p[k].grad.numpy(), ref, rtol=3e-4, atol=3e-6 * max(1.0, np.abs(ref).max()))


def test_front_closed_form_matches_autograd():
    g = load_golden("eddi_reg_d14.npz")
    p = _params(g, dtype=torch.float64)
    port = E.EDDIPort(p, int(g["L"]))
    x, m = _t(g, "x").double(), _t(g, "mask")
    agg = port.front(x, m)
    rng = np.random.default_rng(0)
    dagg = rng.normal(size=tuple(agg.shape))
    (agg * torch.from_numpy(dagg)).sum().backward()
    cf, gr = E.front_closed_form(g["x"], g["mask"], g["param.type_pars1"], g["param.type_bias1"],
                                 g["param.pnp_encoder1.0.weight"], g["param.pnp_encoder1.0.bias"], dagg)
    np.testing.assert_allclose(cf, agg.detach().numpy(), rtol=1e-12, atol=1e-13)
    for k, v in gr.items():
        np.testing.assert_allclose(v, p[k].grad.numpy(), rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("kind", ["reg", "van"])
def test_adam_trajectory(kind):
    g = load_golden(f"eddi_traj_{kind}_d14.npz")
    L = int(g["L"])
    p = {k: torch.from_numpy(g["param0." + k]).clone().requires_grad_(True) for k in E.EDDI_KEYS}
    opt = torch.optim.Adam([p[k] for k in E.EDDI_KEYS], lr=1e-3)
    port = E.EDDIPort(p, L, "kl_reg")
    x, m = _t(g, "x"), _t(g, "mask")
    for s in range(len(g["losses"])):
        eps = torch.from_numpy(g["eps"][s])
        if kind == "reg":
            mp = torch.from_numpy(g["mask_p"][s])
            o = port.reg_forward(x, m, mp, eps[0], eps[1])
            _, tl = port.reg_loss(x, o[2], o[3], o[0], o[1], o[6], o[7], o[4], o[5], m, mp, s + 1, alpha=0.5)
        else:
            mf = m.float()
            o = port.vanilla_forward(x, mf, eps[0])
            _, tl = port.vanilla_loss(x, o[2], o[3], o[0], o[1], s + 1, mf)
        opt.zero_grad()
        tl.backward()
        opt.step()
        assert abs(tl.item() - g["losses"][s]) <= 5e-6 * abs(g["losses"][s])
    for k in E.EDDI_KEYS:
        np.testing.assert_allclose(p[k].detach().numpy(), g["param5." + k], rtol=1e-4, atol=2e-6)
