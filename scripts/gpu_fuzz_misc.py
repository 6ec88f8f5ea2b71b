"""Fuzz of the remaining entry points: unscented-filter gradient, forecasts (EKF / UKF, every kernel family), emission moments, UKF
hyper-parameters: python3 scripts/gpu_fuzz_misc.py [seed] [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi
from helpers import linear_model, lorenz96_model, mlp_model, params_from, relerr, FILTER_KEYS

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rng = np.random.default_rng(seed)
L = _ffi.lib()
worst = {}
def note(name, e, tol, tag):
    worst[name] = max(worst.get(name, 0.0), e)
    if not (e < tol):
        print("MISMATCH", name, tag, e, L.cdkf_last_kernel().decode()[:60], flush=True)
def spd(n, s):
    A = rng.standard_normal((n, n))
    return A @ A.T / n * s + 0.3 * np.eye(n)
for case in range(cases):
    # ---- unscented filter: hyper-parameters, gradient ----
    if rng.random() < 0.6:
        m = int(rng.integers(1, 4)); mdl = o.lorenz63_model(m)
    else:
        d, m = [(1, 1), (2, 1), (2, 2), (3, 3)][int(rng.integers(0, 4))]; mdl = linear_model(rng, d, m)
    alpha, beta, kappa = float(rng.uniform(0.5, 2.0)), int(rng.integers(0, 4)), int(rng.integers(0, 3))
    N, T = int(rng.choice([1, 5, 70])), int(rng.integers(1, 12))
    t = o.irregular_times(rng, N, T, 0.012 * T * rng.choice([1, 4]))
    y = o.simulate(mdl, t, rng)
    tag = f"ukf d={mdl.d} m={mdl.m} N={N} T={T} a={alpha:.2f} b={beta} k={kappa}"
    hyp = cd.UKFHyperParams(alpha=alpha, beta=beta, kappa=kappa)
    ref = o.ukf_filter(mdl, t, y, alpha=alpha, beta=beta, kappa=kappa)
    if np.isfinite(ref["filtered_means"]).all():
        post = cd.cdnlgssm_filter(params_from(mdl), y, t[..., None], hyp)
        note("ukf", max([relerr(getattr(post, f), ref[f]) for f in FILTER_KEYS] + [relerr(post.marginal_loglik, ref["marginal_loglik"])]), 1e-8, tag)
        lit = cd.cdnlgssm_filter(params_from(mdl), y, t[..., None], hyp._replace(sigma_points=True))
        note("ukf_sigma", relerr(lit.filtered_means, ref["filtered_means"]), 1e-8, tag)
        try:
            ll_ref, g_ref = o.ukf_loglik_grad(mdl, t, y, alpha=alpha, beta=beta, kappa=kappa)
            ll, g = cd.cdnlgssm_loglik_and_grad(params_from(mdl), y, t[..., None], hyp)
            flat = np.concatenate([np.asarray(a_).reshape(N, -1) for a_ in g], axis=-1)
            note("ukf_grad", max(np.abs(flat - g_ref).max() / (np.abs(g_ref).max() + 1e-300), relerr(ll, ll_ref)), 1e-7, tag)
        except NotImplementedError:
            pass
    # ---- forecasts ----
    kind = rng.choice(["lorenz63", "linear", "lorenz96", "mlp"])
    if kind == "lorenz63": fm = o.lorenz63_model(1)
    elif kind == "linear": fm = linear_model(rng, int(rng.integers(1, 10)), 1)
    elif kind == "lorenz96": fm = lorenz96_model(int(rng.choice([4, 6, 12, 20, 40])), 1)
    else: fm = mlp_model(rng, int(rng.integers(1, 9)), 1, (int(rng.integers(1, 33)), int(rng.integers(1, 33))))
    d = fm.d
    m_init, P_init = fm.m0 + rng.standard_normal(d), spd(d, 1.0)
    n = int(rng.integers(1, 15))
    t_init = float(rng.uniform(0, 1))
    tf = t_init + np.cumsum(rng.uniform(0.001, 0.05, size=n))
    order = str(rng.choice(["second", "first", "zeroth"]))
    tag = f"forecast {kind} d={d} n={n} {order}"
    rm, rP = o.forecast(fm, m_init, P_init, np.array([t_init]), tf[None], method="ekf", state_order=order)
    try:
        fc = cd.cdnlgssm_forecast(params_from(fm), (m_init, P_init), np.array([[t_init]]), tf[:, None], cd.EKFHyperParams(state_order=order))
        note("forecast", max(relerr(fc.forecasted_state_means, rm[0]), relerr(fc.forecasted_state_covariances, rP[0])), 1e-8, tag)
    except NotImplementedError:
        pass
    if d <= 8:
        rm, rP = o.forecast(fm, m_init, P_init, np.array([t_init]), tf[None], method="ukf")
        try:
            fc = cd.cdnlgssm_forecast(params_from(fm), (m_init, P_init), np.array([[t_init]]), tf[:, None], cd.UKFHyperParams())
            if np.isfinite(rm).all():
                note("forecast_ukf", max(relerr(fc.forecasted_state_means, rm[0]), relerr(fc.forecasted_state_covariances, rP[0])), 1e-7, tag)
        except NotImplementedError:
            pass
    # ---- emission moments ----
    d, m = int(rng.integers(1, 50)), int(rng.integers(1, 40))
    em_mdl = linear_model(rng, d, m)
    lead = tuple(int(v) for v in rng.integers(1, 6, size=int(rng.integers(1, 3))))
    mu = rng.standard_normal(lead + (d,))
    A = rng.standard_normal(lead + (d, d))
    cov = A @ np.swapaxes(A, -1, -2)
    em, ec = cd.cdnlgssm_emissions(params_from(em_mdl), np.zeros((lead[-1], 1)), mu, cov)
    note("emissions", max(relerr(em, mu @ em_mdl.H.T + em_mdl.bias), relerr(ec, em_mdl.H @ cov @ em_mdl.H.T + em_mdl.R)), 1e-10, f"emissions d={d} m={m} lead={lead}")
print("fuzz misc seed", seed, "cases", cases, "worst", {k: float(f"{v:.3g}") for k, v in worst.items()}, flush=True)
