"""Fuzz of the solver settings (other Runge-Kutta methods, PIDController) and of the linear front-end (filter with bias / inputs,
smoother types 1 and 2) across kernel families: python3 scripts/gpu_fuzz_solvers.py [seed] [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi
from helpers import linear_model, mlp_model, params_from, relerr, FILTER_KEYS

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rng = np.random.default_rng(seed)
L = _ffi.lib()
def spd(n, s):
    A = rng.standard_normal((n, n))
    return A @ A.T / n * s + 0.3 * np.eye(n)
worst, kernels = {}, {}
def note(name, e, tol, tag):
    worst[name] = max(worst.get(name, 0.0), e)
    if not (e < tol):
        print("MISMATCH", name, tag, e, L.cdkf_last_kernel().decode()[:60], flush=True)
for case in range(cases):
    kind = rng.choice(["linear", "lorenz63", "lorenz96", "mlp"])
    if kind == "linear":
        d = int(rng.integers(1, 13)); drift = linear_model(rng, d, 1).drift
    elif kind == "lorenz63":
        d = 3; drift = o.Lorenz63Drift(10.0, 28.0, 8 / 3)
    elif kind == "lorenz96":
        d = int(rng.choice([4, 6, 8, 12, 16, 20, 40])); drift = o.Lorenz96Drift(8.0)
    else:
        d = int(rng.integers(1, 11)); drift = mlp_model(rng, d, 1, (int(rng.integers(1, 33)), int(rng.integers(1, 33)))).drift
    m = int(rng.integers(1, d + 1))
    H = np.eye(d)[:m] if rng.random() < 0.5 else rng.standard_normal((m, d)) / np.sqrt(d)
    mdl = o.Model(drift, np.eye(d), spd(d, 0.3), H, np.zeros(m), spd(m, 0.5), ({"lorenz96": 8.0}.get(kind, 0.0)) + rng.standard_normal(d), spd(d, 0.5))
    N, T = int(rng.choice([1, 4, 40])), int(rng.integers(2, 8))
    if d > 12: N = min(N, 4)
    t = o.irregular_times(rng, N, T, 0.03 * T)
    y = o.simulate(mdl, t, rng)
    adaptive = rng.random() < 0.4
    solver = str(rng.choice(["dopri5", "tsit5", "bosh3", "heun"] if adaptive else ["tsit5", "bosh3", "heun", "midpoint", "ralston", "euler"]))
    ctrl = dict(rtol=float(10.0 ** rng.uniform(-7, -4)), atol=float(10.0 ** rng.uniform(-9, -6))) if adaptive else None
    settings = {"solver": solver, "dt0": 0.01 if not adaptive else 0.05}
    if adaptive:
        settings["stepsize_controller"] = cd.PIDController(**ctrl)
    order = str(rng.choice(["second", "first"]))
    tag = f"{kind} d={d} m={m} N={N} T={T} {order} {solver} {ctrl}"
    P = params_from(mdl)
    hyp = cd.EKFHyperParams(state_order=order, diffeqsolve_settings=settings)
    try:
        with o.use_solver(solver, adaptive=ctrl):
            ref = o.ekf_filter(mdl, t, y, state_order=order, dt0=settings["dt0"])
            refs = o.ekf_smoother(mdl, t, y, state_order=order, dt0=settings["dt0"])
        if not np.isfinite(ref["filtered_means"]).all():
            continue
        post = cd.cdnlgssm_filter(P, y, t[..., None], hyp)
        k = L.cdkf_last_kernel().decode().split("<")[0]; kernels[k] = kernels.get(k, 0) + 1
        note("ekf_adaptive" if adaptive else "ekf_fixed", max([relerr(getattr(post, f), ref[f]) for f in FILTER_KEYS] + [relerr(post.marginal_loglik, ref["marginal_loglik"])]), 1e-7 if adaptive else 1e-9, tag)
        sm = cd.cdnlgssm_smoother(P, y, t[..., None], hyp)
        k = L.cdkf_last_kernel().decode().split("<")[0]; kernels[k] = kernels.get(k, 0) + 1
        note("eks_adaptive" if adaptive else "eks_fixed", max(relerr(sm.smoothed_means, refs["smoothed_means"]), relerr(sm.smoothed_covariances, refs["smoothed_covariances"])), 1e-6 if adaptive else 1e-8, tag)
    except NotImplementedError as e:
        pass
    # ---- linear front-end ----
    d = int(rng.integers(1, 9)); m = int(rng.integers(1, 9)); nu = int(rng.integers(1, 4))
    base = linear_model(rng, d, m)
    lm = o.Model(o.LinearDrift(base.drift.W, np.zeros(d)), base.L, base.Qc, base.H, base.bias, base.R, base.m0, base.P0)
    b, B, D = 0.3 * rng.standard_normal(d), rng.standard_normal((d, nu)), rng.standard_normal((m, nu))
    N, T = int(rng.choice([1, 3, 20])), int(rng.integers(2, 12))
    t = o.irregular_times(rng, N, T, 0.1 * T)
    u = rng.standard_normal((N, T, nu))
    y = o.simulate(lm, t, rng) + u @ D.T
    model = cd.ContDiscreteLinearGaussianSSM(state_dim=d, emission_dim=m, input_dim=nu, has_dynamics_bias=True, has_emissions_bias=True)
    pp = cd.ParameterProperties()
    params, _ = model.initialize(
        initial_mean={"params": lm.m0, "props": pp}, initial_cov={"params": lm.P0, "props": pp},
        dynamics_weights={"params": lm.drift.W, "props": pp}, dynamics_bias={"params": b, "props": pp},
        dynamics_input_weights={"params": B, "props": pp}, dynamics_diffusion_coefficient={"params": lm.L, "props": pp},
        dynamics_diffusion_cov={"params": lm.Qc, "props": pp}, emission_weights={"params": lm.H, "props": pp},
        emission_bias={"params": lm.bias, "props": pp}, emission_input_weights={"params": D, "props": pp},
        emission_cov={"params": lm.R, "props": pp})
    tag = f"linear-front d={d} m={m} nu={nu} N={N} T={T}"
    ref = o.kf_filter_inputs(lm, t, y, b, B, D, u)
    post = model.filter(params, y, t[..., None], inputs=u)
    note("kf_inputs", max([relerr(getattr(post, f), ref[f]) for f in FILTER_KEYS] + [relerr(post.marginal_loglik, ref["marginal_loglik"])]), 1e-8, tag)
    model0 = cd.ContDiscreteLinearGaussianSSM(state_dim=d, emission_dim=m, has_emissions_bias=True)
    params0, _ = model0.initialize(
        initial_mean={"params": lm.m0, "props": pp}, initial_cov={"params": lm.P0, "props": pp},
        dynamics_weights={"params": lm.drift.W, "props": pp}, dynamics_diffusion_coefficient={"params": lm.L, "props": pp},
        dynamics_diffusion_cov={"params": lm.Qc, "props": pp}, emission_weights={"params": lm.H, "props": pp},
        emission_bias={"params": lm.bias, "props": pp}, emission_cov={"params": lm.R, "props": pp})
    y0 = o.simulate(lm, t, rng)
    ref1 = o.kf_smoother_type1(lm, t, y0)
    p1 = model0.smoother(params0, y0, t[..., None])
    note("kf_smoother1", max(relerr(getattr(p1, f), ref1[f]) for f in ("filtered_means", "smoothed_means", "smoothed_covariances", "smoothed_cross_covariances")) if T > 1 else 0.0, 1e-7, tag)
    ref2 = o.ekf_smoother(lm, t, y0, state_order="first")
    p2 = model0.smoother(params0, y0, t[..., None], smoother_type="cd_smoother_2")
    note("kf_smoother2", max(relerr(p2.smoothed_means, ref2["smoothed_means"]), relerr(p2.smoothed_covariances, ref2["smoothed_covariances"])), 1e-8, tag)
print("fuzz solvers seed", seed, "cases", cases, "worst", {k: float(f"{v:.3g}") for k, v in worst.items()}, "kernels", kernels, flush=True)
