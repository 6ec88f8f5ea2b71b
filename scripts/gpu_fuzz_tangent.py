"""Fuzz of the tangent sweeps of the literal recursions (cdkf_ukf_tangent_kernels.h; round 5): random models -- MLP with ragged hidden
sizes, Lorenz-96, linear, Lorenz-63, random sparse quadratic drifts given as source -- random dimensions up to ten, dense non-diagonal
model matrices, random UKF hyper-parameters / EKF state orders and update iterations; EVERY leaf of the gradient against the oracle:
ukf_loglik_grad_all_literal (forward tangents) for the unscented filter, ekf_loglik_grad_adjoint (reverse mode) for the extended one.
CDKF_UKF_GRAD_TANGENT=1 is set so that the models the closed forms cover go through the tangent sweep as well; the extended filter's leg
uses shapes no other gradient kernel takes (update iterations above eight dimensions, wide MLP layers).
python3 scripts/gpu_fuzz_tangent.py [seed] [cases]"""
import os, sys
os.environ["CDKF_UKF_GRAD_TANGENT"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi
from helpers import params_from, random_quadratic_drift

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rng = np.random.default_rng(seed)
L = _ffi.lib()
worst, kernels = {}, {}


def note(name, e, tol, tag):
    worst[name] = max(worst.get(name, 0.0), float(e))
    if not (e < tol):
        print("MISMATCH", name, tag, e, L.cdkf_last_kernel().decode()[:60], flush=True)


def spd(n, s):
    A = rng.standard_normal((n, n))
    return A @ A.T / n * s + 0.3 * np.eye(n)


def rel(a, b):
    return float(np.abs(np.asarray(a) - b).max() / (np.abs(b).max() + 1e-300))


def with_dtheta(drift):
    f = drift._f
    def dth(x, th, *extra):
        base = f(x, th, *extra)
        return np.stack([f(x, th + np.eye(th.size)[p], *extra) - base for p in range(th.size)], axis=1)
    return o.CallableDrift(drift.th, drift._f, drift._jac, drift._g, vjp=drift._vjp, gvjp=drift._gvjp, ut=drift.ut, dtheta=dth)


for case in range(cases):
    ekf = rng.random() < 0.5
    kind = str(rng.choice(["mlp", "lorenz96", "linear", "lorenz63", "source"]))
    d = 3 if kind == "lorenz63" else int(rng.integers(4, 11)) if kind == "lorenz96" else int(rng.integers(1, 11))
    if ekf and kind in ("lorenz96", "linear", "lorenz63"):
        d = max(d, 9) if kind != "lorenz63" else d      # (below nine dimensions these have reverse sweeps: pick what only the tangent sweep takes)
    m = int(rng.integers(1, min(d, 6) + 1))
    src = None
    theta = None
    if kind == "mlp":
        h1, h2 = (int(rng.integers(65, 90)), int(rng.integers(1, 9))) if ekf else (int(rng.integers(1, 12)), int(rng.integers(1, 12)))
        W = lambda a, b: rng.standard_normal((a, b)) / np.sqrt(b)
        drift = o.MLPDrift(W(h1, d), 0.1 * rng.standard_normal(h1), W(h2, h1), 0.1 * rng.standard_normal(h2), W(d, h2), 0.1 * rng.standard_normal(d))
        scale = 0.0
    elif kind == "lorenz96":
        drift, scale = o.Lorenz96Drift(8.0), 8.0
    elif kind == "lorenz63":
        drift, scale = o.Lorenz63Drift(10.0, 28.0, 8.0 / 3.0), 1.0
    elif kind == "linear":
        drift, scale = o.LinearDrift(-0.6 * np.eye(d) + 0.3 * rng.standard_normal((d, d)) / np.sqrt(d), 0.2 * rng.standard_normal(d)), 0.0
    else:
        src, make = random_quadratic_drift(rng, d)
        theta = np.array([0.5 + 0.5 * rng.random(), 0.2 * rng.standard_normal()])
        drift, scale = with_dtheta(make(theta)), 0.0
    mdl = o.Model(drift, np.eye(d) + 0.1 * rng.standard_normal((d, d)), spd(d, 0.3), rng.standard_normal((m, d)) / np.sqrt(d), 0.1 * rng.standard_normal(m),
                  spd(m, 0.5), scale + 0.5 * rng.standard_normal(d), spd(d, 0.5))
    N, T = int(rng.integers(1, 4)), int(rng.integers(2, 8))
    t = o.irregular_times(rng, N, T, 0.02 * T * float(rng.choice([1, 3])))
    y = o.simulate(mdl, t, rng)
    if src is None:
        P = params_from(mdl)
    else:
        P0 = params_from(o.Model(o.LinearDrift(np.eye(d), np.zeros(d)), mdl.L, mdl.Qc, mdl.H, mdl.bias, mdl.R, mdl.m0, mdl.P0))
        P = P0._replace(dynamics=P0.dynamics._replace(drift=cd.LearnableCustomDrift(theta, src, None, None)))
    if ekf:
        num_iter = int(rng.integers(1, 4)) if kind != "mlp" else int(rng.integers(1, 3))
        if kind in ("lorenz63",) or (kind == "source" and d <= 6):
            num_iter = max(num_iter, 2) if d > 8 else num_iter
        order = "first"
        hyp = cd.EKFHyperParams(state_order=order)
        tag = f"ekf {kind} d={d} m={m} N={N} T={T} num_iter={num_iter}"
        try:
            ll_r, g_r, ex = o.ekf_loglik_grad_adjoint(mdl, t, y, full=True, state_order=order, num_iter=num_iter)
        except NotImplementedError:
            continue
        ll, g = cd.cdnlgssm_loglik_and_grad_all(P, y, t[..., None], hyp, num_iter=num_iter)
        name = "ekf"
    else:
        alpha, beta, kappa = float(rng.uniform(0.7, 2.0)), int(rng.integers(0, 4)), int(rng.integers(0, 3))
        hyp = cd.UKFHyperParams(alpha=alpha, beta=beta, kappa=kappa)
        tag = f"ukf {kind} d={d} m={m} N={N} T={T} a={alpha:.2f} b={beta} k={kappa}"
        with np.errstate(all="ignore"):
            ll_r, g_r, ex = o.ukf_loglik_grad_all_literal(mdl, t, y, alpha=alpha, beta=beta, kappa=kappa)
        if not np.isfinite(ll_r).all():
            continue
        ll, g = cd.cdnlgssm_loglik_and_grad_all(P, y, t[..., None], hyp)
        name = "ukf"
    k = L.cdkf_last_kernel().decode().split("<")[0]
    kernels[k] = kernels.get(k, 0) + 1
    flat = np.concatenate([np.asarray(a).reshape(N, -1) for a in g.dynamics.drift], axis=-1) if src is None else np.asarray(g.dynamics.drift.theta)
    errs = [rel(ll, ll_r), rel(flat, g_r), rel(g.initial.mean.params, ex["m0"]), rel(g.initial.cov.params, ex["P0"]),
            rel(g.dynamics.diffusion_coefficient.params, ex["L"]), rel(g.dynamics.diffusion_cov.params, ex["Qc"]),
            rel(g.emissions.emission_function.weights, ex["H"]), rel(g.emissions.emission_function.bias, ex["bias"]), rel(g.emissions.emission_cov.params, ex["R"])]
    note(name, max(errs), 1e-7, tag)
print(f"fuzz tangent seed {seed} cases {cases} worst", {k: float(f"{v:.3g}") for k, v in worst.items()}, "kernels", kernels)
