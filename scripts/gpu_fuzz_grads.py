"""Fuzz of the gradient sweeps for state / emission dimensions up to eight (forward sensitivities, lane-grid reverse sweep, wavefront
reverse sweep) against the oracle's discrete adjoint: python3 scripts/gpu_fuzz_grads.py [seed] [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi
from helpers import linear_model, mlp_model, params_from, relerr

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = np.random.default_rng(seed)
L = _ffi.lib()

def spd(n, s):
    A = rng.standard_normal((n, n))
    return A @ A.T / n * s + 0.3 * np.eye(n)

worst, kernels = {}, {}
for case in range(cases):
    kind = rng.choice(["linear", "lorenz63", "lorenz96", "mlp"], p=[0.25, 0.25, 0.2, 0.3])
    if kind == "linear":
        d = int(rng.integers(1, 9)); drift = linear_model(rng, d, 1).drift
    elif kind == "lorenz63":
        d = 3; drift = o.Lorenz63Drift(10 + rng.standard_normal(), 28 + rng.standard_normal(), 8 / 3)
    elif kind == "lorenz96":
        d = int(rng.integers(4, 9)); drift = o.Lorenz96Drift(8.0 + 0.5 * rng.standard_normal())
    else:
        d = int(rng.integers(1, 9)); drift = mlp_model(rng, d, 1, (int(rng.integers(1, 65)), int(rng.integers(1, 65)))).drift
    m = int(rng.integers(1, min(d, 8) + 1)) if rng.random() < 0.7 else int(rng.integers(1, 9))
    plain = rng.random() < 0.4    # the tutorial-like model: H = I[:m], identity noise
    if plain and m <= d:
        mdl = o.Model(drift, np.eye(d), np.eye(d), np.eye(d)[:m], np.zeros(m), np.eye(m), ({"lorenz96": 8.0}.get(kind, 0.0)) + np.zeros(d), (5.0 if kind == "lorenz63" else 1.0) * np.eye(d))
    else:
        mdl = o.Model(drift, np.eye(d) + 0.2 * rng.standard_normal((d, d)), spd(d, 0.5), rng.standard_normal((m, d)), 0.1 * rng.standard_normal(m), spd(m, 0.7),
                      ({"lorenz96": 8.0}.get(kind, 0.0)) + rng.standard_normal(d), spd(d, 1.0))
    LONG = os.environ.get("CDKF_FUZZ_LONG")  # long scans and long intervals: checkpoint windows, replay chunks, accumulation
    N, T = (int(rng.choice([1, 2, 5])), int(rng.integers(40, 160))) if LONG else (int(rng.choice([1, 3, 17, 66])), int(rng.integers(1, 9)))
    t = o.irregular_times(rng, N, T, 0.012 * T * rng.choice([1, 1, 5]))
    if LONG and T > 10:
        for _ in range(3):
            t[:, int(rng.integers(1, T)):] += rng.uniform(0.1, 1.5)   # a few intervals of 10 .. 150 steps
    if T > 3 and rng.random() < 0.3:
        t[0, 2] = t[0, 1]
    y = o.simulate(mdl, t, rng)
    order = str(rng.choice(["second", "first"]))
    tag = f"{kind} d={d} m={m} N={N} T={T} {order} plain={plain}"
    hyp = cd.EKFHyperParams(state_order=order)
    try:
        ll_ref, g_ref, ex = o.ekf_loglik_grad_adjoint(mdl, t, y, full=True, state_order=order)
    except np.linalg.LinAlgError:
        continue  # (the moment equations of this random model blow up over a long interval: nothing to compare)
    if not (np.isfinite(g_ref).all() and np.isfinite(ll_ref).all()):
        continue
    P = params_from(mdl)
    def note(name, e, tol):
        worst[name] = max(worst.get(name, 0.0), e)
        if not (e < tol):
            print("MISMATCH", name, tag, e, L.cdkf_last_kernel().decode()[:60], flush=True)
    try:
        ll, g = cd.cdnlgssm_loglik_and_grad(P, y, t[..., None], hyp)
        k = L.cdkf_last_kernel().decode().split("<")[0]; kernels[k] = kernels.get(k, 0) + 1
        flat = np.concatenate([np.asarray(a_).reshape(N, -1) for a_ in g], axis=-1)
        note("drift", max(np.abs(flat - g_ref).max() / (np.abs(g_ref).max() + 1e-300), relerr(ll, ll_ref)), 1e-7)
        ll32, g32 = cd.cdnlgssm_loglik_and_grad(P, y.astype(np.float32), t[..., None].astype(np.float32), hyp)
        flat32 = np.concatenate([np.asarray(a_).reshape(N, -1) for a_ in g32], axis=-1)
        if not LONG:  # (a chaotic flow amplifies fp32 rounding over long scans: no bound to hold it to)
            note("drift32", np.abs(flat32 - g_ref).max() / (np.abs(g_ref).max() + 1e-300), 3e-2)
    except NotImplementedError:
        pass
    try:
        ll, g = cd.cdnlgssm_loglik_and_grad_all(P, y, t[..., None], hyp)
        k = L.cdkf_last_kernel().decode().split("<")[0]; kernels[k] = kernels.get(k, 0) + 1
        flat = np.concatenate([np.asarray(a_).reshape(N, -1) for a_ in g.dynamics.drift], axis=-1)
        pairs = [(flat, g_ref), (g.initial.mean.params, ex["m0"]), (g.initial.cov.params, ex["P0"]), (g.dynamics.diffusion_coefficient.params, ex["L"]),
                 (g.dynamics.diffusion_cov.params, ex["Qc"]), (g.emissions.emission_function.weights, ex["H"]),
                 (g.emissions.emission_function.bias, ex["bias"]), (g.emissions.emission_cov.params, ex["R"])]
        note("all", max(np.abs(np.asarray(a_) - b_).max() / (np.abs(b_).max() + 1e-300) for a_, b_ in pairs), 1e-7)
    except NotImplementedError:
        pass
print("fuzz grads seed", seed, "cases", cases, "worst", {k: float(f"{v:.3g}") for k, v in worst.items()}, "kernels", kernels, flush=True)
