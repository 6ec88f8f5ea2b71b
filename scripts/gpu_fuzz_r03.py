"""Round-3 fuzz (dev aid, not part of the suite): the workgroup reverse sweep and the partially observed wavefront Lorenz-96 sweeps on
random problems with run-dependent seeds: python3 scripts/gpu_fuzz_r03.py [seed] [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi
from helpers import linear_model, params_from, relerr

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rng = np.random.default_rng(seed)
worst = {"adj": 0.0, "w40": 0.0}

def spd(n, s):
    A = rng.standard_normal((n, n))
    return A @ A.T / n * s + 0.3 * np.eye(n)

for case in range(cases):
    # ---- workgroup reverse sweep ----
    lin = rng.random() < 0.3
    d = int(rng.integers(9, 18)) if lin else int(rng.integers(9, 44))
    m = int(rng.integers(1, min(d, 43) + 1))
    drift = linear_model(rng, d, m).drift if lin else o.Lorenz96Drift(8.0 + rng.standard_normal())
    if rng.random() < 0.4:
        H, bias = np.eye(d)[rng.permutation(d)[:m]], np.zeros(m)
    else:
        H, bias = rng.standard_normal((m, d)) / np.sqrt(d), 0.1 * rng.standard_normal(m)
    mdl = o.Model(drift, np.eye(d) + 0.1 * rng.standard_normal((d, d)), spd(d, 1.0), H, bias, spd(m, 1.0),
                  (0.0 if lin else 8.0) + rng.standard_normal(d), spd(d, 1.0))
    N, T = int(rng.integers(1, 4)), int(rng.integers(1, 7))
    t = o.irregular_times(rng, N, T, 0.02 * T)
    if T > 2:
        t[:, 2:] += rng.uniform(0.0, 0.25)
    y = o.simulate(mdl, t, rng)
    ll_ref, g_ref, ex = o.ekf_loglik_grad_adjoint(mdl, t, y, full=True)
    ll, g = cd.cdnlgssm_loglik_and_grad_all(params_from(mdl), y, t[..., None], cd.EKFHyperParams(state_order="first"))
    # (round 4: Lorenz-96 through a selection of components at a wavefront-kernel width and a bitwise symmetric R takes ekf_adjoint_wave2_l96_kernel)
    kern = _ffi.lib().cdkf_last_kernel().decode()
    assert kern.startswith(("ekf_adjoint_wg_kernel<double", "ekf_adjoint_wave2_l96_kernel<double", "ekf_adjoint_wave_l96_kernel<double")), (case, kern)
    kinds = globals().setdefault("kinds", {})
    kinds[kern.split("<")[0]] = kinds.get(kern.split("<")[0], 0) + 1
    flat = np.concatenate([np.asarray(a_).reshape(N, -1) for a_ in g.dynamics.drift], axis=-1)
    pairs = [(flat, g_ref), (g.initial.mean.params, ex["m0"]), (g.initial.cov.params, ex["P0"]), (g.dynamics.diffusion_coefficient.params, ex["L"]),
             (g.dynamics.diffusion_cov.params, ex["Qc"]), (g.emissions.emission_function.weights, ex["H"]),
             (g.emissions.emission_function.bias, ex["bias"]), (g.emissions.emission_cov.params, ex["R"])]
    e = max(np.abs(np.asarray(a_) - b_).max() / (np.abs(b_).max() + 1e-300) for a_, b_ in pairs)
    e = max(e, relerr(ll, ll_ref))
    worst["adj"] = max(worst["adj"], e)
    if e > 1e-7:
        print("MISMATCH adj", case, d, m, lin, N, T, e, flush=True)
    # ---- wavefront Lorenz-96, a random selection of components ----
    d = int(rng.choice([12, 16, 20, 24, 28, 32, 36, 40]))
    m = int(rng.integers(1, d + 1))
    H = np.eye(d)[np.sort(rng.permutation(d)[:m]) if rng.random() < 0.5 else rng.permutation(d)[:m]]
    mdl = o.Model(o.Lorenz96Drift(8.0), np.eye(d) + 0.1 * rng.standard_normal((d, d)), spd(d, 0.5), H, np.zeros(m), spd(m, 0.7),
                  8.0 + rng.standard_normal(d), spd(d, 1.0))
    N, T = int(rng.integers(1, 6)), int(rng.integers(1, 10))
    t = o.irregular_times(rng, N, T, 0.015 * T)
    y = o.simulate(mdl, t, rng)
    ref = o.ekf_smoother(mdl, t, y)
    post = cd.cdnlgssm_smoother(params_from(mdl), y, t[..., None])
    assert _ffi.lib().cdkf_last_kernel().decode().startswith("ekf_smoother_wave_l96_kernel<double, %d>" % d), case
    e = max(relerr(getattr(post, k), ref[k]) for k in ("filtered_means", "filtered_covariances", "smoothed_means", "smoothed_covariances"))
    e = max(e, relerr(post.marginal_loglik, ref["marginal_loglik"]))
    worst["w40"] = max(worst["w40"], e)
    if e > 1e-8:
        print("MISMATCH w40", case, d, m, N, T, e, flush=True)
print("fuzz seed", seed, "cases", cases, "worst", worst, "reverse sweeps", globals().get("kinds", {}), flush=True)
