"""Fuzz of the filters / smoother / gradients of models whose EMISSION is given as source above the register-resident kernels' six
dimensions (round 5: the tangent kernels' value mode + the workgroup backward sweep): random (d, m) with max(d, m) in 7 .. 16, Lorenz-96
or linear drift, random emission parameters, both filters (extended: random state_order / num_iter), the smoother, the emission moments of
the filtered marginals (both reference versions), fp64 against the oracle at 1e-9; d ll / d eta against central differences of the value mode.
python3 scripts/gpu_fuzz_wide_emission.py [seed] [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi
from test_wide_emission import wide_emission, KEYS, sigma_point_emission_moments

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = np.random.default_rng(seed)
L = _ffi.lib()
worst, kernels = {}, {}


def rel(a, b):
    """Relative error; where the oracle is NaN (a sigma-point covariance that lost positive definiteness under a negative centre weight:
    jnp.linalg.cholesky's NaN, inference_ukf.py:57) the kernel must be NaN in the same places."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if np.isnan(b).any() or np.isnan(a).any():
        if not np.array_equal(np.isnan(a), np.isnan(b)):
            return float("inf")
        ok = ~np.isnan(b)
        if not ok.any():
            return 0.0
        a, b = a[ok], b[ok]
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))


def note(name, e, tol, tag):
    worst[name] = max(worst.get(name, 0.0), float(e))
    k = L.cdkf_last_kernel().decode().split("<")[0]
    kernels[k] = kernels.get(k, 0) + 1
    if not (e < tol):
        print("MISMATCH", name, tag, e, L.cdkf_last_kernel().decode()[:70], flush=True)


def spd(n, s):
    A = rng.standard_normal((n, n))
    return A @ A.T / n * s + 0.4 * np.eye(n)


for case in range(cases):
    while True:
        d, m = int(rng.integers(4, 17)), int(rng.integers(1, 17))
        if max(d, m) > 6:
            break
    linear = rng.random() < 0.4
    src, em = wide_emission(d, m)
    eta = np.concatenate([0.5 + rng.random(m), 0.05 * rng.standard_normal(m)])
    if eta.size > m * d + m:
        continue
    full = np.concatenate([eta, np.zeros(m * d + m - eta.size)])
    if linear:
        W, b = -0.6 * np.eye(d) + 0.3 * rng.standard_normal((d, d)) / np.sqrt(d), 0.2 * rng.standard_normal(d)
        drift, ldrift, scale = o.LinearDrift(W, b), cd.LearnableLinear(W, b), 0.0
    else:
        drift, ldrift, scale = o.Lorenz96Drift(8.0), cd.LearnableLorenz96(8.0), 8.0
    mdl = o.Model(drift, np.eye(d) + 0.1 * rng.standard_normal((d, d)), spd(d, 0.3), full[:m * d].reshape(m, d), full[m * d:], spd(m, 0.5),
                  scale + 0.5 * rng.standard_normal(d), spd(d, 0.5), emission=em)
    N, T = int(rng.integers(1, 5)), int(rng.integers(2, 9))
    t = o.irregular_times(rng, N, T, 0.02 * T * float(rng.choice([1, 3])))
    y = mdl.h(mdl.m0 + np.cumsum(rng.standard_normal((N, T, d)) * 0.3, axis=1)) + rng.standard_normal((N, T, m))
    P = cd.ParamsCDNLGSSM(
        initial=cd.ParamsLGSSMInitial(cd.LearnableVector(mdl.m0), cd.LearnableMatrix(mdl.P0)),
        dynamics=cd.ParamsCDNLGSSMDynamics(ldrift, cd.LearnableMatrix(mdl.L), cd.LearnableMatrix(mdl.Qc), 2.0),
        emissions=cd.ParamsCDNLGSSMEmissions(cd.LearnableCustomEmission(eta, src, None), cd.LearnableMatrix(mdl.R)))
    order = "second" if (d <= 8 and rng.random() < 0.5) else "first"
    num_iter = int(rng.integers(1, 4))
    tag = f"case {case} d={d} m={m} {'linear' if linear else 'l96'} N={N} T={T} order={order} num_iter={num_iter}"
    hyper = cd.EKFHyperParams(state_order=order)
    ref = o.ekf_filter(mdl, t, y, order, num_iter)
    post = cd.cdnlgssm_filter(P, y, t[..., None], hyper, num_iter=num_iter)
    note("ekf", max([rel(post.marginal_loglik, ref["marginal_loglik"])] + [rel(getattr(post, k), ref[k]) for k in KEYS]), 1e-9, tag)
    alpha, beta, kappa = float(rng.choice([np.sqrt(3), 1.0, 0.5])), float(rng.choice([2.0, 0.0])), float(rng.choice([1.0, 0.0, 3.0 - d]))
    if alpha * alpha * (d + kappa) <= 0.05:
        kappa = 1.0
    refu = o.ukf_filter(mdl, t, y, alpha, beta, kappa)
    postu = cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams(alpha=alpha, beta=beta, kappa=kappa))
    note("ukf", max([rel(postu.marginal_loglik, refu["marginal_loglik"])] + [rel(getattr(postu, k), refu[k]) for k in KEYS]), 1e-8, tag + f" a={alpha:.2f} b={beta} k={kappa}")
    refs = o.ekf_smoother(mdl, t, y, order)
    posts = cd.cdnlgssm_smoother(P, y, t[..., None], hyper)
    note("eks", max(rel(getattr(posts, k), refs[k]) for k in ("filtered_means", "filtered_covariances", "smoothed_means", "smoothed_covariances")), 1e-9, tag)
    # emission moments of the filtered marginals, both reference versions (cdkf_custom_emission_moments_*)
    fm, fc = np.asarray(post.filtered_means).reshape(-1, d), np.asarray(post.filtered_covariances).reshape(-1, d, d)
    ym, yc = cd.cdnlgssm_emissions(P, np.zeros((fm.shape[0], 1)), fm, fc, hyperparams=hyper)
    Hj = mdl.Hjac(fm)
    note("emis_ekf", max(rel(ym, mdl.h(fm)), rel(yc, Hj @ fc @ np.swapaxes(Hj, -1, -2) + mdl.R)), 1e-11, tag)
    if not np.isnan(np.asarray(postu.filtered_covariances)).any():
        ym, yc = cd.cdnlgssm_emissions(P, np.zeros((fm.shape[0], 1)), fm, fc, hyperparams=cd.UKFHyperParams(alpha=alpha, beta=beta, kappa=kappa))
        rm, rc = sigma_point_emission_moments(mdl, fm, fc, alpha, beta, kappa)
        note("emis_ukf", max(rel(ym, rm), rel(yc, rc)), 1e-8, tag + f" a={alpha:.2f} b={beta} k={kappa}")
    if case % 3 == 0:
        ll, g = cd.cdnlgssm_loglik_and_grad_all(P, y, t[..., None], cd.EKFHyperParams(state_order="first"), num_iter=num_iter)
        got = np.asarray(g.emissions.emission_function.eta)
        pidx = int(rng.integers(0, eta.size))
        e = np.zeros_like(eta)
        e[pidx] = 1e-5
        w = lambda ev: P._replace(emissions=cd.ParamsCDNLGSSMEmissions(cd.LearnableCustomEmission(ev, src, None), P.emissions.emission_cov))
        ll_of = lambda Pv: np.asarray(cd.cdnlgssm_filter(Pv, y, t[..., None], cd.EKFHyperParams(state_order="first"), num_iter=num_iter).marginal_loglik)
        fd = (8 * (ll_of(w(eta + e)) - ll_of(w(eta - e))) - (ll_of(w(eta + 2 * e)) - ll_of(w(eta - 2 * e)))) / (12 * 1e-5)   # (five-point stencil: eta multiplies x x ~ 64 -- the two-point one is 1.6e-6 off at h = 1e-5, the kernel 3e-11)
        note("grad_eta", float(np.abs(got[:, pidx] - fd).max() / max(1.0, np.abs(fd).max())), 1e-6, tag + f" eta[{pidx}]")
print("fuzz wide emission seed", seed, "cases", cases, "worst", {k: float(f"{v:.3g}") for k, v in worst.items()}, "kernels", kernels, flush=True)
