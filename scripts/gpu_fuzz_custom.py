"""Fuzz of the drifts given as source (dev aid, not part of the suite): random sparse quadratic drifts
f = c - x + theta_0 B x + theta_1 Q(x), Q_i = sum a_i,jk x_j x_k, generated as C statements of random shape (loops, temporaries, pow, products),
any state / emission dimension the kernels take -- register-resident kernels up to six, workgroup kernels beyond -- against the oracle
running the same drift from its coefficient arrays: EKF orders, UKF, smoother, forecast, every gradient:
python3 scripts/gpu_fuzz_custom.py [seed] [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi
from helpers import FILTER_KEYS, random_quadratic_drift, relerr

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rng = np.random.default_rng(seed)
worst = {}


def note(name, e, tol, tag):
    worst[name] = max(worst.get(name, 0.0), float(e))
    if not (e < tol):
        print("MISMATCH", name, tag, e, _ffi.lib().cdkf_last_kernel().decode()[:60], flush=True)


def spd(n, s):
    A = rng.standard_normal((n, n))
    return A @ A.T / n * s + 0.3 * np.eye(n)


for case in range(cases):
    big = rng.random() < 0.7
    dmax = int(os.environ.get("CDKF_FUZZ_DMAX", "24"))
    d = int(rng.integers(7, dmax + 1)) if big else int(rng.integers(1, 7))
    m = int(rng.integers(1, (min(d + 4, dmax) if big else 7)))
    if not big and rng.random() < 0.3:
        m = int(rng.integers(7, 12))          # small state, wide emission: the workgroup kernels too
    if os.environ.get("CDKF_FUZZ_D"):
        d = int(os.environ["CDKF_FUZZ_D"])
        m = min(m, d + 4)
    src, make = random_quadratic_drift(rng, d)
    if os.environ.get("CDKF_FUZZ_NOPOW"):  # (debugging aid: the same drift without pow(), or -- "2.0" -- with the library's pow)
        import re
        src = re.sub(r"pow\((x\[\d+\]), 2\)", r"pow(\1, 2.0)" if os.environ["CDKF_FUZZ_NOPOW"] == "2.0" else r"(\1 * \1)", src)
    theta = np.array([0.5 + 0.5 * rng.random(), 0.2 * rng.standard_normal()])
    if rng.random() < 0.4 and m <= d:
        H, bias = np.eye(d)[rng.permutation(d)[:m]], np.zeros(m)
    else:
        H, bias = rng.standard_normal((m, d)) / np.sqrt(d), 0.1 * rng.standard_normal(m)
    mdl = o.Model(make(theta), np.eye(d) + 0.1 * rng.standard_normal((d, d)), spd(d, 0.3), H, bias, spd(m, 0.5), 0.5 * rng.standard_normal(d), spd(d, 0.3))
    N, T = int(rng.integers(1, 5)), int(rng.integers(2, 9))
    t = o.irregular_times(rng, N, T, 0.03 * T)
    y = o.simulate(mdl, t, rng)
    mk = lambda g_: cd.ParamsCDNLGSSM(
        initial=cd.ParamsLGSSMInitial(cd.LearnableVector(mdl.m0), cd.LearnableMatrix(mdl.P0)),
        dynamics=cd.ParamsCDNLGSSMDynamics(cd.LearnableCustomDrift(theta, src, None, g_), cd.LearnableMatrix(mdl.L), cd.LearnableMatrix(mdl.Qc), 2.0),
        emissions=cd.ParamsCDNLGSSMEmissions(cd.LearnableLinear(mdl.H, mdl.bias), cd.LearnableMatrix(mdl.R)))
    P = mk("auto")
    tag = (case, d, m, N, T)
    order = str(rng.choice(["second", "first", "zeroth"]))
    only_grad = bool(os.environ.get("CDKF_FUZZ_ONLY_GRAD"))
    if os.environ.get("CDKF_FUZZ_ONLY_CASE") and int(os.environ["CDKF_FUZZ_ONLY_CASE"]) != case:
        rng.random(); rng.random(); rng.random()   # (the three draws below: which checks run, the gradient's state order)
        continue
    ref = o.ekf_filter(mdl, t, y, state_order=order)
    post = cd.cdnlgssm_filter(P, y, t[..., None], cd.EKFHyperParams(state_order=order)) if not only_grad else None
    if post is not None:
        note("ekf", max(max(relerr(getattr(post, k), ref[k]) for k in FILTER_KEYS), relerr(post.marginal_loglik, ref["marginal_loglik"])), 1e-9, tag + (order,))
    if rng.random() < 0.5 and not only_grad:
        ref = o.ukf_filter(mdl, t, y)
        post = cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams())
        note("ukf", max(relerr(getattr(post, k), ref[k]) for k in FILTER_KEYS), 1e-8, tag)
        if os.environ.get("CDKF_FUZZ_VERBOSE") and not (max(relerr(getattr(post, k), ref[k]) for k in FILTER_KEYS) < 1e-8):
            print("  ukf per field:", {k: relerr(getattr(post, k), ref[k]) for k in FILTER_KEYS}, "ll", np.asarray(post.marginal_loglik).tolist(), ref["marginal_loglik"].tolist())
            for n_ in range(N):
                e_t = [float(np.abs(np.asarray(post.filtered_means)[n_, k_] - ref["filtered_means"][n_, k_]).max()) for k_ in range(T)]
                print("   trajectory", n_, "per-step mean error", ["%.1e" % v for v in e_t], "min eig filtered cov", ["%.1e" % np.linalg.eigvalsh(ref["filtered_covariances"][n_, k_]).min() for k_ in range(T)])
            print("  source:\n" + src, flush=True)
    if rng.random() < 0.5 and not only_grad:
        ref = o.ekf_smoother(mdl, t, y, state_order="second")
        sm = cd.cdnlgssm_smoother(P, y, t[..., None])
        note("eks", max(relerr(sm.smoothed_means, ref["smoothed_means"]), relerr(sm.smoothed_covariances, ref["smoothed_covariances"])), 1e-8, tag)
    # gradients: state_order 'first', every leaf on the reverse sweep; the drift block alone (forward sensitivities up to six dimensions)
    gorder = "second" if rng.random() < 0.5 else "first"     # 'second': grad(div f) "auto" -- third derivatives in the reverse sweep
    ll_ref, g_ref, ex = o.ekf_loglik_grad_adjoint(mdl, t, y, full=True, state_order=gorder)
    hyp = cd.EKFHyperParams(state_order=gorder)
    Pn = mk("auto") if gorder == "second" else mk(None)
    ll, g = cd.cdnlgssm_loglik_and_grad_all(Pn, y, t[..., None], hyp)
    pairs = [(g.dynamics.drift.theta, g_ref), (g.initial.mean.params, ex["m0"]), (g.initial.cov.params, ex["P0"]),
             (g.dynamics.diffusion_coefficient.params, ex["L"]), (g.dynamics.diffusion_cov.params, ex["Qc"]),
             (g.emissions.emission_function.weights, ex["H"]), (g.emissions.emission_function.bias, ex["bias"]), (g.emissions.emission_cov.params, ex["R"])]
    scale = max(np.abs(b_).max() for _, b_ in pairs)
    tag = tag + (gorder,)
    note("grad_all", max(max(np.abs(np.asarray(a_) - b_).max() / scale for a_, b_ in pairs), relerr(ll, ll_ref)), 1e-7, tag)
    ll, g1 = cd.cdnlgssm_loglik_and_grad(Pn, y, t[..., None], hyp)
    note("grad_theta", np.abs(np.asarray(g1.theta) - g_ref).max() / (np.abs(g_ref).max() + 1e-300), 1e-7, tag)
    if os.environ.get("CDKF_FUZZ_VERBOSE") and not (np.abs(np.asarray(g1.theta) - g_ref).max() / (np.abs(g_ref).max() + 1e-300) < 1e-7):
        print("  source:\n" + src + "\n  theta", theta.tolist(), "H", mdl.H.tolist())
        print("  drift-only:", np.asarray(g1.theta).tolist(), "\n  all-leaf:  ", np.asarray(g.dynamics.drift.theta).tolist(), "\n  oracle:    ", g_ref.tolist(),
              "\n  ll", np.asarray(ll).tolist(), ll_ref.tolist(), flush=True)
    if os.environ.get("CDKF_FUZZ_VERBOSE"):
        print(tag, "theta grad got", np.asarray(g1.theta).ravel(), "all-leaf", np.asarray(g.dynamics.drift.theta).ravel(), "want", g_ref.ravel(), flush=True)
print("fuzz custom seed", seed, "cases", cases, "worst", {k: float("%.3g" % v) for k, v in worst.items()}, flush=True)
