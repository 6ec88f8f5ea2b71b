"""Fuzz of the filter / smoother / UKF kernels across every kernel family (dev aid): random drifts, shapes, emission matrices, orders,
update iterations, time grids; run-dependent seeds: python3 scripts/gpu_fuzz_filters.py [seed] [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi
from helpers import linear_model, mlp_model, params_from, relerr, FILTER_KEYS

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = np.random.default_rng(seed)
L = _ffi.lib()

def spd(n, s):
    A = rng.standard_normal((n, n))
    return A @ A.T / n * s + 0.3 * np.eye(n)

worst, kernels, unsupported = {}, {}, 0
for case in range(cases):
    kind = rng.choice(["linear", "lorenz63", "lorenz96", "mlp"], p=[0.3, 0.15, 0.35, 0.2])
    if kind == "linear":
        d = int(rng.integers(1, 21)); drift = linear_model(rng, d, 1).drift
    elif kind == "lorenz63":
        d = 3; drift = o.Lorenz63Drift(10 + rng.standard_normal(), 28 + rng.standard_normal(), 8 / 3)
    elif kind == "lorenz96":
        d = int(rng.integers(4, 49)); drift = o.Lorenz96Drift(8.0 + 0.5 * rng.standard_normal())
    else:
        d = int(rng.integers(1, 11)); drift = mlp_model(rng, d, 1, (int(rng.integers(1, 65)), int(rng.integers(1, 65)))).drift
    m = int(rng.integers(1, min(d + 3, 40) + 1))
    sel = m <= d and rng.random() < 0.5
    H = np.eye(d)[:m] if (sel and rng.random() < 0.5) else (np.eye(d)[rng.permutation(d)[:m]] if sel else rng.standard_normal((m, d)) / np.sqrt(d))
    bias = np.zeros(m) if sel else 0.1 * rng.standard_normal(m)
    scale = {"linear": 0.0, "lorenz63": 1.0, "lorenz96": 8.0, "mlp": 0.0}[kind]
    mdl = o.Model(drift, np.eye(d) + 0.1 * rng.standard_normal((d, d)), spd(d, 0.5), H, bias, spd(m, 0.7), scale + rng.standard_normal(d), spd(d, 1.0))
    LONG = os.environ.get("CDKF_FUZZ_LONG")  # long scans with a few long intervals
    N, T = int(rng.choice([1, 2, 5, 33, 70])), int(rng.integers(1, 10))
    if LONG: N, T = int(rng.choice([1, 2, 4])), int(rng.integers(60, 250))
    if d > 20: N = min(N, 5)
    t = o.irregular_times(rng, N, T, 0.012 * T * rng.choice([1, 1, 4]))
    if LONG:
        for _ in range(3):
            t[:, int(rng.integers(1, T)):] += rng.uniform(0.05, 0.8)
    y = o.simulate(mdl, t, rng)
    order = str(rng.choice(["second", "first", "zeroth"]))
    num_iter = int(rng.choice([1, 1, 2]))
    P = params_from(mdl)
    hyp = cd.EKFHyperParams(state_order=order)
    tag = f"{kind} d={d} m={m} N={N} T={T} {order} it={num_iter} sel={sel}"
    def note(name, e, tol):
        worst[name] = max(worst.get(name, 0.0), e)
        if not (e < tol):
            print("MISMATCH", name, tag, e, L.cdkf_last_kernel().decode()[:50], flush=True)
    try:
        ref = o.ekf_filter(mdl, t, y, state_order=order, num_iter=num_iter)
        if not np.isfinite(ref["filtered_means"]).all():
            continue
        post = cd.cdnlgssm_filter(P, y, t[..., None], hyp, num_iter=num_iter)
        k = L.cdkf_last_kernel().decode().split("<")[0]; kernels[k] = kernels.get(k, 0) + 1
        note("ekf", max([relerr(getattr(post, f), ref[f]) for f in FILTER_KEYS] + [relerr(post.marginal_loglik, ref["marginal_loglik"])]), 1e-8)
        p32 = cd.cdnlgssm_filter(P, y.astype(np.float32), t[..., None].astype(np.float32), hyp, num_iter=num_iter)
        if not LONG:
            note("ekf32", relerr(p32.filtered_means, ref["filtered_means"]), 5e-3)
    except NotImplementedError as e:
        unsupported += 1
    try:
        refs = o.ekf_smoother(mdl, t, y, state_order=order)
        sm = cd.cdnlgssm_smoother(P, y, t[..., None], hyp)
        k = L.cdkf_last_kernel().decode().split("<")[0]; kernels[k] = kernels.get(k, 0) + 1
        note("eks", max(relerr(sm.smoothed_means, refs["smoothed_means"]), relerr(sm.smoothed_covariances, refs["smoothed_covariances"])), 1e-7)
    except NotImplementedError:
        unsupported += 1
    if d <= 12:
        try:
            refu = o.ukf_filter(mdl, t, y)
            pu = cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams())
            k = L.cdkf_last_kernel().decode().split("<")[0]; kernels[k] = kernels.get(k, 0) + 1
            ok = np.isfinite(refu["filtered_means"]).all()
            if ok:
                note("ukf", max(relerr(pu.filtered_means, refu["filtered_means"]), relerr(pu.marginal_loglik, refu["marginal_loglik"])), 1e-7)
        except NotImplementedError:
            unsupported += 1
print("fuzz filters seed", seed, "cases", cases, "worst", {k: float(f"{v:.3g}") for k, v in worst.items()}, "unsupported", unsupported, "kernels", kernels, flush=True)
