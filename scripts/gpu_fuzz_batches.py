"""Fuzz of the batch-size dependent mappings and of the layouts through the C ABI: batch sizes around every wavefront / grouping
boundary, the three layouts, shared and per-trajectory time grids, output subsets: python3 scripts/gpu_fuzz_batches.py [seed] [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi, models
from helpers import linear_model, lorenz96_model, mlp_model, params_from, relerr
from test_gpu_soak import _run

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rng = np.random.default_rng(seed)
L = _ffi.lib()
worst, kernels = {}, {}
KEYS = ("filtered_means", "filtered_covariances", "predicted_means", "predicted_covariances")
for case in range(cases):
    kind = rng.choice(["lorenz63", "linear3", "linear", "lorenz96", "mlp"], p=[0.3, 0.15, 0.2, 0.2, 0.15])
    if kind == "lorenz63":
        mdl = o.lorenz63_model(int(rng.integers(1, 4)))
    elif kind == "linear3":
        mdl = linear_model(rng, 3, int(rng.integers(1, 4)))
    elif kind == "linear":
        mdl = linear_model(rng, int(rng.integers(1, 7)), int(rng.integers(1, 7)))
    elif kind == "lorenz96":
        d = int(rng.choice([4, 6, 8, 12])); mdl = lorenz96_model(d, int(rng.integers(1, d + 1)))
    else:
        d = int(rng.integers(1, 9)); mdl = mlp_model(rng, d, int(rng.integers(1, d + 1)), (int(rng.integers(1, 33)), int(rng.integers(1, 33))))
    big = mdl.d <= 4
    N = int(rng.choice([1, 2, 3, 4, 5, 15, 16, 17, 63, 64, 65, 127, 129, 255, 257, 1023, 1025, 4095, 4097, 8191, 8193, 9000] if big else [1, 3, 4, 5, 63, 65, 130]))
    T = int(rng.integers(1, 6))
    shared = rng.random() < 0.3
    t = o.irregular_times(rng, 1 if shared else N, T, 0.012 * T * rng.choice([1, 3]))
    tt = np.broadcast_to(t, (N, T)) if shared else t
    y = o.simulate(mdl, tt, rng)
    layout = int(rng.choice([_ffi.LAYOUT_NT, _ffi.LAYOUT_TN, _ffi.LAYOUT_TCN]))
    dtype = np.float64 if rng.random() < 0.7 else np.float32
    algo = str(rng.choice(["ekf_filter", "ukf_filter", "ekf_smoother"]))
    order = str(rng.choice(["second", "first"]))
    opts = _ffi.default_opts()
    opts.t_shared = 1 if shared else 0
    opts.state_order = _ffi.ORDER[order]
    want = tuple(bool(b) for b in rng.integers(0, 2, 4)) if algo != "ekf_smoother" else (True,) * 4
    tag = f"{kind} d={mdl.d} m={mdl.m} N={N} T={T} {algo} {order} layout={layout} shared={shared} {dtype.__name__} want={want}"
    try:
        if algo == "ekf_filter":
            ref = o.ekf_filter(mdl, tt, y, state_order=order); keys = KEYS
        elif algo == "ukf_filter":
            ref = o.ukf_filter(mdl, tt, y); keys = KEYS
        else:
            ref = o.ekf_smoother(mdl, tt, y, state_order=order); keys = ("filtered_means", "filtered_covariances", "smoothed_means", "smoothed_covariances")
        if not np.isfinite(ref[keys[0]]).all():
            continue
        ll, outs, st = _run(algo, mdl, opts, layout, t[0] if shared else t, y, dtype, want=want)
    except (NotImplementedError, _ffi.CdkfError) as e:
        if not isinstance(e, NotImplementedError) and getattr(e, "code", 0) != -2:
            print("ERROR", tag, e, flush=True)
        continue
    k = L.cdkf_last_kernel().decode().split("<")[0]; kernels[k] = kernels.get(k, 0) + 1
    tol = 1e-8 if dtype == np.float64 else 5e-3
    errs = [relerr(a, ref[kk]) for a, kk in zip(outs, keys) if a is not None] + [relerr(ll, ref["marginal_loglik"])]
    e = max(errs)
    name = algo + ("32" if dtype == np.float32 else "")
    worst[name] = max(worst.get(name, 0.0), e)
    if not (e < tol) or st.any():
        print("MISMATCH", tag, e, "status", st[st != 0][:3], L.cdkf_last_kernel().decode()[:50], flush=True)
    for a, w in zip(outs, want):
        assert (a is None) == (not w)
print("fuzz batches seed", seed, "cases", cases, "worst", {k: float(f"{v:.3g}") for k, v in worst.items()}, "kernels", kernels, flush=True)
