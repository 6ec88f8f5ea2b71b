import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi
from helpers import mlp_model, params_from, relerr
L = _ffi.lib()
rng = np.random.default_rng(7)
for d, m, h in ((9, 2, (8, 8)), (9, 2, (40, 40)), (9, 2, (64, 64)), (9, 2, (3, 50)), (9, 2, (50, 3)), (10, 8, (17, 33)), (10, 8, (1, 1)), (6, 9, (20, 20)), (6, 9, (64, 5)),
                (12, 3, (64, 64)), (12, 3, (30, 10)), (9, 9, (33, 64))):
    mdl = mlp_model(rng, d, min(m, d), h)
    if m > d:
        H = rng.standard_normal((m, d)) / np.sqrt(d)
        mdl = o.Model(mdl.drift, mdl.L, mdl.Qc, H, np.zeros(m), 0.5 * np.eye(m), mdl.m0, mdl.P0)
    N, T = 2, 5
    t = o.irregular_times(rng, N, T, 0.012 * T)
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    out = []
    for order in ("first", "second"):
        ref = o.ekf_filter(mdl, t, y, state_order=order)
        post = cd.cdnlgssm_filter(P, y, t[..., None], cd.EKFHyperParams(state_order=order))
        k = L.cdkf_last_kernel().decode()[:24]
        refs = o.ekf_smoother(mdl, t, y, state_order=order)
        sm = cd.cdnlgssm_smoother(P, y, t[..., None], cd.EKFHyperParams(state_order=order))
        out.append(f"{order}: pm {relerr(post.predicted_means, ref['predicted_means']):.1e} fm {relerr(post.filtered_means, ref['filtered_means']):.1e} sm {relerr(sm.smoothed_means, refs['smoothed_means']):.1e}")
    refu = o.ukf_filter(mdl, t, y)
    pu = cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams())
    out.append(f"ukf {relerr(pu.filtered_means, refu['filtered_means']):.1e}")
    print(d, m, h, k, " | ".join(out), flush=True)
print("---- wave8 shapes, fp32 against the fp64 oracle")
for d, m, h in ((5, 2, (9, 7)), (5, 2, (64, 64)), (5, 2, (3, 50)), (2, 1, (20, 20)), (2, 1, (1, 1)), (6, 2, (40, 12)), (8, 4, (64, 64)), (8, 4, (17, 33))):
    mdl = mlp_model(rng, d, m, h)
    N, T = 2, 6
    t = o.irregular_times(rng, N, T, 0.012 * T)
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    out = []
    for order in ("first", "second"):
        ref = o.ekf_filter(mdl, t, y, state_order=order)
        ref32 = o.ekf_filter(mdl, t, y, state_order=order, dtype=np.float32) if "dtype" in o.ekf_filter.__code__.co_varnames else None
        p64 = cd.cdnlgssm_filter(P, y, t[..., None], cd.EKFHyperParams(state_order=order))
        p32 = cd.cdnlgssm_filter(P, y.astype(np.float32), t[..., None].astype(np.float32), cd.EKFHyperParams(state_order=order))
        k = L.cdkf_last_kernel().decode()[:30]
        out.append(f"{order}: f64 {relerr(p64.filtered_means, ref['filtered_means']):.1e} f32 fm {relerr(p32.filtered_means, ref['filtered_means']):.1e} fP {relerr(p32.filtered_covariances, ref['filtered_covariances']):.1e}"
                   + (f" oracle32 {relerr(ref32['filtered_means'], ref['filtered_means']):.1e}" if ref32 is not None else ""))
    print(d, m, h, k, " | ".join(out), flush=True)
