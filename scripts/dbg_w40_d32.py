import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi
from helpers import lorenz96_model, params_from, relerr
rng = np.random.default_rng(3)
for d, m in ((32, 32), (32, 16), (16, 16), (16, 7), (24, 24), (40, 40)):
    mdl = lorenz96_model(d, m)
    N, T = 3, 6
    t = o.irregular_times(rng, N, T, 0.015 * T)
    y = o.simulate(mdl, t, rng)
    ref = o.ekf_smoother(mdl, t, y)
    for dt in (np.float64,):
        post = cd.cdnlgssm_smoother(params_from(mdl), y, t[..., None])
        flt = cd.cdnlgssm_filter(params_from(mdl), y, t[..., None])
        refl = o.ekf_filter(mdl, t, y)
        print(d, m, _ffi.lib().cdkf_last_kernel().decode()[:36], "fm", relerr(post.filtered_means, ref["filtered_means"]), "fP", relerr(post.filtered_covariances, ref["filtered_covariances"]),
              "sm", relerr(post.smoothed_means, ref["smoothed_means"]), "sP", relerr(post.smoothed_covariances, ref["smoothed_covariances"]),
              "pm", relerr(flt.predicted_means, refl["predicted_means"]), "pP", relerr(flt.predicted_covariances, refl["predicted_covariances"]), "ll", relerr(post.marginal_loglik, ref["marginal_loglik"]), flush=True)
        # first step only
        print("   step0 fm", relerr(post.filtered_means[:, 0], ref["filtered_means"][:, 0]), "fP", relerr(post.filtered_covariances[:, 0], ref["filtered_covariances"][:, 0]),
              "pm0", relerr(flt.predicted_means[:, 0], refl["predicted_means"][:, 0]), "pP0", relerr(flt.predicted_covariances[:, 0], refl["predicted_covariances"][:, 0]))
