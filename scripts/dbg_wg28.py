import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
os.environ["CDKF_NO_WAVE40"] = "1"
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi
from helpers import lorenz96_model, linear_model, params_from, relerr
L = _ffi.lib()
rng = np.random.default_rng(5)
for d, m in ((24, 24), (28, 28), (28, 14), (31, 5), (23, 23), (48, 48), (46, 10)):
    mdl = lorenz96_model(d, m)
    N, T = 3, 6
    t = o.irregular_times(rng, N, T, 0.012 * T)
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    ref = o.ekf_smoother(mdl, t, y)
    import time
    t0 = time.time()
    post = cd.cdnlgssm_smoother(P, y, t[..., None])
    el = time.time() - t0
    print(d, m, L.cdkf_last_kernel().decode(), f"{el:.2f}s", "fm", relerr(post.filtered_means, ref["filtered_means"]), "fP", relerr(post.filtered_covariances, ref["filtered_covariances"]),
          "sm", relerr(post.smoothed_means, ref["smoothed_means"]), "sP", relerr(post.smoothed_covariances, ref["smoothed_covariances"]), flush=True)
