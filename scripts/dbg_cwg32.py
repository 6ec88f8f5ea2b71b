"""Dev aid: the fp32 instantiation of the run-time compiled workgroup kernels (python3 scripts/dbg_cwg32.py d m)."""
import os, sys, faulthandler
import numpy as np
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import cd_dynamax_amd as cd
import cdkf_oracle as o
from cd_dynamax_amd import _ffi
from helpers import relerr
from test_custom_drift import cubic_l96_src, wide_model, params_for

d, m = int(sys.argv[1]), int(sys.argv[2])
g = sys.argv[3] if len(sys.argv) > 3 else "auto"
rng = np.random.default_rng(1)
theta = np.array([4.0, 0.05])
mdl = wide_model(rng, d, m, theta, False)
N, T = 3, 6
t = o.irregular_times(rng, N, T, 0.03)
y = o.simulate(mdl, t, rng)
P = params_for(mdl, cd.LearnableCustomDrift(theta, cubic_l96_src(d), None, g))
ref = o.ekf_filter(mdl, t, y)
for order in ("first", "second"):
    print("fp32 filter", order, flush=True)
    post = cd.cdnlgssm_filter(P, y.astype(np.float32), t[..., None].astype(np.float32), cd.EKFHyperParams(state_order=order))
    print("err", relerr(post.filtered_covariances, ref["filtered_covariances"]), _ffi.lib().cdkf_last_kernel().decode(), flush=True)
