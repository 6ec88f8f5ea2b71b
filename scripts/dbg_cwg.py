"""Dev aid: a drift given as source on the workgroup kernels at growing state dimension (python3 scripts/dbg_cwg.py d m)."""
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
rng = np.random.default_rng(1)
theta = np.array([8.0, 0.0])
mdl = wide_model(rng, d, m, theta, True)
N, T = 3, 6
t = o.irregular_times(rng, N, T, 0.03)
y = o.simulate(mdl, t, rng)
P = params_for(mdl, cd.LearnableCustomDrift(theta, cubic_l96_src(d), None, ""))
print("filter", flush=True)
post = cd.cdnlgssm_filter(P, y, t[..., None])
ref = o.ekf_filter(mdl, t, y)
print("filter err", relerr(post.filtered_covariances, ref["filtered_covariances"]), _ffi.lib().cdkf_last_kernel().decode(), flush=True)
print("smoother", flush=True)
sm = cd.cdnlgssm_smoother(P, y, t[..., None])
ref = o.ekf_smoother(mdl, t, y)
print("smoother err", relerr(sm.smoothed_covariances, ref["smoothed_covariances"]), flush=True)
