"""Which run-time compiled kernel of the d = 2 test drift is wrong under which build policy: log-likelihood of the filter variant and
gradient of the forward-sensitivity variant against the oracle, per trajectory.  python scripts/r5_d2_check.py  (children per policy)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORK = r'''
import sys, os
sys.path[:0] = [os.environ["ROOT"], os.path.join(os.environ["ROOT"], "oracle"), os.path.join(os.environ["ROOT"], "tests")]
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from test_custom_drift import NL_F, make_model, params_for
rng = np.random.default_rng(90)
ref_mdl = o.lorenz63_model(2); N, T = 7, 25
t = o.irregular_times(rng, N, T, 0.3); y = o.simulate(ref_mdl, t, rng)   # (the draws the test makes before its second part)
theta = np.array([1.7, 0.25, 0.4])
def f_np(x, thv):
    s, e = np.sin(x[..., 0]), np.exp(-thv[2] * x[..., 1] ** 2)
    return np.stack([x[..., 1] + thv[1] * np.tanh(x[..., 0] * x[..., 1]),
                     -thv[0] * s * e - thv[1] * x[..., 1] + 0.3 * np.cos(2 * x[..., 0]) / (1 + x[..., 0] ** 2) + np.sqrt(1 + x[..., 1] ** 2) * thv[2] ** 2], -1)
def jac_np(x, thv, h=1e-6):
    return np.stack([(f_np(x + h * np.eye(2)[j], thv) - f_np(x - h * np.eye(2)[j], thv)) / (2 * h) for j in range(2)], -1)
mdl = make_model(o.CallableDrift(theta, f_np, jac_np, None), 1)
N, T = 6, 30
t = o.irregular_times(rng, N, T, 0.4); y = o.simulate(mdl, t, rng)
hyp = cd.EKFHyperParams(state_order="first")
Pg = params_for(mdl, cd.LearnableCustomDrift(theta, NL_F, None, None))
ll_ref = o.ekf_filter(mdl, t, y, state_order="first")["marginal_loglik"]
ll_f = cd.cdnlgssm_filter(Pg, y, t[..., None], hyp).marginal_loglik
ll_g, g = cd.cdnlgssm_loglik_and_grad(Pg, y, t[..., None], hyp)
def ll_of(th):
    return o.ekf_filter(make_model(o.CallableDrift(th, f_np, jac_np, None), 1), t, y, state_order="first")["marginal_loglik"]
fd = np.stack([(ll_of(theta + 1e-5 * np.eye(3)[p]) - ll_of(theta - 1e-5 * np.eye(3)[p])) / 2e-5 for p in range(3)], -1)
g = np.asarray(g.theta)
print("RESULT filter ll err per trajectory", np.array2string(np.abs(ll_f - ll_ref) / np.abs(ll_ref), precision=1), "| grad-kernel ll err",
      np.array2string(np.abs(ll_g - ll_ref) / np.abs(ll_ref), precision=1), "| grad err per trajectory", np.array2string(np.abs(g - fd).max(1) / np.abs(fd).max(), precision=1))
'''
for pol in (sys.argv[1:] or ["", "o3", "o1"]):
    env = dict(os.environ, ROOT=ROOT, CDKF_RTC_CACHE="0")
    if pol: env["CDKF_RTC_POLICY"] = pol
    p = subprocess.run([sys.executable, "-c", WORK], env=env, capture_output=True, text=True, timeout=3000)
    print("[policy %s]" % (pol or "default"), ([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")] or [p.stderr[-600:]])[-1], flush=True)
