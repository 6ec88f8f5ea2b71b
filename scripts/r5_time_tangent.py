"""Round 5: what the unscented filter's tangent sweep (cdkf_ukf_tangent_kernels.h) costs -- MLP drift d = 8 (hidden 32 / 32: 1 608 drift
parameters, 1 796 leaf entries) and a source drift d = 6, N trajectories x T = 100, fp64 and fp32; the unscented FILTER of the same
batch beside it.  gpurun -- 'python scripts/r5_time_tangent.py'"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi
from helpers import mlp_model, params_from, random_quadratic_drift


def timed(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


rng = np.random.default_rng(9)
T = 100
for name, N, mk in (("MLP d=8 m=4 hidden 32/32", 64, lambda: params_from(mlp_model(rng, 8, 4, 32))),
                    ("MLP d=8 m=4 hidden 64/64 (config 5's network)", 16, lambda: params_from(mlp_model(rng, 8, 4, 64)))):
    P = mk()
    t = o.irregular_times(rng, N, T, 0.005 * T)
    y = rng.standard_normal((N, T, 4))
    for dt in (np.float64, np.float32):
        yy, tt = y.astype(dt), t[..., None].astype(dt)
        f = timed(lambda: cd.cdnlgssm_filter(P, yy, tt, cd.UKFHyperParams(), output_fields=[]))
        a = timed(lambda: cd.cdnlgssm_loglik_and_grad_all(P, yy, tt, cd.UKFHyperParams()))
        k = _ffi.lib().cdkf_last_kernel().decode()
        e = timed(lambda: cd.cdnlgssm_loglik_and_grad_all(P, yy, tt, cd.EKFHyperParams(state_order="first")))
        print("RESULT %s N=%d T=%d %s: ukf filter %.1f ms | ukf value + every gradient %.1f ms (%s) | ekf value + every gradient (reverse sweep) %.1f ms"
              % (name, N, T, np.dtype(dt).name, f, a, k, e), flush=True)
d = 6
src, make = random_quadratic_drift(np.random.default_rng(101), d)
theta = np.array([0.7, -0.15])
mdl = o.Model(make(theta), np.eye(d), 0.3 * np.eye(d), rng.standard_normal((3, d)) / np.sqrt(d), np.zeros(3), 0.5 * np.eye(3), 0.5 * rng.standard_normal(d), 0.3 * np.eye(d))
P = cd.ParamsCDNLGSSM(initial=cd.ParamsLGSSMInitial(cd.LearnableVector(mdl.m0), cd.LearnableMatrix(mdl.P0)),
                      dynamics=cd.ParamsCDNLGSSMDynamics(cd.LearnableCustomDrift(theta, src, None, None), cd.LearnableMatrix(mdl.L), cd.LearnableMatrix(mdl.Qc), 2.0),
                      emissions=cd.ParamsCDNLGSSMEmissions(cd.LearnableLinear(mdl.H, mdl.bias), cd.LearnableMatrix(mdl.R)))
N = 512
t = o.irregular_times(rng, N, T, 0.005 * T)
y = rng.standard_normal((N, T, 3))
f = timed(lambda: cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams(), output_fields=[]))
g = timed(lambda: cd.cdnlgssm_loglik_and_grad(P, y, t[..., None], cd.UKFHyperParams()))
a = timed(lambda: cd.cdnlgssm_loglik_and_grad_all(P, y, t[..., None], cd.UKFHyperParams()))
print("RESULT source drift d=6 m=3 N=%d T=%d float64: ukf filter %.1f ms | value + d/dtheta (2 lanes per trajectory) %.1f ms | value + every gradient (77 leaf entries) %.1f ms"
      % (N, T, f, g, a), flush=True)
