"""Dev aid: per-leaf errors of the custom-drift gradient on the cubic Lorenz-96 model at a given (d, m): python3 scripts/dbg_cgrad.py d m"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import cd_dynamax_amd as cd
import cdkf_oracle as o
from cd_dynamax_amd import _ffi
from test_custom_drift import cubic_l96_src, wide_model, params_for

d, m = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(5)
theta = np.array([4.0, 0.05])
mdl = wide_model(rng, d, m, theta, False)
N, T = 2, 5
t = o.irregular_times(rng, N, T, float(sys.argv[3]) if len(sys.argv) > 3 else 0.04)
y = o.simulate(mdl, t, rng)
P = params_for(mdl, cd.LearnableCustomDrift(theta, cubic_l96_src(d), None, None))
hyp = cd.EKFHyperParams(state_order="first")
ll_ref, g_ref, full = o.ekf_loglik_grad_adjoint(mdl, t, y, full=True, state_order="first")
ll, g = cd.cdnlgssm_loglik_and_grad_all(P, y, t[..., None], hyp)
print(d, m, _ffi.lib().cdkf_last_kernel().decode())
print(" ll", np.abs(ll - ll_ref).max())
print(" theta got", np.asarray(g.dynamics.drift.theta)[0], "want", g_ref[0])
for name, got, want in (("m0", g.initial.mean.params, full["m0"]), ("P0", g.initial.cov.params, full["P0"]), ("H", g.emissions.emission_function.weights, full["H"]),
                        ("R", g.emissions.emission_cov.params, full["R"]), ("Qc", g.dynamics.diffusion_cov.params, full["Qc"])):
    print(" ", name, np.abs(np.asarray(got) - want).max() / np.abs(want).max())
