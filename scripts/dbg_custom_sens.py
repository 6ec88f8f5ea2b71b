"""Drift-only gradient of source drifts on the register-resident forward-sensitivity kernel against the all-leaf reverse sweep, over
state / emission dimensions 1 .. 6: python3 scripts/dbg_custom_sens.py [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi
from helpers import random_quadratic_drift
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(seed)
def spd(n, s):
    A = rng.standard_normal((n, n)); return A @ A.T / n * s + 0.3 * np.eye(n)
for d in range(1, 7):
    for m in range(1, 7):
        for sel in (False, True):
            if sel and m > d: continue
            src, make = random_quadratic_drift(rng, d)
            theta = np.array([0.5 + 0.5 * rng.random(), 0.2 * rng.standard_normal()])
            H, bias = (np.eye(d)[rng.permutation(d)[:m]], np.zeros(m)) if sel else (rng.standard_normal((m, d)) / np.sqrt(d), 0.1 * rng.standard_normal(m))
            mdl = o.Model(make(theta), np.eye(d) + 0.1 * rng.standard_normal((d, d)), spd(d, 0.3), H, bias, spd(m, 0.5), 0.5 * rng.standard_normal(d), spd(d, 0.3))
            N, T = 3, 6
            t = o.irregular_times(rng, N, T, 0.03 * T)
            y = o.simulate(mdl, t, rng)
            P = cd.ParamsCDNLGSSM(
                initial=cd.ParamsLGSSMInitial(cd.LearnableVector(mdl.m0), cd.LearnableMatrix(mdl.P0)),
                dynamics=cd.ParamsCDNLGSSMDynamics(cd.LearnableCustomDrift(theta, src, None, None), cd.LearnableMatrix(mdl.L), cd.LearnableMatrix(mdl.Qc), 2.0),
                emissions=cd.ParamsCDNLGSSMEmissions(cd.LearnableLinear(mdl.H, mdl.bias), cd.LearnableMatrix(mdl.R)))
            hyp = cd.EKFHyperParams(state_order="first")
            ll_ref, g_ref = o.ekf_loglik_grad_adjoint(mdl, t, y, state_order="first")
            ll, g1 = cd.cdnlgssm_loglik_and_grad(P, y, t[..., None], hyp)
            k1 = _ffi.lib().cdkf_last_kernel().decode()[:50]
            e = np.abs(np.asarray(g1.theta) - g_ref).max() / (np.abs(g_ref).max() + 1e-300)
            if e > 1e-8: print("MISMATCH d", d, "m", m, "sel", sel, "err %.2e" % e, k1, flush=True)
print("done", seed)
