"""d = 6, m = 1 source drift with pow(): drift-only gradient (forward sensitivities) per spelling, several draws: python3 scripts/dbg_custom_pow.py"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi
from helpers import random_quadratic_drift
d, m = int(sys.argv[1]) if len(sys.argv) > 1 else 6, int(sys.argv[2]) if len(sys.argv) > 2 else 1
found = 0
for seed in range(400):
    rng = np.random.default_rng(5000 + seed)
    src, make = random_quadratic_drift(rng, d)
    if "pow(" not in src: continue
    found += 1
    theta = np.array([0.7, -0.15])
    A = rng.standard_normal((d, d)); B = rng.standard_normal((m, m))
    mdl = o.Model(make(theta), np.eye(d) + 0.1 * rng.standard_normal((d, d)), A @ A.T / d * 0.3 + 0.3 * np.eye(d),
                  rng.standard_normal((m, d)) / np.sqrt(d), 0.1 * rng.standard_normal(m), B @ B.T / m * 0.5 + 0.3 * np.eye(m),
                  0.5 * rng.standard_normal(d), 0.3 * np.eye(d))
    N, T = 3, 6
    t = o.irregular_times(rng, N, T, 0.03 * T); y = o.simulate(mdl, t, rng)
    hyp = cd.EKFHyperParams(state_order="first")
    ll_ref, g_ref = o.ekf_loglik_grad_adjoint(mdl, t, y, state_order="first")
    out = []
    for name, s_ in (("pow2", src), ("pow2.0", re.sub(r"pow\((x\[\d+\]), 2\)", r"pow(\1, 2.0)", src)), ("x*x", re.sub(r"pow\((x\[\d+\]), 2\)", r"(\1 * \1)", src))):
        P = cd.ParamsCDNLGSSM(
            initial=cd.ParamsLGSSMInitial(cd.LearnableVector(mdl.m0), cd.LearnableMatrix(mdl.P0)),
            dynamics=cd.ParamsCDNLGSSMDynamics(cd.LearnableCustomDrift(theta, s_, None, None), cd.LearnableMatrix(mdl.L), cd.LearnableMatrix(mdl.Qc), 2.0),
            emissions=cd.ParamsCDNLGSSMEmissions(cd.LearnableLinear(mdl.H, mdl.bias), cd.LearnableMatrix(mdl.R)))
        ll, g1 = cd.cdnlgssm_loglik_and_grad(P, y, t[..., None], hyp)
        out.append("%s %.1e" % (name, np.abs(np.asarray(g1.theta) - g_ref).max() / np.abs(g_ref).max()))
    print(seed, " | ".join(out), "|", src.count("pow("), "pow terms", flush=True)
    if found >= int(os.environ.get("DBG_MAX", "6")): break
