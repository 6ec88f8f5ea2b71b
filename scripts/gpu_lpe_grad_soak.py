"""Random soak of the gradient on the lane grid against the other kernels (dev helper): the same cases in two processes, one with
CDKF_NO_LPE_GRAD=1 (forward sensitivities / wavefront-per-trajectory reverse sweep)."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import cdkf_oracle as o
import cd_dynamax_amd as cd
from helpers import params_from

def cases():
    rng = np.random.default_rng(2024)
    for c in range(40):
        m = int(rng.choice([3, 3, 3, 2, 1]))
        N, T = int(rng.integers(1, 45)), int(rng.integers(1, 40))
        span = float(10 ** rng.uniform(-2.5, 0.4)) * max(T, 2) / 10
        base = o.lorenz63_model(m)
        A, B, C = rng.standard_normal((3, 3)), rng.standard_normal((m, m)), rng.standard_normal((3, 3))
        Rm = B @ B.T / m + 0.5 * np.eye(m); Rm = 0.5 * (Rm + Rm.T)
        mdl = o.Model(base.drift, np.eye(3) + 0.2 * rng.standard_normal((3, 3)), A @ A.T / 3 + 0.3 * np.eye(3), np.eye(3)[:m], np.zeros(m),
                      Rm, rng.standard_normal(3) * 3, C @ C.T / 3 + 0.5 * np.eye(3))
        t = o.irregular_times(rng, N, T, span)
        y = o.simulate(mdl, t, rng)
        yield c, m, mdl, t, y

out = {}
for c, m, mdl, t, y in cases():
    P = params_from(mdl)
    ll, g = cd.cdnlgssm_loglik_and_grad(P, y, t[..., None])
    out[f"{c}_ll"], out[f"{c}_g"] = ll, np.stack([g.sigma, g.rho, g.beta], -1)
    ll, ga = cd.cdnlgssm_loglik_and_grad_all(P, y, t[..., None], cd.EKFHyperParams(state_order="first"))
    out[f"{c}_R"], out[f"{c}_H"], out[f"{c}_P0"] = ga.emissions.emission_cov.params, ga.emissions.emission_function.weights, ga.initial.cov.params
    out[f"{c}_Qc"] = ga.dynamics.diffusion_cov.params
if len(sys.argv) > 1:
    np.savez(sys.argv[1], **out)
    sys.exit(0)
f = "/tmp/lpe_grad_soak_other.npz"
subprocess.run([sys.executable, __file__, f], check=True, env={**os.environ, "CDKF_NO_LPE_GRAD": "1"})
other = np.load(f)
worst = {}
for k, v in out.items():
    b = other[k]
    e = np.abs(np.asarray(v) - b).max() / (np.abs(b).max() + 1e-300)
    kind = k.split("_")[1]
    worst[kind] = max(worst.get(kind, 0.0), e)
    if e > 1e-8: print("MISMATCH", k, e)
print("worst relative differences over 40 cases:", {k: float(f"{v:.2e}") for k, v in worst.items()})
