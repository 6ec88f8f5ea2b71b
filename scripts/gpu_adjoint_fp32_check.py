"""Dev aid: float32 against float64 on the shape-generic reverse sweep (every leaf), random Lorenz-96 / linear problems:
python3 scripts/gpu_adjoint_fp32_check.py [seed] [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from helpers import linear_model, params_from

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rng = np.random.default_rng(seed)
worst = 0.0
for case in range(cases):
    lin = rng.random() < 0.3
    d = int(rng.integers(9, 18)) if lin else int(rng.integers(9, 44))
    m = int(rng.integers(1, d + 1))
    drift = linear_model(rng, d, m).drift if lin else o.Lorenz96Drift(8.0)
    if rng.random() < 0.5:
        H, bias = np.eye(d)[rng.permutation(d)[:m]], np.zeros(m)
    else:
        H, bias = rng.standard_normal((m, d)) / np.sqrt(d), 0.1 * rng.standard_normal(m)
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    Rm = rng.standard_normal((m, m)) / np.sqrt(m)
    rscale = float(rng.choice([1.0, 0.1, 0.01]))     # small R: a worse-conditioned S
    mdl = o.Model(drift, np.eye(d), 0.3 * np.eye(d) + 0.1 * A @ A.T, H, bias, rscale * (0.5 * np.eye(m) + 0.1 * Rm @ Rm.T),
                  (0.0 if lin else 8.0) + rng.standard_normal(d), 0.5 * np.eye(d) + 0.2 * A.T @ A)
    N, T = 2, int(rng.integers(2, 7))
    t = o.irregular_times(rng, N, T, 0.015 * T)
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    hyp = cd.EKFHyperParams(state_order="first")
    ll64, g64 = cd.cdnlgssm_loglik_and_grad_all(P, y, t[..., None], hyp)
    ll32, g32 = cd.cdnlgssm_loglik_and_grad_all(P, y.astype(np.float32), t[..., None].astype(np.float32), hyp)
    leaves = lambda g: [np.concatenate([np.asarray(a_).reshape(N, -1) for a_ in g.dynamics.drift], -1), g.initial.mean.params, g.initial.cov.params,
                        g.dynamics.diffusion_cov.params, g.emissions.emission_function.weights, g.emissions.emission_cov.params]
    e = max(np.abs(np.asarray(a_, np.float64) - np.asarray(b_)).max() / (np.abs(np.asarray(b_)).max() + 1e-300) for a_, b_ in zip(leaves(g32), leaves(g64)))
    worst = max(worst, e)
    print(case, d, m, "lin" if lin else "l96", "R x", rscale, "fp32 vs fp64: %.2e" % e, flush=True)
print("worst", worst)
