"""Edge sizes through the Python surface (dev helper): T = 1, N = 0, single trajectory, every entry point."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from helpers import linear_model, lorenz96_model, mlp_model, params_from, relerr
rng = np.random.default_rng(0)
for name, mdl in (("l63", o.lorenz63_model(3)), ("lin4", linear_model(rng, 4, 2)), ("l96", lorenz96_model(6, 3)), ("mlp", mlp_model(rng, 4, 2, (5, 6))),
                  ("l96-40", lorenz96_model(40, 7))):
    P = params_from(mdl)
    for N, T in ((1, 1), (3, 1), (1, 2), (0, 3)):
        t = o.irregular_times(rng, max(N, 1), T, 0.05)[:N]
        y = o.simulate(mdl, t if N else o.irregular_times(rng, 1, T, 0.05), rng)[:N]
        hyp = cd.EKFHyperParams(state_order="first")
        f = cd.cdnlgssm_filter(P, y, t[..., None], hyp)
        s = cd.cdnlgssm_smoother(P, y, t[..., None], hyp)
        u = cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams())
        msg = f"{name} N={N} T={T}: filter {f.filtered_means.shape} smoother {s.smoothed_means.shape} ukf {u.filtered_means.shape}"
        if N:
            ref = o.ekf_smoother(mdl, t, y, state_order="first")
            msg += f" err {relerr(s.smoothed_covariances, ref['smoothed_covariances']):.1e}"
            if mdl.d <= 8:
                ll, g = cd.cdnlgssm_loglik_and_grad_all(P, y, t[..., None], hyp)
                _, gr, ex = o.ekf_loglik_grad_adjoint(mdl, t, y, full=True)
                msg += f" gradR {relerr(g.emissions.emission_cov.params, ex['R']):.1e}"
        print(msg, flush=True)
lin = linear_model(rng, 3, 2)
lin = o.Model(o.LinearDrift(lin.drift.W, np.zeros(3)), lin.L, lin.Qc, lin.H, lin.bias, lin.R, lin.m0, lin.P0)
model = cd.ContDiscreteLinearGaussianSSM(3, 2, has_emissions_bias=True)
pp = cd.ParameterProperties()
params, _ = model.initialize(initial_mean={"params": lin.m0, "props": pp}, initial_cov={"params": lin.P0, "props": pp},
                             dynamics_weights={"params": lin.drift.W, "props": pp}, dynamics_diffusion_coefficient={"params": lin.L, "props": pp},
                             dynamics_diffusion_cov={"params": lin.Qc, "props": pp}, emission_weights={"params": lin.H, "props": pp},
                             emission_bias={"params": lin.bias, "props": pp}, emission_cov={"params": lin.R, "props": pp})
for T in (1, 2):
    y = rng.standard_normal((T, 2))
    sm = model.smoother(params, y)
    print("type-1 smoother T =", T, sm.smoothed_means.shape, sm.smoothed_cross_covariances.shape)
print("edge cases ok")
