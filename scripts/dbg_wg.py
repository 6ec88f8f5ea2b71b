import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi
from helpers import mlp_model, lorenz96_model, linear_model, params_from, relerr
L = _ffi.lib()
rng = np.random.default_rng(41)
cases = [("mlp5", mlp_model(rng, 5, 2, (9, 7))), ("l96_6", lorenz96_model(6, 3)), ("l96_12", lorenz96_model(12, 12)), ("lin5", linear_model(rng, 5, 3))]
for name, mdl in cases:
    N, T = 4, 9
    t = o.irregular_times(rng, N, T, 0.025)
    t[:, 5:] += 0.06
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    for solver in ("dopri5", "tsit5", "euler"):
        for order in ("first", "second", "zeroth"):
            with o.use_solver(solver):
                ref = o.ekf_filter(mdl, t, y, state_order=order)
            hyp = cd.EKFHyperParams(state_order=order, diffeqsolve_settings={"solver": solver})
            post = cd.cdnlgssm_filter(P, y, t[..., None], hyp)
            print(name, solver, order, L.cdkf_last_kernel().decode()[:44], "ll", f"{relerr(post.marginal_loglik, ref['marginal_loglik']):.1e}",
                  "fm", f"{relerr(post.filtered_means, ref['filtered_means']):.1e}", "pP", f"{relerr(post.predicted_covariances, ref['predicted_covariances']):.1e}", flush=True)
