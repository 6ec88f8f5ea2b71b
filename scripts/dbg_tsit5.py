import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi
from helpers import mlp_model, params_from, relerr
rng = np.random.default_rng(41)
mdl = mlp_model(rng, 5, 2, (9, 7))
N, T = 4, 9
t = o.irregular_times(rng, N, T, 0.025)
t[:, 5:] += 0.06
y = o.simulate(mdl, t, rng)
P = params_from(mdl)
L = _ffi.lib()
for solver in ("dopri5", "tsit5", "bosh3"):
    for order in ("first", "second"):
        with o.use_solver(solver):
            ref = o.ekf_filter(mdl, t, y, state_order=order)
        hyp = cd.EKFHyperParams(state_order=order, diffeqsolve_settings={"solver": solver})
        post = cd.cdnlgssm_filter(P, y, t[..., None], hyp)
        print(solver, order, "filter kernel", L.cdkf_last_kernel().decode()[:40], "ll err", relerr(post.marginal_loglik, ref["marginal_loglik"]),
              "fm err", relerr(post.filtered_means, ref["filtered_means"]), flush=True)
        with o.use_solver(solver):
            ll_ref, g_ref = o.ekf_loglik_grad_adjoint(mdl, t, y, state_order=order)
        ll, g = cd.cdnlgssm_loglik_and_grad(P, y, t[..., None], hyp)
        flat = np.concatenate([np.asarray(a_).reshape(N, -1) for a_ in g], axis=-1)
        print("   grad kernel", L.cdkf_last_kernel().decode()[:50], "ll err", relerr(ll, ll_ref), "grad err", np.abs(flat - g_ref).max() / np.abs(g_ref).max(), flush=True)
