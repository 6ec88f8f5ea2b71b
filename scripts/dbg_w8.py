import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi
from helpers import mlp_model, params_from, relerr
L = _ffi.lib()
for (d, m, h) in [(8, 8, (32, 16)), (8, 4, (64, 64)), (4, 2, (7, 5))]:
    rng = np.random.default_rng(9)
    mdl = mlp_model(rng, d, m, h)
    N, T = 5, 9
    t = o.irregular_times(rng, N, T, 0.02)
    t[:, 5:] += 0.25
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    for order in ("first", "second"):
        ref = o.ekf_filter(mdl, t, y, state_order=order)
        hyp = cd.EKFHyperParams(state_order=order)
        post = cd.cdnlgssm_filter(P, y, t[..., None], hyp)
        print((d, m, h), order, L.cdkf_last_kernel().decode()[:34], "ll", relerr(post.marginal_loglik, ref["marginal_loglik"]),
              "fm", relerr(post.filtered_means, ref["filtered_means"]), "pP", relerr(post.predicted_covariances, ref["predicted_covariances"]),
              "pm", relerr(post.predicted_means, ref["predicted_means"]), flush=True)
        ll_ref, g_ref = o.ekf_loglik_grad_adjoint(mdl, t, y, state_order=order)
        ll, g = cd.cdnlgssm_loglik_and_grad(P, y, t[..., None], hyp)
        names = ["W1", "b1", "W2", "b2", "W3", "b3"]
        off = 0
        errs = {}
        scale = np.abs(g_ref).max()
        for nm, a_ in zip(names, g):
            a2 = np.asarray(a_).reshape(N, -1)
            errs[nm] = float(np.abs(a2 - g_ref[:, off:off + a2.shape[1]]).max() / scale)
            off += a2.shape[1]
        print("    grad", L.cdkf_last_kernel().decode()[:48], "ll", relerr(ll, ll_ref), {k: f"{v:.1e}" for k, v in errs.items()}, flush=True)
