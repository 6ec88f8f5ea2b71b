import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi
from helpers import mlp_model, params_from, relerr
L = _ffi.lib()
rng = np.random.default_rng(7)
for d, m, h in ((5, 2, (9, 7)), (5, 2, (64, 64)), (8, 4, (64, 64))):
    mdl = mlp_model(rng, d, m, h)
    N, T = 2, 6
    t = o.irregular_times(rng, N, T, 0.012 * T)
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    out = []
    for order in ("first", "second"):
        ref = o.ekf_filter(mdl, t, y, state_order=order)
        p32 = cd.cdnlgssm_filter(P, y.astype(np.float32), t[..., None].astype(np.float32), cd.EKFHyperParams(state_order=order))
        out.append(f"{order}: f32 fm {relerr(p32.filtered_means, ref['filtered_means']):.1e} pm {relerr(p32.predicted_means, ref['predicted_means']):.1e}")
    print(os.environ.get("CDKF_LIB_PATH", "HEAD")[-22:], d, m, h, " | ".join(out), flush=True)
