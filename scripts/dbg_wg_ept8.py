import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
os.environ["CDKF_NO_WAVE40"] = "1"
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi
from helpers import lorenz96_model, params_from, relerr
L = _ffi.lib()
rng = np.random.default_rng(5)
for d, m in ((44, 44), (45, 9), (46, 46), (48, 48), (48, 6), (56, 8), (64, 4)):
    mdl = lorenz96_model(d, m)
    N, T = 2, 4
    t = o.irregular_times(rng, N, T, 0.012 * T)
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    ref = o.ekf_filter(mdl, t, y)
    for dt in (np.float64, np.float32):
        t0 = time.time()
        try:
            post = cd.cdnlgssm_filter(P, y.astype(dt), t[..., None].astype(dt))
        except Exception as e:
            print(d, m, dt.__name__, "ERR", str(e)[:100]); continue
        el = time.time() - t0
        fm = np.asarray(post.filtered_means, np.float64)
        bad = np.argwhere(~np.isfinite(fm))
        print(d, m, dt.__name__, L.cdkf_last_kernel().decode(), f"{el:.2f}s", "fm", relerr(fm, ref["filtered_means"]), "ll", relerr(post.marginal_loglik, ref["marginal_loglik"]),
              "first nan", bad[0].tolist() if len(bad) else None, flush=True)
