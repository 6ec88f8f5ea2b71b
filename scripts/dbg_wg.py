import sys, os
sys.path[:0] = [os.path.join(os.path.dirname(__file__), ".."), os.path.join(os.path.dirname(__file__), "..", "oracle"), os.path.join(os.path.dirname(__file__), "..", "tests")]
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi, models
from helpers import lorenz96_model, params_from, relerr
rng = np.random.default_rng(12)
for d, m in ((12, 6), (12, 5), (12, 4), (10, 5), (12,12)):
    mdl = lorenz96_model(d, m)
    N, T = 5, 40
    t = o.irregular_times(rng, N, T, 0.012 * T)
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    ref = o.ekf_filter(mdl, t, y)
    for rep in range(3):
      for want in ([True] * 4, [False, False, True, True]):
        blk = models._model_block(P)
        opts = _ffi.default_opts()
        ll, outs, st = _ffi.run_host("ekf_filter", blk, opts, t, y, want, np.float64)
        k = [i for i,w in enumerate(want) if w][0]
        key = ["filtered_means","filtered_covariances","predicted_means","predicted_covariances"][k]
        print(d, m, want, "ll err %.2e" % relerr(ll, ref["marginal_loglik"]), "status", st, "err %.2e" % relerr(outs[k], ref[key]))
