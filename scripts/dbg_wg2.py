import sys, os
sys.path[:0] = [os.path.join(os.path.dirname(__file__), ".."), os.path.join(os.path.dirname(__file__), "..", "oracle"), os.path.join(os.path.dirname(__file__), "..", "tests")]
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi, models
from helpers import lorenz96_model, params_from, relerr
rng = np.random.default_rng(12)
for d, m in ((20, 20), (28, 28), (31,31), (32, 32), (33,33)):
    mdl = lorenz96_model(d, m)
    N, T = 2, 6
    t = o.irregular_times(rng, N, T, 0.012 * T)
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    ref = o.ekf_filter(mdl, t, y)
    for dt in (np.float64, np.float32):
        blk = models._model_block(P)
        opts = _ffi.default_opts()
        ll, outs, st = _ffi.run_host("ekf_filter", blk, opts, t, y, [True]*4, dt)
        print(d, m, dt.__name__, "ll err %.2e" % relerr(ll, ref["marginal_loglik"]), "status", st, "err %.2e" % relerr(outs[1], ref["filtered_covariances"]), flush=True)
