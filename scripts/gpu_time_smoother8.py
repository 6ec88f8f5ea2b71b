import ctypes as C, sys, os, time
ROOT='/root/repo'
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import cdkf_oracle as o
from cd_dynamax_amd import _ffi, models
from helpers import lorenz96_model, mlp_model, params_from, linear_model
L = _ffi.lib(); rng = np.random.default_rng(0)
def run(name, mdl, N, T):
    d, m = mdl.d, mdl.m
    t = o.irregular_times(rng, N, T, 0.005 * T); y = rng.standard_normal((N, T, m))
    blk = models._model_block(params_from(mdl)); opts = _ffi.default_opts(); opts.layout = _ffi.LAYOUT_TN; opts.state_order = 1
    A = lambda a: _ffi.DeviceArray.from_numpy(a)
    td, yd = A(np.ascontiguousarray(t.T)), A(np.ascontiguousarray(y.transpose(1, 0, 2)))
    ll, st = _ffi.DeviceArray((N,), np.float64), _ffi.DeviceArray((N,), np.int32)
    b = [_ffi.DeviceArray((T, N, w), np.float64) for w in (d, d * d, d, d * d)]
    for algo in ("ekf_filter", "ekf_smoother"):
        fn = getattr(L, f"cdkf_{algo}_f64_dev")
        for rep in range(2):
            t0 = time.perf_counter(); _ffi.check(fn(C.byref(blk.c), C.byref(opts), N, T, td.ptr, yd.ptr, ll.ptr, *[x.ptr for x in b], st.ptr, None)); _ffi.check(L.cdkf_synchronize(None)); el = time.perf_counter() - t0
        print(f"{name} {algo} N={N} T={T}: {el*1e3:.1f} ms", flush=True)
run("MLP d=8", mlp_model(np.random.default_rng(2), 8, 4, 64), 1024, 1000)
run("L96 d=8", lorenz96_model(8, 4), 1024, 1000)
run("lin d=6", linear_model(rng, 6, 3), 1024, 1000)
