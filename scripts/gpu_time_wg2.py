import ctypes as C, sys, os, time
sys.path[:0] = [os.path.join(os.path.dirname(__file__), ".."), os.path.join(os.path.dirname(__file__), "..", "oracle"), os.path.join(os.path.dirname(__file__), "..", "tests")]
import numpy as np
import cdkf_oracle as o
from cd_dynamax_amd import _ffi, models
from helpers import lorenz96_model, mlp_model, params_from, linear_model
L = _ffi.lib()
rng = np.random.default_rng(0)
def dev(arr=None, nbytes=None):
    p = C.c_void_p(); nb = arr.nbytes if arr is not None else nbytes
    _ffi.check(L.cdkf_malloc(C.byref(p), nb))
    if arr is not None: _ffi.check(L.cdkf_memcpy_h2d(p, arr.ctypes.data_as(C.c_void_p), nb))
    return p
def run(name, mdl, N, T, order=2, outputs=False, gap=0.005):
    d, m = mdl.d, mdl.m
    t = o.irregular_times(rng, N, T, gap * T)
    y = rng.standard_normal((N, T, m)) + (8.0 if mdl.drift.kind == "lorenz96" else 0.0)
    blk = models._model_block(params_from(mdl))
    opts = _ffi.default_opts(); opts.layout = _ffi.LAYOUT_TN; opts.state_order = order
    td, yd = dev(np.ascontiguousarray(t.T)), dev(np.ascontiguousarray(y.transpose(1, 0, 2)))
    ll, st = dev(nbytes=N * 8), dev(nbytes=N * 4)
    bufs = [dev(nbytes=N * T * w * 8) if outputs else None for w in (d, d * d, d, d * d)]
    for rep in range(2):
        t0 = time.perf_counter()
        _ffi.check(L.cdkf_ekf_filter_f64_dev(C.byref(blk.c), C.byref(opts), N, T, td, yd, ll, *bufs, st, None))
        _ffi.check(L.cdkf_synchronize(None))
        el = time.perf_counter() - t0
    print(f"{name}: d={d} m={m} order={order} N={N} T={T} out={outputs}: {el*1e3:.1f} ms = {el/T*1e6/ max(1,(N+255)//256):.1f} us per step-round", flush=True)
    for p in [td, yd, ll, st] + [b for b in bufs if b is not None]: L.cdkf_free(p)
mlp = mlp_model(np.random.default_rng(2), 8, 4, 64)
for order in (2, 1, 0):
    run("MLP", mlp, 256, 200, order)
run("MLP gap x4", mlp, 256, 200, 1, gap=0.02)
for m in (40, 8, 1):
    run("L96", lorenz96_model(40, m), 256, 100, 1)
run("L96 out", lorenz96_model(40, 40), 256, 100, 1, outputs=True)
run("L96 zeroth", lorenz96_model(40, 40), 256, 100, 0)
run("L96 gapx4", lorenz96_model(40, 1), 256, 100, 1, gap=0.02)
run("lin d=16", linear_model(rng, 16, 4), 256, 200, 1)
run("lin d=5", linear_model(rng, 5, 3), 256, 200, 1)
