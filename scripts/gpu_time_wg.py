"""Timing of the per-GPU slices of BASELINE configs 4 (Lorenz-96 d=40 EKF + smoother) and 5 (MLP drift, LL only)."""
import ctypes as C, sys, os, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "oracle"), os.path.join(_R, "tests")]
import numpy as np
import cdkf_oracle as o
from cd_dynamax_amd import _ffi, models
from helpers import lorenz96_model, mlp_model, params_from
L = _ffi.lib()
rng = np.random.default_rng(0)

def dev(arr=None, nbytes=None):
    p = C.c_void_p(); nb = arr.nbytes if arr is not None else nbytes
    _ffi.check(L.cdkf_malloc(C.byref(p), nb))
    if arr is not None: _ffi.check(L.cdkf_memcpy_h2d(p, arr.ctypes.data_as(C.c_void_p), nb))
    return p

def run(name, mdl, N, T, algo, dtype, outputs, T_total):
    d, m = mdl.d, mdl.m
    t = o.irregular_times(rng, N, T, T_total)
    y = rng.standard_normal((N, T, m)) + (8.0 if mdl.drift.kind == "lorenz96" else 0.0)
    blk = models._model_block(params_from(mdl))
    opts = _ffi.default_opts(); opts.layout = _ffi.LAYOUT_TN
    sz = np.dtype(dtype).itemsize; suf = "f64" if dtype == np.float64 else "f32"
    td, yd = dev(np.ascontiguousarray(t.T, dtype)), dev(np.ascontiguousarray(y.transpose(1, 0, 2), dtype))
    ll, st = dev(nbytes=N * sz), dev(nbytes=N * 4)
    bufs = [dev(nbytes=N * T * w * sz) if outputs else None for w in (d, d * d, d, d * d)]
    fn = getattr(L, f"cdkf_{algo}_{suf}_dev")
    for rep in range(3):
        t0 = time.perf_counter()
        _ffi.check(fn(C.byref(blk.c), C.byref(opts), N, T, td, yd, ll, *bufs, st, None))
        _ffi.check(L.cdkf_synchronize(None))
        el = time.perf_counter() - t0
    stat = np.empty(N, np.int32); _ffi.check(L.cdkf_memcpy_d2h(stat.ctypes.data_as(C.c_void_p), st, N * 4))
    print(f"{name}: {algo} {suf} N={N} T={T} d={d} m={m} outputs={outputs}: {el*1e3:.2f} ms -> {N/el:.3e} traj/s (status flags: {int((stat != 0).sum())})", flush=True)
    for p in [td, yd, ll, st] + [b for b in bufs if b is not None]: L.cdkf_free(p)

l96 = lorenz96_model(40, 40)
run("C4 slice", l96, 2048, 500, "ekf_filter", np.float64, True, 0.005 * 500)
run("C4 slice", l96, 2048, 500, "ekf_smoother", np.float64, True, 0.005 * 500)
run("C4 slice", l96, 2048, 500, "ekf_filter", np.float32, True, 0.005 * 500)
mlp = mlp_model(np.random.default_rng(2), 8, 4, 64)
run("C5 slice", mlp, 1024, 1000, "ekf_filter", np.float64, False, 0.005 * 1000)
run("C5 slice", mlp, 1024, 1000, "ekf_filter", np.float64, True, 0.005 * 1000)
run("C5 slice", mlp, 1024, 1000, "ekf_filter", np.float32, False, 0.005 * 1000)
run("C5 x8", mlp, 8192, 1000, "ekf_filter", np.float64, False, 0.005 * 1000)
