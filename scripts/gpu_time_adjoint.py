"""Time the C5-slice value-and-gradient (MLP drift d=8, m=4, 2x64; dev helper)."""
import ctypes as C, sys, time
import numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import cdkf_oracle as o
from cd_dynamax_amd import _ffi
from cd_dynamax_amd.models import _model_block, _opts
import cd_dynamax_amd as cd
from helpers import params_from, mlp_model

L = _ffi.lib()
N, T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 1000
for dtype, sfx in ((np.float64, "f64"), (np.float32, "f32")):
    rng = np.random.default_rng(0)
    mdl = mlp_model(rng, 8, 4, 64)
    t = o.irregular_times(rng, N, T, 0.01).astype(dtype)
    y = (rng.standard_normal((N, T, 4))).astype(dtype)
    mb = _model_block(params_from(mdl)); opts = _opts(cd.EKFHyperParams(state_order="first"), 1); opts.layout = _ffi.LAYOUT_TN
    tt = np.ascontiguousarray(t.T); yy = np.ascontiguousarray(y.transpose(1, 0, 2))
    dt_, dy_ = _ffi.DeviceArray.from_numpy(tt), _ffi.DeviceArray.from_numpy(yy)
    dll, dg, dst = _ffi.DeviceArray((N,), dtype), _ffi.DeviceArray((N, mb.theta.size), dtype), _ffi.DeviceArray((N,), np.int32)
    fn = getattr(L, f"cdkf_ekf_loglik_grad_{sfx}_dev")
    flt = getattr(L, f"cdkf_ekf_filter_{sfx}_dev")
    for rep in range(3):
        t0 = time.perf_counter()
        _ffi.check(flt(C.byref(mb.c), C.byref(opts), N, T, dt_.ptr, dy_.ptr, dll.ptr, None, None, None, None, dst.ptr, None)); _ffi.check(L.cdkf_synchronize(None))
        el_f = time.perf_counter() - t0
        t0 = time.perf_counter()
        _ffi.check(fn(C.byref(mb.c), C.byref(opts), N, T, dt_.ptr, dy_.ptr, dll.ptr, dg.ptr, dst.ptr, None)); _ffi.check(L.cdkf_synchronize(None))
        el = time.perf_counter() - t0
    g = dg.numpy()
    print(f"{sfx} N={N} T={T}: filter(LL only) {el_f*1e3:.1f} ms, value+grad {el*1e3:.1f} ms, |g| finite={np.isfinite(g).all()} max={np.abs(g).max():.3g}")
