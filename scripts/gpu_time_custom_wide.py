"""Time a drift given as source against the built-in one on the same model (Lorenz-96, d = 40 by default): filter sweep on the
workgroup kernels (built-in: CDKF_NO_WAVE40=1 keeps it there too) and value + gradient on the shape-generic reverse sweep:
python3 scripts/gpu_time_custom_wide.py [d=40] [n=256] [t=100]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi
from cd_dynamax_amd.models import _model_block, _opts
from helpers import lorenz96_model, params_from

L = _ffi.lib()
arg = lambda k, v: next((int(a[len(k) + 1:]) for a in sys.argv[1:] if a.startswith(k + "=")), v)
d, N, T = arg("d", 40), arg("n", 256), arg("t", 100)
rng = np.random.default_rng(1)
mdl = lorenz96_model(d, d)
u = rng.uniform(0, 1, (N, T)); t = np.cumsum(u, 1); t = t / t[:, -1:] * (0.005 * T)
y = 8.0 + rng.standard_normal((N, T, d))
src = (f"for (int i = 0; i < {d}; ++i) {{ const int ip1 = (i + 1) % {d}, im1 = (i + {d - 1}) % {d}, im2 = (i + {d - 2}) % {d}; "
       f"fx[i] = (x[ip1] - x[im2]) * x[im1] - x[i] + theta[0]; }}")
Pb = params_from(mdl)
Pc = Pb._replace(dynamics=Pb.dynamics._replace(drift=cd.LearnableCustomDrift(np.array([8.0]), src, None, "")))
def dev(a):
    p = C.c_void_p(); _ffi.check(L.cdkf_malloc(C.byref(p), a.nbytes)); _ffi.check(L.cdkf_memcpy_h2d(p, a.ctypes.data_as(C.c_void_p), a.nbytes)); return p
for dtype, sfx in ((np.float64, "f64"), (np.float32, "f32")):
    dt_, dy_ = dev(np.ascontiguousarray(t.astype(dtype))), dev(np.ascontiguousarray(y.astype(dtype)))
    dll, dg, dst = dev(np.zeros(N, dtype)), dev(np.zeros((N, 1), dtype)), dev(np.zeros(N, np.int32))
    dfm, dfP = dev(np.zeros((N, T, d), dtype)), dev(np.zeros((N, T, d, d), dtype))
    for name, P in (("built-in", Pb), ("as source", Pc)):
        mb = _model_block(P); opts = _opts(cd.EKFHyperParams(), 1)
        ff = getattr(L, f"cdkf_ekf_filter_{sfx}_dev")
        for rep in range(3):
            t0 = time.perf_counter()
            _ffi.check(ff(C.byref(mb.c), C.byref(opts), N, T, dt_, dy_, dll, dfm, dfP, None, None, dst, None)); _ffi.check(L.cdkf_synchronize(None))
            el = time.perf_counter() - t0
        print(f"{name:10s} filter {sfx} d={d} N={N} T={T}: {el*1e3:8.1f} ms  ({L.cdkf_last_kernel().decode()[:60]})", flush=True)
        fn = getattr(L, f"cdkf_ekf_loglik_grad_{sfx}_dev")
        for rep in range(3):
            t0 = time.perf_counter()
            _ffi.check(fn(C.byref(mb.c), C.byref(opts), N, T, dt_, dy_, dll, dg, dst, None)); _ffi.check(L.cdkf_synchronize(None))
            el = time.perf_counter() - t0
        print(f"{name:10s} value + gradient {sfx}:        {el*1e3:8.1f} ms  ({L.cdkf_last_kernel().decode()[:60]})", flush=True)
    _ffi.check(L.cdkf_release_workspace())
