"""Time the C2-sized log-likelihood + gradient sweep (dev helper)."""
import ctypes as C, sys, time
import numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import cdkf_oracle as o
from cd_dynamax_amd import _ffi
from cd_dynamax_amd.models import _model_block, _opts
import cd_dynamax_amd as cd
from helpers import params_from

L = _ffi.lib()
M = next((int(a[2:]) for a in sys.argv[1:] if a.startswith("m=")), 3)  # emission dimension (H = I[:M])
for dtype, sfx in ((np.float64, "f64"), (np.float32, "f32")):
    rng = np.random.default_rng(0)
    N, T = next((int(a[2:]) for a in sys.argv[1:] if a.startswith("n=")), 4096), 1000
    mdl = o.lorenz63_model(M)
    if "bench" in sys.argv[1:]:  # the benchmark's batch (every gap <= dt0: one step per interval)
        import bench
        t, y = bench.make_batch(7, N, T)
        t, y = t.astype(dtype), y[..., :M].astype(dtype)
    else:
        t = o.irregular_times(rng, N, T, 0.01).astype(dtype)
        y = rng.standard_normal((N, T, M)).astype(dtype) * 5
    mb = _model_block(params_from(mdl)); opts = _opts(cd.EKFHyperParams(), 1); opts.layout = _ffi.LAYOUT_TCN
    tt = np.ascontiguousarray(t.T); yy = np.ascontiguousarray(y.transpose(1, 2, 0))
    def dev(a):
        p = C.c_void_p(); _ffi.check(L.cdkf_malloc(C.byref(p), a.nbytes)); _ffi.check(L.cdkf_memcpy_h2d(p, a.ctypes.data_as(C.c_void_p), a.nbytes)); return p
    dt_, dy_ = dev(tt), dev(yy)
    dll, dg, dst = dev(np.zeros(N, dtype)), dev(np.zeros((N, 3), dtype)), dev(np.zeros(N, np.int32))
    fn = getattr(L, f"cdkf_ekf_loglik_grad_{sfx}_dev")
    for rep in range(4):
        t0 = time.perf_counter()
        _ffi.check(fn(C.byref(mb.c), C.byref(opts), N, T, dt_, dy_, dll, dg, dst, None)); _ffi.check(L.cdkf_synchronize(None))
        el = time.perf_counter() - t0
    print(f"grad {sfx} N={N} T={T} m={M}: {el*1e3:.2f} ms  ({L.cdkf_last_kernel().decode()[:50]})")
    if "all" in sys.argv[1:]:  # every parameter of the model (the reverse sweep of cdkf_adjoint_kernels.h)
        dgm = dev(np.zeros((N, _ffi.model_grad_size(3, M)), dtype))
        fa = getattr(L, f"cdkf_ekf_loglik_grad_all_{sfx}_dev")
        for rep in range(3):
            t0 = time.perf_counter()
            _ffi.check(fa(C.byref(mb.c), C.byref(opts), N, T, dt_, dy_, dll, dg, dgm, dst, None)); _ffi.check(L.cdkf_synchronize(None))
            el = time.perf_counter() - t0
        print(f"grad_all {sfx} N={N} T={T}: {el*1e3:.2f} ms  ({L.cdkf_last_kernel().decode()[:60]})")
