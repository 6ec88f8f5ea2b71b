"""Time value + gradient of the EKF log-likelihood for Lorenz-96 at larger state dimensions (forward sweep + ekf_adjoint_wg_kernel):
python3 scripts/gpu_time_grad_l96.py [d=40] [n=2048] [t=500] [all]"""
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
d, N, T = arg("d", 40), arg("n", 2048), arg("t", 500)
m = arg("m", d)
rng = np.random.default_rng(1)
mdl = lorenz96_model(d, m)
u = rng.uniform(0, 1, (N, T)); t = np.cumsum(u, 1); t = t / t[:, -1:] * (0.005 * T)
y = 8.0 + rng.standard_normal((N, T, m))
for dtype, sfx in ((np.float64, "f64"), (np.float32, "f32")):
    mb = _model_block(params_from(mdl)); opts = _opts(cd.EKFHyperParams(), 1)
    def dev(a):
        p = C.c_void_p(); _ffi.check(L.cdkf_malloc(C.byref(p), a.nbytes)); _ffi.check(L.cdkf_memcpy_h2d(p, a.ctypes.data_as(C.c_void_p), a.nbytes)); return p
    dt_, dy_ = dev(np.ascontiguousarray(t.astype(dtype))), dev(np.ascontiguousarray(y.astype(dtype)))
    dll, dg, dst = dev(np.zeros(N, dtype)), dev(np.zeros((N, 1), dtype)), dev(np.zeros(N, np.int32))
    fn = getattr(L, f"cdkf_ekf_loglik_grad_{sfx}_dev")
    for rep in range(3):
        t0 = time.perf_counter()
        _ffi.check(fn(C.byref(mb.c), C.byref(opts), N, T, dt_, dy_, dll, dg, dst, None)); _ffi.check(L.cdkf_synchronize(None))
        el = time.perf_counter() - t0
    print(f"value + d/dF {sfx} d={d} m={m} N={N} T={T}: {el*1e3:.1f} ms  ({L.cdkf_last_kernel().decode()[:40]})", flush=True)
    if "all" in sys.argv[1:]:
        dgm = dev(np.zeros((N, _ffi.model_grad_size(d, m)), dtype))
        fa = getattr(L, f"cdkf_ekf_loglik_grad_all_{sfx}_dev")
        for rep in range(2):
            t0 = time.perf_counter()
            _ffi.check(fa(C.byref(mb.c), C.byref(opts), N, T, dt_, dy_, dll, dg, dgm, dst, None)); _ffi.check(L.cdkf_synchronize(None))
            el = time.perf_counter() - t0
        print(f"value + every gradient {sfx}: {el*1e3:.1f} ms", flush=True)
    _ffi.check(L.cdkf_release_workspace())
