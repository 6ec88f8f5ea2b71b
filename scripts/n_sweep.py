"""Headline sweep time against the batch size N (dev helper): python scripts/n_sweep.py <lib.so> <N> [emission_dim]
(emission_dim < 3: the first coordinates are observed, H = I[:m])."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from cd_dynamax_amd._ffi import CdkfModel, CdkfOpts
import bench

L = C.CDLL(sys.argv[1])
reps = 20
N, T, D, M = int(sys.argv[2]), 1000, 3, (int(sys.argv[3]) if len(sys.argv) > 3 else 3)
t_h, y_h = bench.make_batch(0, N, T)
f64 = lambda a: np.ascontiguousarray(a, np.float64)
keep = [f64([10.0, 28.0, 8 / 3]), f64(np.eye(3)), f64(np.eye(3)), f64(np.eye(3)[:M]), f64(np.zeros(M)), f64(np.eye(M)), f64(np.zeros(3)), f64(5 * np.eye(3))]
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
mdl = CdkfModel(1, 3, M, 0, 0, 0, 3, *map(dp, keep))
opts = CdkfOpts(); L.cdkf_default_opts(C.byref(opts)); opts.layout = 2
def dev(a):
    p = C.c_void_p(); assert L.cdkf_malloc(C.byref(p), C.c_int64(a.nbytes)) == 0
    assert L.cdkf_memcpy_h2d(p, a.ctypes.data_as(C.c_void_p), C.c_int64(a.nbytes)) == 0
    return p
t_d, y_d = dev(f64(t_h.T)), dev(f64(y_h[:, :, :M].transpose(1, 2, 0)))
ll, st = dev(np.zeros(N)), dev(np.zeros(N, np.int32))
fm, pm = dev(np.zeros((T, D, N))), dev(np.zeros((T, D, N)))
fP, pP = dev(np.zeros((T, D, D, N))), dev(np.zeros((T, D, D, N)))
L.cdkf_ekf_filter_f64_dev.argtypes = [C.POINTER(CdkfModel), C.POINTER(CdkfOpts), C.c_int64, C.c_int64] + [C.c_void_p] * 9
run = lambda: L.cdkf_ekf_filter_f64_dev(C.byref(mdl), C.byref(opts), N, T, t_d, y_d, ll, fm, fP, pm, pP, st, None)
for _ in range(3):
    assert run() == 0
L.cdkf_synchronize(None)
t0 = time.perf_counter()
for _ in range(reps):
    run()
L.cdkf_synchronize(None)
print("N", N, "m", M, "lanes", os.environ.get("CDKF_LANES_PER_WAVE"), f"{(time.perf_counter() - t0) / reps * 1e3:.4f} ms per sweep")
