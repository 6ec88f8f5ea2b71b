"""Lane-per-trajectory sweeps on the BASELINE config 2 / 3 batch (Lorenz-63, 4096 x 1000 irregular grids): EKF, UKF, EKF smoother,
fp64 and fp32, native layout, device-resident.  CDKF_LANES_PER_WAVE=64 reproduces the full-wavefront grouping."""
import ctypes as C, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np
from bench import make_batch
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi
from cd_dynamax_amd.models import _model_block
L = _ffi.lib()
params = cd.ParamsCDNLGSSM(
    initial=cd.ParamsLGSSMInitial(cd.LearnableVector(np.zeros(3)), cd.LearnableMatrix(5.0 * np.eye(3))),
    dynamics=cd.ParamsCDNLGSSMDynamics(cd.LearnableLorenz63(10.0, 28.0, 8.0 / 3.0), cd.LearnableMatrix(np.eye(3)), cd.LearnableMatrix(np.eye(3)), 2.0),
    emissions=cd.ParamsCDNLGSSMEmissions(cd.LearnableLinear(np.eye(3), np.zeros(3)), cd.LearnableMatrix(np.eye(3))))
blk = _model_block(params)
N, T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 1000
t, y = make_batch(0, N, T)
print("trajectories per wavefront:", L.cdkf_trajectories_per_wavefront(N))
def dev(arr=None, nbytes=None):
    p = C.c_void_p(); nb = arr.nbytes if arr is not None else nbytes
    _ffi.check(L.cdkf_malloc(C.byref(p), nb))
    if arr is not None: _ffi.check(L.cdkf_memcpy_h2d(p, arr.ctypes.data_as(C.c_void_p), nb))
    return p
for dtype, sfx in ((np.float64, "f64"), (np.float32, "f32")):
    s = np.dtype(dtype).itemsize
    yd = dev(np.ascontiguousarray(y.transpose(1, 2, 0).astype(dtype)))
    td = dev(np.ascontiguousarray(t.T.astype(dtype)))
    ll, st = dev(nbytes=N * s), dev(nbytes=N * 4)
    bufs = [dev(nbytes=N * T * w * s) for w in (3, 9, 3, 9)]
    for algo in ("ekf_filter", "ukf_filter", "ekf_smoother"):
        opts = _ffi.default_opts(); opts.layout = _ffi.LAYOUT_TCN
        fn = getattr(L, f"cdkf_{algo}_{sfx}_dev")
        run = lambda: _ffi.check(fn(C.byref(blk.c), C.byref(opts), N, T, td, yd, ll, *bufs, st, None))
        for _ in range(3): run()
        _ffi.check(L.cdkf_synchronize(None))
        t0 = time.perf_counter()
        for _ in range(10): run()
        _ffi.check(L.cdkf_synchronize(None))
        el = (time.perf_counter() - t0) / 10
        print(f"{algo} {sfx}: {el*1e3:.3f} ms -> {N/el:.3e} traj/s")
        if algo != "ekf_smoother":  # log-likelihood only (marginal_log_prob): no moment stores
            run0 = lambda: _ffi.check(fn(C.byref(blk.c), C.byref(opts), N, T, td, yd, ll, None, None, None, None, st, None))
            for _ in range(3): run0()
            _ffi.check(L.cdkf_synchronize(None))
            t0 = time.perf_counter()
            for _ in range(10): run0()
            _ffi.check(L.cdkf_synchronize(None))
            print(f"{algo} {sfx}, log-likelihood only: {(time.perf_counter() - t0) * 100:.3f} ms")
            run1 = lambda: _ffi.check(fn(C.byref(blk.c), C.byref(opts), N, T, td, yd, ll, bufs[0], bufs[1], None, None, st, None))
            for _ in range(3): run1()
            _ffi.check(L.cdkf_synchronize(None))
            t0 = time.perf_counter()
            for _ in range(10): run1()
            _ffi.check(L.cdkf_synchronize(None))
            print(f"{algo} {sfx}, filtered moments only (the smoother's forward sweep): {(time.perf_counter() - t0) * 100:.3f} ms")
    for p in [yd, td, ll, st] + bufs: L.cdkf_free(p)
