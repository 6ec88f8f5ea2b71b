"""C2 shape with (a) per-trajectory irregular grids (the bench), (b) one irregular grid shared by the batch (no RK
step-count divergence inside a wavefront)."""
import ctypes as C, sys, os, time
sys.path[:0] = [os.path.join(os.path.dirname(__file__), ".."), os.path.join(os.path.dirname(__file__), "..", "oracle")]
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
N, T = 4096, 1000
t, y = make_batch(0, N, T)
def dev(arr=None, nbytes=None):
    p = C.c_void_p(); nb = arr.nbytes if arr is not None else nbytes
    _ffi.check(L.cdkf_malloc(C.byref(p), nb))
    if arr is not None: _ffi.check(L.cdkf_memcpy_h2d(p, arr.ctypes.data_as(C.c_void_p), nb))
    return p
yd = dev(np.ascontiguousarray(y.transpose(1, 2, 0)))
ll, st = dev(nbytes=N * 8), dev(nbytes=N * 4)
bufs = [dev(nbytes=N * T * w * 8) for w in (3, 9, 3, 9)]
for shared in (0, 1):
    td = dev(np.ascontiguousarray(t[0] if shared else t.T))
    opts = _ffi.default_opts(); opts.layout = _ffi.LAYOUT_TCN; opts.t_shared = shared
    for rep in range(4):
        t0 = time.perf_counter()
        _ffi.check(L.cdkf_ekf_filter_f64_dev(C.byref(blk.c), C.byref(opts), N, T, td, yd, ll, *bufs, st, None))
        _ffi.check(L.cdkf_synchronize(None))
        el = time.perf_counter() - t0
    print(f"t_shared={shared}: {el*1e3:.3f} ms -> {N/el:.3e} traj/s")
