"""Linear drift, state_dim 3, 4096 x 1000 (fp64, H = I, all four outputs): sixteen-lane sweep vs the lane-per-trajectory kernel
(CDKF_NO_LPE=1 in a second process)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
import bench
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi
from cd_dynamax_amd.models import _model_block
from cd_dynamax_amd._ffi import DeviceArray
lib = _ffi.lib()
rng = np.random.default_rng(0)
W = -0.5 * np.eye(3) + 0.3 * rng.standard_normal((3, 3)) / np.sqrt(3)
params = cd.ParamsCDNLGSSM(
    initial=cd.ParamsLGSSMInitial(cd.LearnableVector(np.zeros(3)), cd.LearnableMatrix(np.eye(3))),
    dynamics=cd.ParamsCDNLGSSMDynamics(cd.LearnableLinear(W, 0.1 * np.ones(3)), cd.LearnableMatrix(np.eye(3)), cd.LearnableMatrix(np.eye(3)), 2.0),
    emissions=cd.ParamsCDNLGSSMEmissions(cd.LearnableLinear(np.eye(3), np.zeros(3)), cd.LearnableMatrix(np.eye(3))))
blk = _model_block(params)
t_h, y_h = bench.make_batch(0, 4096, 1000)
N, T = t_h.shape
stream = C.c_void_p(); _ffi.check(lib.cdkf_stream_create(C.byref(stream)))
timer = bench.Timer(lib, _ffi, stream)
t_d = DeviceArray.from_numpy(np.ascontiguousarray(t_h.T)); y_d = DeviceArray.from_numpy(np.ascontiguousarray(y_h.transpose(1, 2, 0)))
ll = DeviceArray((N,), np.float64); st = DeviceArray.from_numpy(np.zeros(N, np.int32))
bufs = [DeviceArray((T, N, w), np.float64) for w in (3, 9, 3, 9)]
opts = _ffi.default_opts(); opts.layout, opts.layout_in = _ffi.LAYOUT_TN, _ffi.LAYOUT_TCN
for algo in ("ekf_filter", "ukf_filter"):
    fn = getattr(lib, f"cdkf_{algo}_f64_dev")
    run = lambda: _ffi.check(fn(C.byref(blk.c), C.byref(opts), N, T, t_d.ptr, y_d.ptr, ll.ptr, *[b.ptr for b in bufs], st.ptr, stream))
    for _ in range(30): run()
    print(algo, lib.cdkf_last_kernel().decode()[:70], f"{timer.ms_per_call(run, 20):.3f} ms")
