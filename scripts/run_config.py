"""Profiling target: ONE of bench.py's configurations, a few launches of each of its kernels (device-resident, torch-free).
Usage (directly after `rocprofv3 ... --`):  python3 scripts/run_config.py <config2|config2_with_smoother|config3|config4|config5|grad> [rounds]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
import bench
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi
from cd_dynamax_amd.models import _model_block
from cd_dynamax_amd._ffi import DeviceArray

name = sys.argv[1]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lib = _ffi.lib()
stream = C.c_void_p()
_ffi.check(lib.cdkf_stream_create(C.byref(stream)))
timer = bench.Timer(lib, _ffi, stream)
t_h, y_h = bench.make_batch(0, bench.N_PER_GPU, bench.T_STEPS)
if name in ("config2", "config2_tn", "grad"):
    blk = _model_block(bench.l63_params(cd))
    opts = _ffi.default_opts(); opts.layout = _ffi.LAYOUT_TCN
    if name == "config2_tn":  # outputs [T,N,comp] (a wavefront's store = 4 x 72 contiguous bytes), inputs still [T,comp,N]
        opts.layout, opts.layout_in = _ffi.LAYOUT_TN, _ffi.LAYOUT_TCN
    N, T = t_h.shape
    t_d = DeviceArray.from_numpy(np.ascontiguousarray(t_h.T)); y_d = DeviceArray.from_numpy(np.ascontiguousarray(y_h.transpose(1, 2, 0)))
    ll = DeviceArray((N,), np.float64); st = DeviceArray.from_numpy(np.zeros(N, np.int32))
    if name == "grad":
        print(bench.value_and_grad(lib, blk, opts, N, T, t_d, y_d, ll, timer, stream, reps=4 * rounds))
    else:
        bufs = [DeviceArray((T, w, N), np.float64) for w in (3, 9, 3, 9)]
        run = lambda: _ffi.check(lib.cdkf_ekf_filter_f64_dev(C.byref(blk.c), C.byref(opts), N, T, t_d.ptr, y_d.ptr, ll.ptr, *[b.ptr for b in bufs], st.ptr, stream))
        print(lib.cdkf_last_kernel().decode(), timer.ms_per_call(run, 4 * rounds), "ms")
else:
    for _ in range(rounds):
        print(bench.other_configs(lib, timer, stream, t_h, y_h, only=name))
