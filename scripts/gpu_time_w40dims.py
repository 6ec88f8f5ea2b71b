"""Wavefront-per-trajectory Lorenz-96 sweeps against the workgroup kernels at other state dimensions: python3 scripts/gpu_time_w40dims.py
(the kernel choice is made per call: CDKF_NO_WAVE40 is read at every launch)."""
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
stream = C.c_void_p(); _ffi.check(lib.cdkf_stream_create(C.byref(stream)))
timer = bench.Timer(lib, _ffi, stream)
rng = np.random.default_rng(1)
n, T = 2048, 200
for d in (12, 20, 28, 36, 40):
    eye = np.eye
    P = cd.ParamsCDNLGSSM(
        initial=cd.ParamsLGSSMInitial(cd.LearnableVector(8.0 * np.ones(d)), cd.LearnableMatrix(eye(d))),
        dynamics=cd.ParamsCDNLGSSMDynamics(cd.LearnableLorenz96(8.0), cd.LearnableMatrix(eye(d)), cd.LearnableMatrix(eye(d)), 2.0),
        emissions=cd.ParamsCDNLGSSMEmissions(cd.LearnableLinear(eye(d), np.zeros(d)), cd.LearnableMatrix(eye(d))))
    blk = _model_block(P); opts = _ffi.default_opts(); opts.layout = _ffi.LAYOUT_TN
    u = rng.uniform(0, 1, (n, T)); t = np.cumsum(u, 1); t = t / t[:, -1:] * (0.005 * T)
    y = 8.0 + rng.standard_normal((n, T, d))
    t_d = DeviceArray.from_numpy(np.ascontiguousarray(t.T)); y_d = DeviceArray.from_numpy(np.ascontiguousarray(y.transpose(1, 0, 2)))
    ll = DeviceArray((n,), np.float64); st = DeviceArray.from_numpy(np.zeros(n, np.int32))
    bufs = [DeviceArray((n * T * w,), np.float64) for w in (d, d * d, d, d * d)]
    row = {}
    for label, env in (("wave", None), ("workgroup", "1")):
        if env: os.environ["CDKF_NO_WAVE40"] = env
        else: os.environ.pop("CDKF_NO_WAVE40", None)
        for algo in ("ekf_filter", "ekf_smoother"):
            fn = getattr(lib, f"cdkf_{algo}_f64_dev")
            run = lambda: _ffi.check(fn(C.byref(blk.c), C.byref(opts), n, T, t_d.ptr, y_d.ptr, ll.ptr, *[b.ptr for b in bufs], st.ptr, stream))
            row[(label, algo)] = (timer.ms_per_call(run, 2), lib.cdkf_last_kernel().decode()[:28])
    os.environ.pop("CDKF_NO_WAVE40", None)
    print(f"d={d:2d} 2048x200: filter wave {row[('wave','ekf_filter')][0]:7.2f} ms / workgroup {row[('workgroup','ekf_filter')][0]:7.2f} ms;"
          f" filter+smoother {row[('wave','ekf_smoother')][0]:7.2f} / {row[('workgroup','ekf_smoother')][0]:7.2f} ms   [{row[('wave','ekf_filter')][1]} | {row[('workgroup','ekf_filter')][1]}]", flush=True)
    for a in [t_d, y_d, ll, st] + bufs: a.free()
