"""Profiling target: the workgroup reverse sweep (ekf_adjoint_wg_kernel) on a Lorenz-96 d = m = 40 slice, one of its two forms:
    python3 scripts/awg_traffic.py <all|drift> [N] [T]
`all` = value + every gradient (cdkf_ekf_loglik_grad_all: the model block accumulates per step), `drift` = the drift block alone."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi
from cd_dynamax_amd.models import _model_block
from cd_dynamax_amd._ffi import DeviceArray
mode = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
T = int(sys.argv[3]) if len(sys.argv) > 3 else 100
d = 40
eye = np.eye
l96 = cd.ParamsCDNLGSSM(
    initial=cd.ParamsLGSSMInitial(cd.LearnableVector(8.0 * np.ones(d)), cd.LearnableMatrix(eye(d))),
    dynamics=cd.ParamsCDNLGSSMDynamics(cd.LearnableLorenz96(8.0), cd.LearnableMatrix(eye(d)), cd.LearnableMatrix(eye(d)), 2.0),
    emissions=cd.ParamsCDNLGSSMEmissions(cd.LearnableLinear(eye(d), np.zeros(d)), cd.LearnableMatrix(eye(d))))
rng = np.random.default_rng(1)
u = rng.uniform(0.0, 1.0, size=(N, T)); s = np.cumsum(u, axis=1); t = s / s[:, -1:] * (0.005 * T)
y = 8.0 + rng.standard_normal((N, T, d))
lib = _ffi.lib()
blk = _model_block(l96)
opts = _ffi.default_opts(); opts.layout = _ffi.LAYOUT_TCN
t_d = DeviceArray.from_numpy(np.ascontiguousarray(t.T)); y_d = DeviceArray.from_numpy(np.ascontiguousarray(y.transpose(1, 2, 0)))
ll = DeviceArray((N,), np.float64); st = DeviceArray.from_numpy(np.zeros(N, np.int32))
g = DeviceArray((N, 1), np.float64); gm = DeviceArray((N, _ffi.model_grad_size(d, d)), np.float64)
for _ in range(3):
    if mode == "all":
        _ffi.check(lib.cdkf_ekf_loglik_grad_all_f64_dev(C.byref(blk.c), C.byref(opts), N, T, t_d.ptr, y_d.ptr, ll.ptr, g.ptr, gm.ptr, st.ptr, None))
    else:
        _ffi.check(lib.cdkf_ekf_loglik_grad_f64_dev(C.byref(blk.c), C.byref(opts), N, T, t_d.ptr, y_d.ptr, ll.ptr, g.ptr, st.ptr, None))
_ffi.check(lib.cdkf_synchronize(None))
print(mode, lib.cdkf_last_kernel().decode(), float(g.numpy().sum()))
