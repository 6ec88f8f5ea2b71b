"""Times the reverse sweep on the Lorenz-96 d = m = 40 slice (value + every gradient): python3 scripts/time_awg.py [N] [T] [reps] [f64|f32]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi
from cd_dynamax_amd.models import _model_block
from cd_dynamax_amd._ffi import DeviceArray
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
T = int(sys.argv[2]) if len(sys.argv) > 2 else 500
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
suf = sys.argv[4] if len(sys.argv) > 4 else "f64"
dt = np.float64 if suf == "f64" else np.float32
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
t_d = DeviceArray.from_numpy(np.ascontiguousarray(t.T, dt)); y_d = DeviceArray.from_numpy(np.ascontiguousarray(y.transpose(1, 2, 0), dt))
ll = DeviceArray((N,), dt); st = DeviceArray.from_numpy(np.zeros(N, np.int32))
g = DeviceArray((N, 1), dt); gm = DeviceArray((N, _ffi.model_grad_size(d, d)), dt)
def run():
    _ffi.check(getattr(lib, f"cdkf_ekf_loglik_grad_all_{suf}_dev")(C.byref(blk.c), C.byref(opts), N, T, t_d.ptr, y_d.ptr, ll.ptr, g.ptr, gm.ptr, st.ptr, None))
run(); _ffi.check(lib.cdkf_synchronize(None))
t0 = time.perf_counter()
for _ in range(reps): run()
_ffi.check(lib.cdkf_synchronize(None))
ms = (time.perf_counter() - t0) / reps * 1e3
print(f"value + every gradient, Lorenz-96 d = m = 40 {suf}, {N} x {T}: {ms:.1f} ms  ({lib.cdkf_last_kernel().decode()})  sum g = {float(g.numpy().sum()):.12g}  sum gm = {float(gm.numpy().sum()):.12g}")
