"""A/B of the config-5 sweeps (MLP drift d = 8, m = 4, 2 x 64; 1024 x 1000 slice) under CDKF_W8_SPLIT / CDKF_ADJ_SPLIT: one child
process per setting (the library reads the switch once), each prints kernel, time and the checks against the oracle on a subset.
    python scripts/gpu_ab_w8split.py [settings ...]     e.g.  0 2 4"""
import ctypes as C, os, subprocess, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    for s in sys.argv[1:]:
        env = dict(os.environ, CDKF_W8_SPLIT=s)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", s], env=env)
    sys.exit(0)
sys.path[:0] = [_R, os.path.join(_R, "oracle"), os.path.join(_R, "tests")]
import numpy as np
import cdkf_oracle as o
from cd_dynamax_amd import _ffi, models
from helpers import mlp_model, params_from
tag = sys.argv[2] if len(sys.argv) > 2 else os.environ.get("CDKF_W8_SPLIT", "default")
L = _ffi.lib()
N, T, d, m = 1024, 1000, 8, 4
mdl = mlp_model(np.random.default_rng(2), d, m, 64)
rng = np.random.default_rng(3)
t = o.irregular_times(rng, N, T, 0.005 * T)
y = rng.standard_normal((N, T, m))
blk = models._model_block(params_from(mdl))
stream = C.c_void_p()
_ffi.check(L.cdkf_stream_create(C.byref(stream)))
ev = [C.c_void_p(), C.c_void_p()]
for e in ev: _ffi.check(L.cdkf_event_create(C.byref(e)))

def timed(run, reps=3):
    run(); _ffi.check(L.cdkf_synchronize(stream))
    _ffi.check(L.cdkf_event_record(ev[0], stream))
    for _ in range(reps): run()
    _ffi.check(L.cdkf_event_record(ev[1], stream))
    ms = C.c_float(); _ffi.check(L.cdkf_event_elapsed_ms(ev[0], ev[1], C.byref(ms)))
    return ms.value / reps

ns = 4
for dtype, suf in ((np.float64, "f64"), (np.float32, "f32")):
    for order in (2, 1):
        opts = _ffi.default_opts(); opts.layout = _ffi.LAYOUT_TN; opts.state_order = order
        td = _ffi.DeviceArray.from_numpy(np.ascontiguousarray(t.T, dtype)); yd = _ffi.DeviceArray.from_numpy(np.ascontiguousarray(y.transpose(1, 0, 2), dtype))
        ll = _ffi.DeviceArray((N,), dtype); st = _ffi.DeviceArray.from_numpy(np.zeros(N, np.int32))
        fm = _ffi.DeviceArray((T, N, d), dtype); fP = _ffi.DeviceArray((T, N, d, d), dtype)
        fn = getattr(L, f"cdkf_ekf_filter_{suf}_dev")
        run0 = lambda: _ffi.check(fn(C.byref(blk.c), C.byref(opts), N, T, td.ptr, yd.ptr, ll.ptr, None, None, None, None, st.ptr, stream))
        ms = timed(run0)
        kern = L.cdkf_last_kernel().decode()
        run1 = lambda: _ffi.check(fn(C.byref(blk.c), C.byref(opts), N, T, td.ptr, yd.ptr, ll.ptr, fm.ptr, fP.ptr, None, None, st.ptr, stream))
        run1(); _ffi.check(L.cdkf_synchronize(stream))
        ref = o.ekf_filter(mdl, t[:ns], y[:ns], state_order={1: "first", 2: "second"}[order])
        llh = ll.numpy()[:ns].astype(np.float64)
        e_ll = np.max(np.abs(llh - ref["marginal_loglik"]) / np.abs(ref["marginal_loglik"]))
        fmh = fm.numpy()[:, :ns].transpose(1, 0, 2).astype(np.float64); fPh = fP.numpy()[:, :ns].transpose(1, 0, 2, 3).astype(np.float64)
        e_m = np.max(np.abs(fmh - ref["filtered_means"])) / np.max(np.abs(ref["filtered_means"]))
        e_P = np.max(np.abs(fPh - ref["filtered_covariances"])) / np.max(np.abs(ref["filtered_covariances"]))
        nbad = int(np.count_nonzero(st.numpy()))
        # value + every gradient
        n_th, n_md = blk.theta.size, _ffi.model_grad_size(d, m)
        g = _ffi.DeviceArray((N, n_th), dtype); gm = _ffi.DeviceArray((N, n_md), dtype)
        o2 = _ffi.default_opts(); o2.layout = _ffi.LAYOUT_TCN; o2.state_order = order
        yg = _ffi.DeviceArray.from_numpy(np.ascontiguousarray(y.transpose(1, 2, 0), dtype))
        gfn = getattr(L, f"cdkf_ekf_loglik_grad_all_{suf}_dev")
        rung = lambda: _ffi.check(gfn(C.byref(blk.c), C.byref(o2), N, T, td.ptr, yg.ptr, ll.ptr, g.ptr, gm.ptr, st.ptr, stream))
        msg = timed(rung, 2)
        kg = L.cdkf_last_kernel().decode()
        gs = g.numpy().astype(np.float64).sum(0)
        print(f"[split {tag}] {suf} order {order}: forward {ms:7.2f} ms ({kern}); value+grad {msg:7.2f} ms ({kg}); "
              f"vs oracle on {ns}: ll {e_ll:.1e} means {e_m:.1e} covs {e_P:.1e}; flags {nbad}; |grad sum| {np.linalg.norm(gs):.9e} ll sum {ll.numpy().astype(np.float64).sum():.9e}", flush=True)
        for a_ in (td, yd, ll, st, fm, fP, g, gm, yg): a_.free()
