"""ctypes wrapper around oracle/_build/libcdkf_oracle.so (the C restatement, cdkf_oracle.c).
TEST INFRASTRUCTURE ONLY -- used by tests/ and by bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CDKF_ORACLE_SO: the sanitizer build of scripts/sanitize_cpu.sh
_SO = os.environ.get("CDKF_ORACLE_SO") or os.path.join(_HERE, "_build", "libcdkf_oracle.so")
_KIND = {"linear": 0, "lorenz63": 1, "lorenz96": 2}
_ORDER = {"zeroth": 0, "first": 1, "second": 2}


_native = None  # set by build_native(): the library compiled on (and for) this machine


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def build_native():
    """Compile the restatement with -march=native on the machine that is going to time it (the shipped library is built elsewhere
    for a portable ISA level).  Returns True when that build is now in use; on any failure the portable library stays."""
    global _native
    out = os.path.join(_HERE, "_build", "native")
    try:
        subprocess.check_call(["make", "-C", _HERE, "-s", "ARCH=native", "OUT=_build/native"], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
        _native = C.CDLL(os.path.join(out, "libcdkf_oracle.so"))
        return True
    except Exception:
        _native = None
        return False


def _lib():
    if _native is not None:
        return _native
    if not os.path.exists(_SO):
        build()
    return C.CDLL(_SO)


def ekf_filter(mdl, t, y, state_order="second", num_iter=1, dt0=0.01, dt_final=1e-10, max_steps=100000,
               cov_rescaling=1.0, dtype=np.float64, nthreads=0, outputs=True):
    """Same contract as cdkf_oracle.ekf_filter (t [N,T], y [N,T,m]); nthreads = 0 -> OpenMP default."""
    dtype = np.dtype(dtype)
    fn = getattr(_lib(), "cdkf_oracle_ekf_filter_f64" if dtype == np.float64 else "cdkf_oracle_ekf_filter_f32")
    t = np.ascontiguousarray(t, dtype)
    y = np.ascontiguousarray(y, dtype)
    N, T, m = y.shape
    d = mdl.d
    f64 = lambda a: np.ascontiguousarray(a, np.float64)
    th, L, Qc, H, hb, R, m0, P0 = map(f64, (mdl.drift.theta(), mdl.L, mdl.Qc, mdl.H, mdl.bias, mdl.R, mdl.m0, mdl.P0))
    ll = np.zeros(N, dtype)
    shapes = [(N, T, d), (N, T, d, d), (N, T, d), (N, T, d, d)]
    outs = [np.full(s, 0, dtype) if outputs else None for s in shapes]  # np.full touches the pages (no lazy calloc)
    p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    fn.argtypes = [C.c_int] * 3 + [C.c_void_p] * 8 + [C.c_int, C.c_int, C.c_double, C.c_double, C.c_long, C.c_double,
                                                        C.c_long, C.c_long] + [C.c_void_p] * 7 + [C.c_int]
    import time
    t_start = time.perf_counter()
    rc = fn(_KIND[mdl.drift.kind], d, m, p(th), p(L), p(Qc), p(H), p(hb), p(R), p(m0), p(P0), _ORDER[state_order],
            num_iter, dt0, dt_final, max_steps, cov_rescaling, N, T, p(t), p(y), p(ll), *[p(o) for o in outs], nthreads)
    seconds = time.perf_counter() - t_start
    if rc != 0:
        raise ValueError("cdkf_oracle_ekf_filter: unsupported sizes")
    keys = ["filtered_means", "filtered_covariances", "predicted_means", "predicted_covariances"]
    res = {"marginal_loglik": ll, "_seconds": seconds}  # wall time of the C call alone (buffers pre-touched)
    if outputs:
        res.update(dict(zip(keys, outs)))
    return res
