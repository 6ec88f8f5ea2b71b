"""Quick GPU sanity + timing (development aid, not a test): HIP path vs the oracle on small Lorenz-63
cases, then a timing of the C2 configuration."""
import ctypes as C
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi


def params_from(mdl):
    dr = mdl.drift
    if dr.kind == "lorenz63":
        drift = cd.LearnableLorenz63(float(dr.sigma), float(dr.rho), float(dr.beta))
    elif dr.kind == "linear":
        drift = cd.LearnableLinear(dr.W, dr.b)
    return cd.ParamsCDNLGSSM(
        initial=cd.ParamsLGSSMInitial(cd.LearnableVector(mdl.m0), cd.LearnableMatrix(mdl.P0)),
        dynamics=cd.ParamsCDNLGSSMDynamics(drift, cd.LearnableMatrix(mdl.L), cd.LearnableMatrix(mdl.Qc), 2.0),
        emissions=cd.ParamsCDNLGSSMEmissions(cd.LearnableLinear(mdl.H, mdl.bias), cd.LearnableMatrix(mdl.R)))


def relerr(a, b):
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))


rng = np.random.default_rng(0)
for mobs in (3, 1):
    mdl = o.lorenz63_model(mobs)
    N, T = 70, 150
    t = o.irregular_times(rng, N, T, 0.005 * T * 1.3)
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    for dt in (np.float64, np.float32):
        ref = o.ekf_filter(mdl, t, y, dtype=np.float64)
        got = cd.cdnlgssm_filter(P, y.astype(dt), t[..., None].astype(dt), cd.EKFHyperParams())
        print(f"EKF m={mobs} {dt.__name__}: ll {relerr(got.marginal_loglik, ref['marginal_loglik']):.2e} "
              f"fm {relerr(got.filtered_means, ref['filtered_means']):.2e} fP {relerr(got.filtered_covariances, ref['filtered_covariances']):.2e} "
              f"pm {relerr(got.predicted_means, ref['predicted_means']):.2e} pP {relerr(got.predicted_covariances, ref['predicted_covariances']):.2e}")
        ref = o.ukf_filter(mdl, t, y, dtype=np.float64)
        got = cd.cdnlgssm_filter(P, y.astype(dt), t[..., None].astype(dt), cd.UKFHyperParams())
        print(f"UKF m={mobs} {dt.__name__}: ll {relerr(got.marginal_loglik, ref['marginal_loglik']):.2e} "
              f"fm {relerr(got.filtered_means, ref['filtered_means']):.2e} fP {relerr(got.filtered_covariances, ref['filtered_covariances']):.2e} "
              f"pP {relerr(got.predicted_covariances, ref['predicted_covariances']):.2e}")
        ref = o.ekf_smoother(mdl, t, y, dtype=np.float64)
        got = cd.cdnlgssm_smoother(P, y.astype(dt), t[..., None].astype(dt), cd.EKFHyperParams())
        print(f"EKS m={mobs} {dt.__name__}: sm {relerr(got.smoothed_means, ref['smoothed_means']):.2e} "
              f"sP {relerr(got.smoothed_covariances, ref['smoothed_covariances']):.2e}")

# ---- timing at C2: N=4096, T=1000 ----
L = _ffi.lib()
for N in (4096, 65536, 262144):
    T = 1000
    mdl = o.lorenz63_model(3)
    t = o.irregular_times(rng, N, T, 0.005 * T)
    y = rng.standard_normal((N, T, 3)) * 5
    blk = cd.models._model_block(params_from(mdl))
    for dt, suf in ((np.float64, "f64"), (np.float32, "f32")):
        sz = np.dtype(dt).itemsize
        def dev(arr=None, nbytes=None):
            p = C.c_void_p()
            nb = arr.nbytes if arr is not None else nbytes
            _ffi.check(L.cdkf_malloc(C.byref(p), nb))
            if arr is not None:
                _ffi.check(L.cdkf_memcpy_h2d(p, arr.ctypes.data_as(C.c_void_p), nb))
            return p
        td, yd = dev(np.ascontiguousarray(t.T, dt)), dev(np.ascontiguousarray(y.transpose(1, 2, 0), dt))  # TCN
        ll, st = dev(nbytes=N * sz), dev(nbytes=N * 4)
        fm, pm = dev(nbytes=N * T * 3 * sz), dev(nbytes=N * T * 3 * sz)
        fP, pP = dev(nbytes=N * T * 9 * sz), dev(nbytes=N * T * 9 * sz)
        opts = _ffi.default_opts()
        opts.layout = _ffi.LAYOUT_TCN
        for algo in ("ekf_filter", "ukf_filter"):
            fn = getattr(L, f"cdkf_{algo}_{suf}_dev")
            for full in (True, False):
                args = (fm, fP, pm, pP) if full else (None, None, None, None)
                for rep in range(3):
                    t0 = time.perf_counter()
                    _ffi.check(fn(C.byref(blk.c), C.byref(opts), N, T, td, yd, ll, *args, st, None))
                    _ffi.check(L.cdkf_synchronize(None))
                    el = time.perf_counter() - t0
                print(f"{algo} {suf} N={N} T={T} full_out={full}: {el*1e3:.3f} ms -> {N/el:.3e} traj/s")
        for p in (td, yd, ll, st, fm, pm, fP, pP):
            L.cdkf_free(p)
