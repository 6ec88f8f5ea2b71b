"""Parity of the HIP path (through the C ABI) with the CPU oracle and the golden vectors.  GPU only.

Tolerances: the fp64 engine must agree with the fp64 oracle to 1e-9 max-relative (observed ~1e-14; the
north-star bar is 1e-5); the fp32 engine is compared with the SAME fp64 oracle at 1e-5 (observed ~1e-6; the
reference itself runs in float32)."""
import ctypes as C

import numpy as np
import pytest

import cd_dynamax_amd as cd
import cdkf_oracle as o
from cd_dynamax_amd import _ffi, models
from helpers import FILTER_KEYS, GOLDEN, linear_model, load_golden, lorenz96_model, model_from_fixture, params_from, relerr

pytestmark = pytest.mark.gpu

TOL = {np.float64: 1e-9, np.float32: 1e-5}  # fp32: the north-star's bar (BASELINE.json), observed ~1e-6


def _check_filter(post, ref, tol):
    ll_err = np.max(np.abs(np.asarray(post.marginal_loglik, np.float64) - ref["marginal_loglik"]) /
                    np.abs(ref["marginal_loglik"]))
    assert ll_err < tol, f"marginal_loglik rel err {ll_err:.2e}"
    for k in FILTER_KEYS:
        e = relerr(getattr(post, k), ref[k])
        assert e < tol, f"{k}: {e:.2e}"


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("m_obs", [3, 1])
def test_lorenz63_ekf_ukf_eks(hip_lib, dtype, m_obs):
    rng = np.random.default_rng(m_obs)
    mdl = o.lorenz63_model(m_obs)
    N, T = 70, 120  # N not a multiple of 64: a partially filled wavefront
    t = o.irregular_times(rng, N, T, 0.0065 * T)  # ~25 % of the gaps exceed dt0 -> 2 RK steps, lanes diverge
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    tol = TOL[dtype]
    for order in ("first", "second"):
        ref = o.ekf_filter(mdl, t, y, state_order=order)
        _check_filter(cd.cdnlgssm_filter(P, y.astype(dtype), t[..., None], cd.EKFHyperParams(state_order=order)), ref, tol)
    ref = o.ukf_filter(mdl, t, y)
    # (fp64: the UKF's extra Cholesky per stage costs a digit; fp32: the north star's 1e-5 -- observed 3e-7 .. 1.2e-6)
    _check_filter(cd.cdnlgssm_filter(P, y.astype(dtype), t[..., None], cd.UKFHyperParams()), ref, tol * 10 if dtype == np.float64 else 1e-5)
    ref = o.ekf_smoother(mdl, t, y)
    post = cd.cdnlgssm_smoother(P, y.astype(dtype), t[..., None])
    assert relerr(post.smoothed_means, ref["smoothed_means"]) < tol
    assert relerr(post.smoothed_covariances, ref["smoothed_covariances"]) < tol
    assert relerr(post.filtered_means, ref["filtered_means"]) < tol


@pytest.mark.parametrize("d,m", [(1, 1), (2, 1), (2, 2), (2, 6), (3, 1), (3, 3), (4, 2), (4, 4)])
def test_linear_drift_shapes(hip_lib, d, m):
    rng = np.random.default_rng(10 * d + m)
    mdl = linear_model(rng, d, m)
    N, T = 5, 40
    t = o.irregular_times(rng, N, T, 2.0)  # mean gap 0.05: ~5 RK steps per interval
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    _check_filter(cd.cdnlgssm_filter(P, y, t[..., None]), o.ekf_filter(mdl, t, y), 1e-9)
    _check_filter(cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams()), o.ukf_filter(mdl, t, y), 1e-8)
    ref = o.ekf_smoother(mdl, t, y)
    post = cd.cdnlgssm_smoother(P, y, t[..., None])
    assert relerr(post.smoothed_covariances, ref["smoothed_covariances"]) < 1e-9


@pytest.mark.parametrize("name", GOLDEN)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_golden_vectors(hip_lib, name, dtype):
    g = load_golden(name)
    mdl = model_from_fixture(g)
    P = params_from(mdl)
    s = int(g["stride"])
    tol = 1e-9 if dtype == np.float64 else 1e-5
    dtf = float(g["dt_final"])
    y, t = g["y"].astype(dtype), g["t"][..., None]
    for order in ("first", "second"):
        if f"ekf_{order}_ll" not in g:
            continue
        post = cd.cdnlgssm_filter(P, y, t, cd.EKFHyperParams(dt_final=dtf, state_order=order))
        assert relerr(post.marginal_loglik, g[f"ekf_{order}_ll"]) < tol
        for k in FILTER_KEYS:
            assert relerr(getattr(post, k)[:, ::s], g[f"ekf_{order}_{k}"]) < tol, (order, k)
    post = cd.cdnlgssm_filter(P, y, t, cd.UKFHyperParams(dt_final=dtf))
    tol_u = tol * 10 if tol < 1e-6 else 1e-5   # (fp32: the north star's bar)
    assert relerr(post.marginal_loglik, g["ukf_ll"]) < tol_u
    for k in FILTER_KEYS:
        assert relerr(getattr(post, k)[:, ::s], g[f"ukf_{k}"]) < tol_u, k
    post = cd.cdnlgssm_smoother(P, y, t, cd.EKFHyperParams(dt_final=dtf))
    assert relerr(post.smoothed_means[:, ::s], g["eks_smoothed_means"]) < tol
    assert relerr(post.smoothed_covariances[:, ::s], g["eks_smoothed_covariances"]) < tol


def test_golden_regular_grid_via_t_emissions_none(hip_lib):
    g = load_golden("linear_d2_m6_regular")
    P = params_from(model_from_fixture(g))
    s = int(g["stride"])
    post = cd.cdnlgssm_filter(P, g["y"][0], None)  # t_emissions=None == arange(T), last interval 1
    assert relerr(post.predicted_covariances[::s], g["ekf_second_predicted_covariances"][0]) < 1e-9
    assert relerr(post.marginal_loglik, g["ekf_second_ll"][0]) < 1e-9


def test_options_num_iter_zeroth_order_subsets(hip_lib):
    rng = np.random.default_rng(5)
    mdl = o.lorenz63_model(2)
    N, T = 4, 30
    t = o.irregular_times(rng, N, T, 0.4)
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    ref = o.ekf_filter(mdl, t, y, num_iter=2)
    _check_filter(cd.cdnlgssm_filter(P, y, t[..., None], num_iter=2), ref, 1e-9)
    ref = o.ekf_filter(mdl, t, y, state_order="zeroth", cov_rescaling=0.7)
    _check_filter(cd.cdnlgssm_filter(P, y, t[..., None], cd.EKFHyperParams(state_order="zeroth", cov_rescaling=0.7)),
                  ref, 1e-9)
    ref = o.ekf_filter(mdl, t, y, dt0=0.003)
    post = cd.cdnlgssm_filter(P, y, t[..., None], cd.EKFHyperParams(diffeqsolve_settings={"dt0": 0.003}),
                              output_fields=["predicted_means"])
    assert post.filtered_means is None and relerr(post.predicted_means, ref["predicted_means"]) < 1e-9
    ref = o.ukf_filter(mdl, t, y, alpha=1.2, beta=1.5, kappa=0.5)
    _check_filter(cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams(alpha=1.2, beta=1.5, kappa=0.5)), ref, 1e-8)


def test_sixteen_lane_kernel_paths(hip_lib):
    """Small Lorenz-63 batches with H = I run on filter_lpe_kernel (cdkf_lpe_kernels.h).  Its update has two forms:
    inside the lane grid (num_iter = 1, symmetric R) and the per-lane fallback (iterated updates; an emission covariance
    that is not exactly symmetric).  Both against the oracle, odd and even T (the time loop is unrolled by two), N not a
    multiple of the four trajectories per wavefront, all three output modes."""
    rng = np.random.default_rng(12)
    mdl = o.lorenz63_model(3)
    P = params_from(mdl)
    for N, T in ((7, 31), (6, 30), (1, 2), (3, 1)):
        t = o.irregular_times(rng, N, T, 0.008 * T)
        y = o.simulate(mdl, t, rng)
        _check_filter(cd.cdnlgssm_filter(P, y, t[..., None]), o.ekf_filter(mdl, t, y), 1e-9)
        ref2 = o.ekf_filter(mdl, t, y, num_iter=2)
        _check_filter(cd.cdnlgssm_filter(P, y, t[..., None], num_iter=2), ref2, 1e-9)
        post = cd.cdnlgssm_filter(P, y, t[..., None], output_fields=[])                        # log-likelihood only
        assert relerr(post.marginal_loglik, o.ekf_filter(mdl, t, y)["marginal_loglik"]) < 1e-9
        sm = cd.cdnlgssm_smoother(P, y, t[..., None])                                           # filtered moments only
        assert relerr(sm.smoothed_covariances, o.ekf_smoother(mdl, t, y)["smoothed_covariances"]) < 1e-9
    # long gaps (up to ~12 Dormand-Prince steps per interval, different per trajectory), a non-default step, fp32, shared grid
    t = o.irregular_times(rng, 9, 25, 1.5)
    y = o.simulate(mdl, t, rng)
    hyp = cd.EKFHyperParams(diffeqsolve_settings={"dt0": 0.007})
    _check_filter(cd.cdnlgssm_filter(P, y, t[..., None], hyp), o.ekf_filter(mdl, t, y, dt0=0.007), 1e-9)
    _check_filter(cd.cdnlgssm_filter(P, y.astype(np.float32), t[..., None]), o.ekf_filter(mdl, t, y), 1e-5)
    ts = np.broadcast_to(t[0], t.shape)
    _check_filter(cd.cdnlgssm_filter(P, y, t[0][:, None]), o.ekf_filter(mdl, ts, y), 1e-9)
    sm = cd.cdnlgssm_smoother(P, y, t[..., None], hyp)
    ref = o.ekf_smoother(mdl, t, y, dt0=0.007)
    assert relerr(sm.smoothed_means, ref["smoothed_means"]) < 1e-9 and relerr(sm.smoothed_covariances, ref["smoothed_covariances"]) < 1e-9
    Rn = np.eye(3) + 0.05 * np.triu(np.ones((3, 3)), 1)                                         # not symmetric
    skew = o.Model(mdl.drift, mdl.L, mdl.Qc, mdl.H, mdl.bias, Rn, mdl.m0, mdl.P0)
    t = o.irregular_times(rng, 5, 20, 0.15)
    y = o.simulate(mdl, t, rng)
    _check_filter(cd.cdnlgssm_filter(params_from(skew), y, t[..., None]), o.ekf_filter(skew, t, y), 1e-9)


def test_edge_cases_T1_N1_duplicates_long_gaps(hip_lib):
    rng = np.random.default_rng(6)
    mdl = o.lorenz63_model(3)
    P = params_from(mdl)
    # T = 1: only the update and the dt_final predict
    y = rng.standard_normal((2, 1, 3))
    t = np.array([[0.3], [1.7]])
    _check_filter(cd.cdnlgssm_filter(P, y, t[..., None]), o.ekf_filter(mdl, t, y), 1e-9)
    sm = cd.cdnlgssm_smoother(P, y, t[..., None])
    np.testing.assert_array_equal(sm.smoothed_means, sm.filtered_means)
    # N = 1 unbatched, repeated time stamps (zero-length intervals), a 0.5-long gap (50 RK steps)
    t1 = np.array([0.0, 0.0, 0.004, 0.004, 0.504, 0.51, 0.51 + 1e-12, 0.53])
    y1 = o.simulate(mdl, t1[None], rng)[0]
    ref = o.ekf_filter(mdl, t1[None], y1[None])
    post = cd.cdnlgssm_filter(P, y1, t1[:, None])
    assert post.filtered_means.shape == (8, 3)
    for k in FILTER_KEYS:
        assert relerr(getattr(post, k), ref[k][0]) < 1e-9, k
    ref = o.ekf_smoother(mdl, t1[None], y1[None])
    assert relerr(cd.cdnlgssm_smoother(P, y1, t1[:, None]).smoothed_means, ref["smoothed_means"][0]) < 1e-9


def _dev(lib, arr=None, nbytes=None):
    p = C.c_void_p()
    nb = arr.nbytes if arr is not None else nbytes
    _ffi.check(lib.cdkf_malloc(C.byref(p), nb))
    if arr is not None:
        _ffi.check(lib.cdkf_memcpy_h2d(p, arr.ctypes.data_as(C.c_void_p), nb))
    return p


def _to_host(lib, p, shape, dtype):
    a = np.empty(shape, dtype)
    _ffi.check(lib.cdkf_memcpy_d2h(a.ctypes.data_as(C.c_void_p), p, a.nbytes))
    return a


def _run_dev(lib, algo, blk, opts, t, y, dtype, layout, outputs=True):
    """Drive the *_dev entry points with explicit device buffers in the requested layout."""
    N, T, m = y.shape
    d = blk.state_dim
    suf = "f64" if dtype == np.float64 else "f32"
    opts.layout = layout
    tn = layout != _ffi.LAYOUT_NT
    tcn = layout == _ffi.LAYOUT_TCN
    th = np.ascontiguousarray((t.T if tn else t).astype(dtype))
    yh = np.ascontiguousarray((y.transpose(1, 2, 0) if tcn else y.transpose(1, 0, 2) if tn else y).astype(dtype))
    sz = np.dtype(dtype).itemsize
    td, yd = _dev(lib, th), _dev(lib, yh)
    ll, st, llsum = _dev(lib, nbytes=N * sz), _dev(lib, nbytes=N * 4), _dev(lib, nbytes=8)
    bufs = [_dev(lib, nbytes=N * T * w * sz) if outputs else None for w in (d, d * d, d, d * d)]
    fn = getattr(lib, f"cdkf_{algo}_{suf}_dev")
    _ffi.check(fn(C.byref(blk.c), C.byref(opts), N, T, td, yd, ll, *bufs, st, None))
    _ffi.check(getattr(lib, f"cdkf_ll_sum_{suf}_dev")(ll, N, llsum, None))
    _ffi.check(lib.cdkf_synchronize(None))
    shp = lambda w: (T,) + w + (N,) if tcn else ((T, N) + w if tn else (N, T) + w)
    outs = [None if b is None else _to_host(lib, b, shp(w), dtype) for b, w in zip(bufs, ((d,), (d, d), (d,), (d, d)))]
    if tcn:
        outs = [None if a is None else np.moveaxis(a, -1, 0) for a in outs]
    elif tn:
        outs = [None if a is None else np.swapaxes(a, 0, 1) for a in outs]
    res = (_to_host(lib, ll, (N,), dtype), outs, _to_host(lib, st, (N,), np.int32), _to_host(lib, llsum, (1,), np.float64)[0])
    for p in [td, yd, ll, st, llsum] + [b for b in bufs if b is not None]:
        lib.cdkf_free(p)
    return res


def test_layouts_are_bitwise_identical_and_ll_sum(hip_lib):
    rng = np.random.default_rng(7)
    mdl = o.lorenz63_model(3)
    N, T = 130, 50
    t = o.irregular_times(rng, N, T, 0.3)
    y = o.simulate(mdl, t, rng)
    blk = models._model_block(params_from(mdl))
    for algo in ("ekf_filter", "ukf_filter", "ekf_smoother"):
        for dtype in (np.float64, np.float32):
            a = _run_dev(hip_lib, algo, blk, _ffi.default_opts(), t, y, dtype, _ffi.LAYOUT_NT)
            for lay in (_ffi.LAYOUT_TN, _ffi.LAYOUT_TCN):
                b = _run_dev(hip_lib, algo, blk, _ffi.default_opts(), t, y, dtype, lay)
                np.testing.assert_array_equal(a[0], b[0])
                for x, z in zip(a[1], b[1]):
                    np.testing.assert_array_equal(x, z)
            assert abs(a[3] - a[0].astype(np.float64).sum()) <= 1e-12 * abs(a[3])
            assert (a[2] == 0).all()


@pytest.mark.parametrize("N,T", [(1100, 24), (5000, 12), (40000, 6)])
def test_results_do_not_depend_on_the_wavefront_grouping(hip_lib, N, T):
    """The launch groups 1 ... 64 distinct trajectories per wavefront depending on the batch size (reg_lanes_per_wave,
    cdkf_api.hip): a trajectory's results must be BITWISE the same inside a large batch (2, 8 and 64 per wavefront here)
    and in a batch of five (one per wavefront), for every lane-per-trajectory sweep."""
    rng = np.random.default_rng(N)
    mdl = o.lorenz63_model(3)
    t = o.irregular_times(rng, N, T, 0.012 * T)  # intervals on both sides of dt0: the RK pass count differs between lanes
    y = rng.standard_normal((N, T, 3)) * 3.0
    sub = np.sort(rng.choice(N, size=5, replace=False))
    P = params_from(mdl)
    for hyper in (cd.EKFHyperParams(), cd.UKFHyperParams()):
        big = cd.cdnlgssm_filter(P, y, t[..., None], hyperparams=hyper)
        small = cd.cdnlgssm_filter(P, y[sub], t[sub][..., None], hyperparams=hyper)
        # both sweeps of this model run on the sixteen-lanes-per-trajectory kernel up to 4096 trajectories (16 per CU)
        # (cdkf_lpe_kernels.h: the predict sums in a different order; the unscented filter there uses the collapsed moment
        # equations instead of forming the sigma points): bitwise on one side of that threshold, rounding-level across it
        same_kernel = N <= 4096
        for k in FILTER_KEYS + ["marginal_loglik"]:
            a, b = np.asarray(getattr(big, k))[sub], np.asarray(getattr(small, k))
            if same_kernel:
                np.testing.assert_array_equal(a, b, err_msg=k)
            else:
                np.testing.assert_allclose(a, b, rtol=1e-11, atol=1e-12, err_msg=k)
    big = cd.cdnlgssm_smoother(P, y, t[..., None])
    small = cd.cdnlgssm_smoother(P, y[sub], t[sub][..., None])
    for k in ("smoothed_means", "smoothed_covariances", "filtered_means"):
        a, b = np.asarray(getattr(big, k))[sub], np.asarray(getattr(small, k))
        if N <= 4096:  # the smoother's forward sweep follows the same kernel choice as the EKF filter above
            np.testing.assert_array_equal(a, b, err_msg=k)
        else:
            np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-12, err_msg=k)
    ll_b, g_b = cd.cdnlgssm_loglik_and_grad(P, y, t[..., None])
    ll_s, g_s = cd.cdnlgssm_loglik_and_grad(P, y[sub], t[sub][..., None])
    # (the gradient follows the same threshold: forward + reverse sweep on the sixteen-lane grid up to 4096 trajectories,
    # forward sensitivities with a lane per (trajectory, parameter) above)
    if N <= 4096:
        np.testing.assert_array_equal(ll_b[sub], ll_s)
    else:
        np.testing.assert_allclose(ll_b[sub], ll_s, rtol=1e-11)
    for name in ("sigma", "rho", "beta"):
        a, b = getattr(g_b, name)[sub], getattr(g_s, name)
        if N <= 4096:
            np.testing.assert_array_equal(a, b)
        else:
            np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-10 * np.abs(b).max())


def test_status_flags_and_nan_propagation(hip_lib):
    rng = np.random.default_rng(8)
    mdl = o.lorenz63_model(3)
    N, T = 3, 12
    t = o.irregular_times(rng, N, T, 0.1)
    y = o.simulate(mdl, t, rng)
    bad = o.Model(mdl.drift, mdl.L, mdl.Qc, mdl.H, mdl.bias, -10.0 * np.eye(3), mdl.m0, mdl.P0)  # S = P + R not PD
    blk = models._model_block(params_from(bad))
    ll, outs, st, _ = _run_dev(hip_lib, "ekf_filter", blk, _ffi.default_opts(), t, y, np.float64, _ffi.LAYOUT_NT)
    assert (st & _ffi.STATUS_NOT_PD).all() and np.isnan(ll).all()  # like the reference: silent NaN, here also flagged
    ref = o.ekf_filter(bad, t, y)
    assert np.isnan(ref["marginal_loglik"]).all()
    # max_steps cap: a 0.5 gap needs 50 steps
    blk = models._model_block(params_from(mdl))
    opts = _ffi.default_opts()
    opts.max_steps = 10
    tl = np.cumsum(np.full((1, 4), 0.5), axis=1)
    ll, outs, st, _ = _run_dev(hip_lib, "ekf_filter", blk, opts, tl, y[:1, :4], np.float64, _ffi.LAYOUT_NT)
    assert st[0] & _ffi.STATUS_MAX_STEPS


def test_c2_full_size_properties(hip_lib):
    """BASELINE config 2 (4096 x 1000, fp64) checked through size-independent properties:
    trajectories are independent (a random subset, re-run alone through the ORACLE, matches), the device-side
    log-likelihood sum equals the host sum, and no status flag is raised."""
    rng = np.random.default_rng(0)
    mdl = o.lorenz63_model(3)
    N, T = 4096, 1000
    t = o.irregular_times(rng, N, T, 0.005 * T)
    y = (rng.standard_normal((N, T, 3)) * 3.0)
    sub = rng.choice(N, size=6, replace=False)
    y[sub] = o.simulate(mdl, t[sub], rng)
    blk = models._model_block(params_from(mdl))
    ll, outs, st, llsum = _run_dev(hip_lib, "ekf_filter", blk, _ffi.default_opts(), t, y, np.float64, _ffi.LAYOUT_TCN)
    assert (st == 0).all() and np.isfinite(ll).all()
    assert abs(llsum - ll.sum()) < 1e-10 * abs(llsum)
    ref = o.ekf_filter(mdl, t[sub], y[sub])
    assert relerr(ll[sub], ref["marginal_loglik"]) < 1e-9
    for a, k in zip(outs, FILTER_KEYS):
        assert relerr(a[sub], ref[k]) < 1e-9, k
    # fp32 engine on the same batch stays within the north-star 1e-5 of the fp64 engine on the filtered moments
    ll32, outs32, st32, _ = _run_dev(hip_lib, "ekf_filter", blk, _ffi.default_opts(), t, y, np.float32, _ffi.LAYOUT_TCN)
    assert (st32 == 0).all()
    assert relerr(outs32[0][sub], ref["filtered_means"]) < 1e-5


def test_c2_full_size_gradient_properties(hip_lib):
    """The SGD objective at BASELINE config 2's size (4096 x 1000, fp64), value and gradient w.r.t. every leaf by the forward +
    reverse sweep on the sixteen-lane grid: trajectories are independent (a subset, re-run alone through the ORACLE's forward
    sensitivities / discrete adjoint, matches), the log-likelihood is the filter's, the drift block of the all-leaf call equals the
    drift-only call, and a trajectory's numbers do not depend on the batch it is in."""
    rng = np.random.default_rng(5)
    mdl = o.lorenz63_model(3)
    N, T = 4096, 1000
    t = o.irregular_times(rng, N, T, 0.005 * T)
    y = (rng.standard_normal((N, T, 3)) * 3.0)
    sub = np.sort(rng.choice(N, size=3, replace=False))
    y[sub] = o.simulate(mdl, t[sub], rng)
    P = params_from(mdl)
    hyp = cd.EKFHyperParams(state_order="first")
    ll, g = cd.cdnlgssm_loglik_and_grad(P, y, t[..., None], hyp)
    assert hip_lib.cdkf_last_kernel().startswith(b"grad_lpe_l63_kernel<double, 3, true, false>")
    gd = np.stack([g.sigma, g.rho, g.beta], -1)
    assert np.isfinite(gd).all() and np.isfinite(ll).all()
    post = cd.cdnlgssm_filter(P, y, t[..., None], hyp)
    np.testing.assert_allclose(ll, post.marginal_loglik, rtol=1e-12)
    ll_ref, g_ref = o.ekf_loglik_grad(mdl, t[sub], y[sub])
    np.testing.assert_allclose(ll[sub], ll_ref, rtol=1e-10)
    assert np.abs(gd[sub] - g_ref).max() < 1e-8 * np.abs(g_ref).max()   # (a thousand steps of a chaotic flow amplify rounding)
    ll2, ga = cd.cdnlgssm_loglik_and_grad_all(P, y, t[..., None], hyp)
    assert hip_lib.cdkf_last_kernel().startswith(b"grad_lpe_l63_kernel<double, 3, true, true>")
    np.testing.assert_array_equal(ll2, ll)
    gda = np.stack([ga.dynamics.drift.sigma, ga.dynamics.drift.rho, ga.dynamics.drift.beta], -1)
    assert np.abs(gda - gd).max() <= 1e-12 * np.abs(gd).max()
    _, _, ex = o.ekf_loglik_grad_adjoint(mdl, t[sub], y[sub], full=True)
    for name, got in (("m0", ga.initial.mean.params), ("P0", ga.initial.cov.params), ("R", ga.emissions.emission_cov.params),
                      ("H", ga.emissions.emission_function.weights), ("bias", ga.emissions.emission_function.bias),
                      ("Qc", ga.dynamics.diffusion_cov.params)):
        assert np.abs(np.asarray(got)[sub] - ex[name]).max() < 1e-8 * np.abs(ex[name]).max(), name
    # the same trajectories as a batch of their own: bitwise the same numbers (one kernel, four trajectories per wavefront either way)
    ll3, g3 = cd.cdnlgssm_loglik_and_grad_all(P, y[sub], t[sub][..., None], hyp)
    np.testing.assert_array_equal(ll3, ll[sub])
    np.testing.assert_array_equal(np.asarray(g3.emissions.emission_cov.params), np.asarray(ga.emissions.emission_cov.params)[sub])
    np.testing.assert_array_equal(np.asarray(g3.dynamics.drift.rho), np.asarray(ga.dynamics.drift.rho)[sub])


def test_unscented_filter_on_the_lane_grid(hip_lib, tmp_path):
    """Small Lorenz-63 batches with H = I run the unscented filter on filter_lpe_kernel<..., UKF = true>, whose moment
    equations are the sigma-point sums of inference_ukf.py:124-143 collapsed for this (quadratic) drift.  Against the oracle,
    which forms the sigma points literally: default and non-default (alpha, beta, kappa), odd / even T, N not a multiple of
    four, every output mode, long gaps, fp32; against the lane-per-trajectory kernel, which also forms them
    (CDKF_UKF_SIGMA_POINTS=1, read once per process: a child process); and the reference's NaN when the covariance the update
    draws its sigma points from is not positive definite."""
    import os, subprocess, sys
    rng = np.random.default_rng(21)
    mdl = o.lorenz63_model(3)
    P = params_from(mdl)
    for (N, T), hyp, kw in (((7, 31), cd.UKFHyperParams(), {}), ((6, 30), cd.UKFHyperParams(alpha=0.7, beta=1.5, kappa=0.5),
                                                                  dict(alpha=0.7, beta=1.5, kappa=0.5)),
                            ((1, 2), cd.UKFHyperParams(), {}), ((3, 1), cd.UKFHyperParams(), {})):
        t = o.irregular_times(rng, N, T, 0.008 * T)
        y = o.simulate(mdl, t, rng)
        ref = o.ukf_filter(mdl, t, y, **kw)
        _check_filter(cd.cdnlgssm_filter(P, y, t[..., None], hyp), ref, 1e-9)
        post = cd.cdnlgssm_filter(P, y, t[..., None], hyp, output_fields=[])
        assert relerr(post.marginal_loglik, ref["marginal_loglik"]) < 1e-9
        post = cd.cdnlgssm_filter(P, y, t[..., None], hyp, output_fields=["filtered_means", "filtered_covariances"])
        assert relerr(post.filtered_covariances, ref["filtered_covariances"]) < 1e-9
    assert hip_lib.cdkf_last_kernel().startswith(b"filter_lpe_kernel<double, cdkf::DriftLorenz63<double, 3>, 3, 2, true, true>")
    t = o.irregular_times(rng, 9, 25, 1.5)  # up to ~12 Dormand-Prince steps per interval
    y = o.simulate(mdl, t, rng)
    ref = o.ukf_filter(mdl, t, y)
    _check_filter(cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams()), ref, 1e-9)
    _check_filter(cd.cdnlgssm_filter(P, y.astype(np.float32), t[..., None], cd.UKFHyperParams()), ref, 1e-5)
    # the same call in a process that keeps the sigma-point kernel
    np.savez(tmp_path / "in.npz", t=t, y=y)
    code = ("import sys, numpy as np; sys.path[:0] = [%r, %r, %r]\n"
            "import cd_dynamax_amd as cd, cdkf_oracle as o\nfrom cd_dynamax_amd import _ffi\nfrom helpers import params_from\n"
            "d = np.load(%r); post = cd.cdnlgssm_filter(params_from(o.lorenz63_model(3)), d['y'], d['t'][..., None], cd.UKFHyperParams())\n"
            "assert _ffi.lib().cdkf_last_kernel().startswith(b'filter_reg_kernel'), _ffi.lib().cdkf_last_kernel()\n"
            "np.savez(%r, ll=post.marginal_loglik, fm=post.filtered_means, pP=post.predicted_covariances)\n"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(o.__file__)),
               os.path.dirname(os.path.abspath(__file__)), str(tmp_path / "in.npz"), str(tmp_path / "out.npz")))
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CDKF_UKF_SIGMA_POINTS="1"), check=True, timeout=600)
    other = np.load(tmp_path / "out.npz")
    post = cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams())
    assert relerr(post.marginal_loglik, other["ll"]) < 1e-11 and relerr(post.filtered_means, other["fm"]) < 1e-11
    assert relerr(post.predicted_covariances, other["pP"]) < 1e-11
    # an indefinite initial covariance: chol(P) of the first update's sigma points is NaN in the reference (S = P + R is fine)
    Pbad = np.array([[1.0, 2.0, 0.0], [2.0, 1.0, 0.0], [0.0, 0.0, 1.0]])
    bad = o.Model(mdl.drift, mdl.L, mdl.Qc, mdl.H, mdl.bias, 5.0 * np.eye(3), mdl.m0, Pbad)
    assert np.isnan(o.ukf_filter(bad, t[:2], y[:2])["marginal_loglik"]).all()
    opts = _ffi.default_opts()
    ll, outs, st, _ = _run_dev(hip_lib, "ukf_filter", models._model_block(params_from(bad)), opts, t[:2], y[:2], np.float64, _ffi.LAYOUT_TCN)
    assert np.isnan(ll).all() and (st & _ffi.STATUS_NOT_PD).all() and np.isnan(outs[0]).all()


def test_linear_drift_on_the_lane_grid(hip_lib):
    """The sixteen-lane sweep also carries linear drifts at state_dim 3 (constant per-lane coefficients): EKF orders, UKF (for a
    linear drift the unscented moment equations are the EKF's), full and partial observation, against the oracle and against
    the closed-form (matrix-exponential) Kalman filter; larger batches fall back to the lane-per-trajectory kernel and agree."""
    from helpers import closed_form_kf
    rng = np.random.default_rng(33)
    for m in (3, 1):
        mdl = linear_model(rng, 3, m)
        if m == 3:
            mdl = o.Model(mdl.drift, mdl.L, mdl.Qc, np.eye(3), np.zeros(3), mdl.R, mdl.m0, mdl.P0)  # H = I: the in-grid update
        P = params_from(mdl)
        N, T = 9, 40
        t = o.irregular_times(rng, N, T, 0.012 * T)
        y = o.simulate(mdl, t, rng)
        for order in ("first", "second"):
            _check_filter(cd.cdnlgssm_filter(P, y, t[..., None], cd.EKFHyperParams(state_order=order)), o.ekf_filter(mdl, t, y, state_order=order), 1e-9)
        assert hip_lib.cdkf_last_kernel().startswith(b"filter_lpe_kernel<double, cdkf::DriftLinear<double, 3>, %d, 1, " % m)
        if m == 3:
            _check_filter(cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams()), o.ukf_filter(mdl, t, y), 1e-9)
            assert hip_lib.cdkf_last_kernel().startswith(b"filter_lpe_kernel<double, cdkf::DriftLinear<double, 3>, 3, 1, true, true>")
        _check_filter(cd.cdnlgssm_filter(P, y.astype(np.float32), t[..., None]), o.ekf_filter(mdl, t, y), 1e-5)
        ex = closed_form_kf(mdl, t[0], y[0])
        post = cd.cdnlgssm_filter(P, y[0], t[0][:, None])
        assert relerr(post.filtered_means, ex["filtered_means"]) < 1e-7 and abs(post.marginal_loglik - ex["marginal_loglik"]) < 1e-6 * abs(ex["marginal_loglik"])
        big_t, big_y = np.tile(t, (920, 1))[:8200], np.tile(y, (920, 1, 1))[:8200]  # > 8192 (two wavefronts per SIMD): lane-per-trajectory kernel
        big = cd.cdnlgssm_filter(P, big_y, big_t[..., None])
        assert hip_lib.cdkf_last_kernel().startswith(b"filter_reg_kernel")
        small = cd.cdnlgssm_filter(P, y, t[..., None])
        assert relerr(np.asarray(big.filtered_covariances)[:N], small.filtered_covariances) < 1e-11


def test_unscented_filter_literal_sigma_points_flag(hip_lib):
    """UKFHyperParams(sigma_points=True) -> opts.flags & CDKF_FLAG_UKF_SIGMA_POINTS: the kernel that forms the sigma points and
    factorises the covariance in every Runge-Kutta stage (inference_ukf.py:57 called from :138), also where the closed form of
    the weighted sums would apply.  (a) same numbers as the closed form to rounding on a well-behaved problem, in one process;
    (b) the reference's failure path: with dt0 = 0.022 Lorenz-63 stage covariances lose positive definiteness INSIDE the first
    interval for two thirds of these trajectories -- the oracle (which restates the reference's per-stage cholesky) turns exactly
    those into NaN, and so does the flagged sweep; the default sweep tests positive definiteness at the observations only (DESIGN
    3.2b) and keeps some of them finite.  Unknown flag bits are refused."""
    rng = np.random.default_rng(33)
    mdl = o.lorenz63_model(3)
    P = params_from(mdl)
    t = o.irregular_times(rng, 9, 40, 0.4)
    y = o.simulate(mdl, t, rng)
    lit = cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams(sigma_points=True))
    assert hip_lib.cdkf_last_kernel().startswith(b"filter_reg_kernel"), hip_lib.cdkf_last_kernel()
    closed = cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams())
    assert hip_lib.cdkf_last_kernel().startswith(b"filter_lpe_kernel"), hip_lib.cdkf_last_kernel()
    for f in ("marginal_loglik", "filtered_means", "filtered_covariances", "predicted_means", "predicted_covariances"):
        assert relerr(getattr(lit, f), getattr(closed, f)) < 1e-11, f
    _check_filter(lit, o.ukf_filter(mdl, t, y), 1e-9)
    # (b)
    rng = np.random.default_rng(0)
    N, T = 32, 30
    t = o.irregular_times(rng, N, T, 0.06 * T)
    y = o.simulate(mdl, t, rng)
    with np.errstate(all="ignore"):
        ref = o.ukf_filter(mdl, t, y, dt0=0.022)
    bad = np.isnan(ref["marginal_loglik"])
    assert 8 <= bad.sum() <= N - 4, bad.sum()  # the regime this test is about: some trajectories fail, some do not
    hyp = cd.UKFHyperParams(diffeqsolve_settings={"dt0": 0.022}, sigma_points=True)
    post = cd.cdnlgssm_filter(P, y, t[..., None], hyp)
    np.testing.assert_array_equal(np.isnan(post.marginal_loglik), bad)
    np.testing.assert_array_equal(np.isnan(post.filtered_means), np.isnan(ref["filtered_means"]))
    assert relerr(post.marginal_loglik[~bad], ref["marginal_loglik"][~bad]) < 1e-9
    assert relerr(post.filtered_covariances[~bad], ref["filtered_covariances"][~bad]) < 1e-9
    default = cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams(diffeqsolve_settings={"dt0": 0.022}))
    assert np.isnan(default.marginal_loglik).sum() < bad.sum()  # the documented difference on the reference's failure path
    assert relerr(default.marginal_loglik[~bad], ref["marginal_loglik"][~bad]) < 1e-9
    # unknown bits
    blk = models._model_block(P)
    opts = _ffi.default_opts()
    opts.flags = 2
    assert hip_lib.cdkf_supported(_ffi.C.byref(blk.c), _ffi.C.byref(opts), 1, 8) in (0, 1)
    tt, yy = np.ascontiguousarray(t[:1]), np.ascontiguousarray(y[:1])
    ll, st = np.zeros(1), np.zeros(1, np.int32)
    dp = lambda a: a.ctypes.data_as(_ffi.C.c_void_p)
    rc = hip_lib.cdkf_ukf_filter_f64(_ffi.C.byref(blk.c), _ffi.C.byref(opts), 1, T, dp(tt), dp(yy), dp(ll), None, None, None, None, dp(st))
    assert rc == _ffi.CDKF_EINVAL and b"flags" in hip_lib.cdkf_last_error()


def test_unscented_loglik_gradient(hip_lib):
    """cdkf_ukf_loglik_grad_*: the unscented filter's marginal log-likelihood and its gradient w.r.t. the drift parameters --
    value_and_grad of the fit_sgd loss with filter_hyperparams=UKFHyperParams() (ssm_temissions.py:500, 555-568 -> models.py:393-408,
    708 -> inference_ukf.py:206-308) -- against the oracle's ukf_loglik_grad (pinned by finite differences of the literal sigma-point
    filter, tests/test_oracle.py): Lorenz-63 with m = 1, 2, 3 observed coordinates, default and non-default (alpha, beta, kappa),
    long gaps, fp32, another Runge-Kutta method; a linear drift (no curvature: the extended filter's gradient); fit_sgd's first
    step; the NaN of a covariance that is not positive definite; refusals."""
    from cd_dynamax_amd import fit
    from test_fit import _l63_problem
    rng = np.random.default_rng(61)
    for m in (3, 2, 1):
        mdl = o.lorenz63_model(m)
        P = params_from(mdl)
        N, T = 7, 30
        t = o.irregular_times(rng, N, T, 0.02 * T)
        y = o.simulate(mdl, t, rng)
        for hyp, kw in ((cd.UKFHyperParams(), {}), (cd.UKFHyperParams(alpha=0.7, beta=1.5, kappa=0.5), dict(alpha=0.7, beta=1.5, kappa=0.5))):
            ll_ref, g_ref = o.ukf_loglik_grad(mdl, t, y, **kw)
            ll, g = cd.cdnlgssm_loglik_and_grad(P, y, t[..., None], hyp)
            assert hip_lib.cdkf_last_kernel().startswith(b"ekf_grad_reg_kernel<double, 3, %d, false, true>" % m), hip_lib.cdkf_last_kernel()
            gd = np.stack([g.sigma, g.rho, g.beta], -1)
            np.testing.assert_allclose(ll, ll_ref, rtol=1e-10)
            assert np.abs(gd - g_ref).max() < 1e-9 * np.abs(g_ref).max()
            post = cd.cdnlgssm_filter(P, y, t[..., None], hyp, output_fields=[])
            np.testing.assert_allclose(ll, post.marginal_loglik, rtol=1e-10)      # the filter's own log-likelihood
        _, ge = cd.cdnlgssm_loglik_and_grad(P, y, t[..., None], cd.EKFHyperParams())
        assert np.abs(np.stack([ge.sigma, ge.rho, ge.beta], -1) - gd).max() > 1e-4 * np.abs(gd).max()  # not the extended filter's
    # fp32; another method with the run-time tableau
    ll32, g32 = cd.cdnlgssm_loglik_and_grad(P, y.astype(np.float32), t[..., None].astype(np.float32), cd.UKFHyperParams())
    assert ll32.dtype == np.float32 and np.abs(np.stack([g32.sigma, g32.rho, g32.beta], -1) - g_ref).max() < 5e-3 * np.abs(g_ref).max()
    with o.use_solver("tsit5"):
        ll_ref, g_ref = o.ukf_loglik_grad(mdl, t, y)
    ll, g = cd.cdnlgssm_loglik_and_grad(P, y, t[..., None], cd.UKFHyperParams(diffeqsolve_settings={"solver": "tsit5"}))
    assert np.abs(np.stack([g.sigma, g.rho, g.beta], -1) - g_ref).max() < 1e-9 * np.abs(g_ref).max()
    # a linear drift: the unscented moment equations are the extended filter's
    lin = linear_model(rng, 3, 3)
    Pl = params_from(lin)
    tl = o.irregular_times(rng, 4, 20, 0.5)
    yl = o.simulate(lin, tl, rng)
    ll_u, g_u = cd.cdnlgssm_loglik_and_grad(Pl, yl, tl[..., None], cd.UKFHyperParams())
    ll_r, g_r = o.ukf_loglik_grad(lin, tl, yl)
    np.testing.assert_allclose(ll_u, ll_r, rtol=1e-10)
    flat = np.concatenate([np.asarray(g_u.weights).reshape(4, -1), np.asarray(g_u.bias).reshape(4, -1)], -1)
    assert np.abs(flat - g_r).max() < 1e-9 * np.abs(g_r).max()
    # fit_sgd with the unscented filter: one plain-SGD step from the oracle's gradient
    model, params, props = _l63_problem(m=3)
    mdl3 = o.lorenz63_model(3)
    mdl3 = o.Model(mdl3.drift, mdl3.L, mdl3.Qc, mdl3.H, mdl3.bias, mdl3.R, np.zeros(3), 100 * np.eye(3))
    t3 = o.irregular_times(rng, 5, 25, 0.4)
    y3 = o.simulate(mdl3, t3, rng)
    new, losses = model.fit_sgd(params, props, y3, t3[..., None], cd.UKFHyperParams(), optimizer=fit.SGD(0.05), batch_size=5, num_epochs=1)
    ll_r, g_r = o.ukf_loglik_grad(mdl3, t3, y3)
    np.testing.assert_allclose(losses[0], -ll_r.sum() / y3.size, rtol=1e-10)
    np.testing.assert_allclose([new.dynamics.drift.sigma, new.dynamics.drift.rho, new.dynamics.drift.beta],
                               mdl3.drift.theta() + 0.05 * g_r.sum(0) / y3.size, rtol=1e-9)
    # an initial covariance that is not positive definite: NaN as in the filter (chol of the sigma points, inference_ukf.py:57)
    bad = o.Model(mdl3.drift, mdl3.L, mdl3.Qc, mdl3.H, mdl3.bias, mdl3.R, mdl3.m0, np.diag([1.0, -1.0, 1.0]))
    ll_b, _ = cd.cdnlgssm_loglik_and_grad(params_from(bad), y3[:2], t3[:2, :, None], cd.UKFHyperParams())
    assert np.isnan(ll_b).all()
    # refusals: no closed form for this drift (the MLP is not quadratic) and beyond the tangent sweep's sixteen dimensions (tests/test_ukf_tangent.py)
    from helpers import mlp_model
    with pytest.raises(NotImplementedError):
        cd.cdnlgssm_loglik_and_grad(params_from(mlp_model(rng, 20, 2, 8)), np.zeros((2, 5, 2)), np.arange(5.0)[None, :, None].repeat(2, 0), cd.UKFHyperParams())


@pytest.mark.parametrize("kind,d,m", [("lorenz63", 3, 1), ("lorenz63", 3, 2), ("lorenz63", 3, 3), ("lorenz96", 6, 3), ("lorenz96", 12, 5),
                                      ("linear", 4, 2), ("lorenz96", 20, 20)])
def test_unscented_loglik_gradient_of_every_leaf(hip_lib, kind, d, m):
    """value_and_grad of the UNSCENTED filter's marginal log-likelihood w.r.t. every leaf (what fit_sgd / fit_mcmc
    differentiate in the reference with filter_hyperparams=UKFHyperParams(): ssm_temissions.py:500-568, 601-679 -> inference_ukf.py:
    93-203) -- cdkf_ukf_loglik_grad_all_*: the reverse sweeps over the moment equations in closed form (exact for the quadratic
    Lorenz-63 / Lorenz-96 drifts and the linear one) -- against the oracle's ukf_loglik_grad_all, which tests/test_oracle.py pins by
    finite differences of the literal sigma-point filter.  Dense non-diagonal model matrices, an eight-step interval; the value is the
    sigma-point filter's own log-likelihood; fp32; then fit_sgd's first step over EVERY leaf and a short fit_mcmc with UKFHyperParams."""
    from cd_dynamax_amd import fit
    rng = np.random.default_rng(700 + 10 * d + m)
    if kind == "lorenz63":
        drift, scale = o.Lorenz63Drift(10.0, 28.0, 8.0 / 3.0), 1.0
    elif kind == "lorenz96":
        drift, scale = o.Lorenz96Drift(8.0), 8.0
    else:
        drift, scale = linear_model(rng, d, m).drift, 0.0
    A, B, Cm = rng.standard_normal((d, d)), rng.standard_normal((m, m)), rng.standard_normal((d, d))
    mdl = o.Model(drift, np.eye(d) + 0.1 * rng.standard_normal((d, d)), A @ A.T / d * 0.3 + 0.3 * np.eye(d), rng.standard_normal((m, d)) / np.sqrt(d),
                  0.1 * rng.standard_normal(m), B @ B.T / m * 0.5 + 0.3 * np.eye(m), scale + rng.standard_normal(d), Cm @ Cm.T / d * 0.5 + 0.5 * np.eye(d))
    N, T = 4, 10
    t = o.irregular_times(rng, N, T, 0.12)
    t[:, 6:] += 0.07
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    ll_ref, g_ref, ex = o.ukf_loglik_grad_all(mdl, t, y)
    ll, g = cd.cdnlgssm_loglik_and_grad_all(P, y, t[..., None], cd.UKFHyperParams())
    kern = _ffi.lib().cdkf_last_kernel().decode()
    assert kern.startswith("ekf_adjoint_wg_kernel<double" if d > 8 else "ekf_adjoint_wave8_kernel<double"), kern
    np.testing.assert_allclose(ll, ll_ref, rtol=1e-9)
    post = cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams(), output_fields=[])
    np.testing.assert_allclose(ll, post.marginal_loglik, rtol=1e-9)          # the sigma-point filter's own log-likelihood

    def close(a, b, name, tol=1e-8):
        sc = np.abs(b).max() + 1e-300
        assert np.abs(np.asarray(a) - b).max() < tol * sc, (name, np.abs(np.asarray(a) - b).max() / sc)

    flat = lambda gg: np.concatenate([np.asarray(a).reshape(N, -1) for a in gg.dynamics.drift], axis=-1)
    close(flat(g), g_ref, "drift")
    close(g.initial.mean.params, ex["m0"], "m0")
    close(g.initial.cov.params, ex["P0"], "P0")
    close(g.dynamics.diffusion_coefficient.params, ex["L"], "L")
    close(g.dynamics.diffusion_cov.params, ex["Qc"], "Qc")
    close(g.emissions.emission_function.weights, ex["H"], "H")
    close(g.emissions.emission_function.bias, ex["bias"], "bias")
    close(g.emissions.emission_cov.params, ex["R"], "R")
    if kind != "linear":  # not the extended filter's gradient: the curvature term is there
        _, ge = cd.cdnlgssm_loglik_and_grad_all(P, y, t[..., None], cd.EKFHyperParams(state_order="first"))
        assert np.abs(np.asarray(ge.initial.mean.params) - ex["m0"]).max() > 1e-5 * np.abs(ex["m0"]).max()
    # the drift block alone comes from the same sweeps where no forward-sensitivity kernel exists
    ll_d, g_d = cd.cdnlgssm_loglik_and_grad(P, y, t[..., None], cd.UKFHyperParams())
    close(np.concatenate([np.asarray(a).reshape(N, -1) for a in g_d], axis=-1), g_ref, "drift block", 1e-7)
    ll32, g32 = cd.cdnlgssm_loglik_and_grad_all(P, y.astype(np.float32), t[..., None].astype(np.float32), cd.UKFHyperParams())
    assert ll32.dtype == np.float32
    close(g32.emissions.emission_cov.params, ex["R"], "R fp32", 2e-2)
    if d > 8:
        return
    # fit_sgd over every leaf with the unscented objective: first plain-SGD step = the oracle's gradient pulled back
    free = cd.ParameterProperties()
    from cd_dynamax_amd.bijectors import RealToPSDBijector
    psd = cd.ParameterProperties(constrainer=RealToPSDBijector())
    frozen = cd.ParameterProperties(trainable=False)
    drift_props = type(P.dynamics.drift)(*([free] * len(P.dynamics.drift)))
    props = P._replace(
        initial=P.initial._replace(mean=cd.LearnableVector(free), cov=cd.LearnableMatrix(psd)),
        dynamics=P.dynamics._replace(drift=drift_props, diffusion_coefficient=cd.LearnableMatrix(frozen), diffusion_cov=cd.LearnableMatrix(psd),
                                     approx_order=frozen),
        emissions=P.emissions._replace(emission_function=cd.LearnableLinear(free, free), emission_cov=cd.LearnableMatrix(psd)))
    model = cd.ContDiscreteNonlinearGaussianSSM(d, m)
    lr = 1e-3
    new, losses = model.fit_sgd(P, props, y, t[..., None], cd.UKFHyperParams(), optimizer=fit.SGD(lr), batch_size=N, num_epochs=1)
    np.testing.assert_allclose(losses[0], -ll_ref.sum() / y.size, rtol=1e-9)
    np.testing.assert_allclose(np.asarray(new.initial.mean.params), mdl.m0 + lr * ex["m0"].sum(0) / y.size, rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(np.asarray(new.emissions.emission_function.weights), mdl.H + lr * ex["H"].sum(0) / y.size, rtol=1e-7, atol=1e-12)
    if kind == "lorenz63" and m == 2:
        out = model.fit_mcmc(P, props, y, t[..., None], cd.UKFHyperParams(), n_mcmc_samples=6,
                             mcmc_algorithm={"type": "hmc", "parameters": {"num_steps": 8, "num_integration_steps": 3}}, verbose=False, key=2)
        assert np.asarray(out[1].initial.mean.params).shape == (6, d) and np.all(np.isfinite(out[3]))


def test_c3_full_size_properties(hip_lib):
    """BASELINE config 3 (Lorenz-63 UKF, 4096 x 1000, fp32) through size-independent properties: a random subset re-run alone
    through the fp64 ORACLE (literal sigma points) matches the fp32 sweep -- filtered means within 1e-5 (the north-star bar; the
    reference itself computes in float32), log-likelihoods within 1e-5 --, no flag is raised, and the fp64 sweep of the same
    batch matches the oracle to 1e-9."""
    rng = np.random.default_rng(3)
    mdl = o.lorenz63_model(3)
    N, T = 4096, 1000
    t = o.irregular_times(rng, N, T, 0.005 * T)
    y = (rng.standard_normal((N, T, 3)) * 3.0)
    sub = rng.choice(N, size=5, replace=False)
    y[sub] = o.simulate(mdl, t[sub], rng)
    blk = models._model_block(params_from(mdl))
    ref = o.ukf_filter(mdl, t[sub], y[sub])
    ll, outs, st, llsum = _run_dev(hip_lib, "ukf_filter", blk, _ffi.default_opts(), t, y, np.float32, _ffi.LAYOUT_TCN)
    assert hip_lib.cdkf_last_kernel().startswith(b"filter_lpe_kernel<float, cdkf::DriftLorenz63<float, 3>, 3, 1, true, true>")
    assert (st == 0).all() and np.isfinite(ll).all()
    assert abs(llsum - ll.astype(np.float64).sum()) < 1e-6 * abs(llsum)
    assert relerr(ll[sub], ref["marginal_loglik"]) < 1e-5
    assert relerr(outs[0][sub], ref["filtered_means"]) < 1e-5
    assert relerr(outs[1][sub], ref["filtered_covariances"]) < 1e-5
    ll, outs, st, _ = _run_dev(hip_lib, "ukf_filter", blk, _ffi.default_opts(), t, y, np.float64, _ffi.LAYOUT_TCN)
    assert (st == 0).all()
    assert relerr(ll[sub], ref["marginal_loglik"]) < 1e-9
    for a, k in zip(outs, FILTER_KEYS):
        assert relerr(a[sub], ref[k]) < 1e-9, k


def test_reference_known_answer_constants_through_the_hip_kernels(hip_lib):
    """The reference's Dopri5 push-forward constants (src/test_scripts/cdlgssm_test_filter_TRegular.py:59-60) through
    the fp32 HIP EKF itself: F = -0.1 I, L = Qc = 0.5 I, unit interval, R huge so that the update is a no-op; the
    predicted mean / covariance after the first interval are A m0 and Q.  FMA contraction on the GPU changes
    individual roundings, so the bar is 2 ulp (the NumPy and C oracles reproduce the constants bit-exactly)."""
    mdl = o.Model(o.LinearDrift(-0.1 * np.eye(2), np.zeros(2)), 0.5 * np.eye(2), 0.5 * np.eye(2), np.eye(2),
                  np.zeros(2), 1e30 * np.eye(2), np.array([1.0, 0.0]), np.zeros((2, 2)))
    P = params_from(mdl)
    y = np.zeros((2, 2), np.float32)
    t = np.array([[0.0], [1.0]], np.float32)
    A_ref, Q_ref = np.float32(0.9048373699188232421875), np.float32(0.11329327523708343505859375)
    # (the UKF cannot start from P0 = 0: its sigma points need chol(P0))
    for hp in (cd.EKFHyperParams(dt_final=1.0, state_order="first"), cd.EKFHyperParams(dt_final=1.0)):
        post = cd.cdnlgssm_filter(P, y, t, hp)
        assert post.predicted_means.dtype == np.float32
        assert abs(float(post.predicted_means[0, 0]) - float(A_ref)) <= 2 * np.spacing(A_ref), type(hp).__name__
        assert abs(float(post.predicted_covariances[0, 0, 0]) - float(Q_ref)) <= 2 * np.spacing(Q_ref), type(hp).__name__
    post = cd.cdnlgssm_filter(P, y.astype(np.float64), t.astype(np.float64), cd.EKFHyperParams(dt_final=1.0))
    assert abs(post.predicted_means[0, 0] - np.exp(-0.1)) < 1e-14
    assert abs(post.predicted_covariances[0, 0, 0] - 0.125 * (1 - np.exp(-0.2)) / 0.2) < 1e-14


def test_linear_model_filters_agree_with_closed_form_kalman_filter(hip_lib):
    """The reference's own assertion (cdnlgssm_test_filter_linear_TRegular.py:314-324, 414-424: EKF first/second and UKF
    equal the CD Kalman filter on a linear model, rtol 1e-5), with the HIP kernels on one side and an exact
    matrix-exponential Kalman filter -- independent of the oracle -- on the other; fp32 run at the reference's
    precision, fp64 run far below its tolerance."""
    from helpers import closed_form_kf
    rng = np.random.default_rng(2026)
    mdl = linear_model(rng, 2, 6)  # the test script's STATE_DIM, EMISSION_DIM
    T = 100
    t = np.arange(T, dtype=float)
    y = o.simulate(mdl, t[None], rng)[0]
    ref = closed_form_kf(mdl, t, y, dt_final=1.0)
    P = params_from(mdl)
    for hp in (cd.EKFHyperParams(dt_final=1.0, state_order="first"), cd.EKFHyperParams(dt_final=1.0, state_order="second"),
               cd.UKFHyperParams(dt_final=1.0)):
        for dtype, tol in ((np.float64, 1e-7), (np.float32, 1e-4)):
            post = cd.cdnlgssm_filter(P, y.astype(dtype), t[:, None], hp)
            for k in FILTER_KEYS:
                assert relerr(getattr(post, k), ref[k]) < tol, (type(hp).__name__, dtype.__name__, k)
            assert abs(post.marginal_loglik - ref["marginal_loglik"]) < tol * abs(ref["marginal_loglik"]) * 10
    sm = cd.cdnlgssm_smoother(P, y, t[:, None], cd.EKFHyperParams(dt_final=1.0))
    np.testing.assert_array_equal(sm.smoothed_means[-1], sm.filtered_means[-1])


def test_baseline_config1_linear_tracking_front_end(hip_lib):
    """BASELINE.json config 1: CD linear-Gaussian tracking model (d_x=4, d_y=2), 1 trajectory, 500 regular steps, through
    the reference's linear-model surface (ContDiscreteLinearGaussianSSM.filter / .smoother), against the exact
    matrix-exponential Kalman filter and the oracle's smoother."""
    from helpers import closed_form_kf
    F = np.zeros((4, 4))
    F[0, 2] = F[1, 3] = 1.0
    H = np.eye(4)[:2]
    mdl = o.Model(o.LinearDrift(F, np.zeros(4)), np.eye(4), 0.1 * np.eye(4), H, np.zeros(2), 0.5 * np.eye(2),
                  np.array([8.0, 10.0, 1.0, 0.0]), np.eye(4))
    T = 500
    t = np.arange(T, dtype=float)
    rng = np.random.default_rng(1)
    y = o.simulate(mdl, t[None], rng)[0]
    model = cd.ContDiscreteLinearGaussianSSM(state_dim=4, emission_dim=2)
    pp = cd.ParameterProperties()
    params, _ = model.initialize(
        initial_mean={"params": mdl.m0, "props": pp}, initial_cov={"params": mdl.P0, "props": pp},
        dynamics_weights={"params": F, "props": pp}, dynamics_diffusion_coefficient={"params": mdl.L, "props": pp},
        dynamics_diffusion_cov={"params": mdl.Qc, "props": pp}, emission_weights={"params": H, "props": pp},
        emission_cov={"params": mdl.R, "props": pp})
    ref = closed_form_kf(mdl, t, y, dt_final=1.0)
    post = model.filter(params, y, filter_hyperparams=cd.KFHyperParams(dt_final=1.0))  # t_emissions=None: regular grid
    for k in FILTER_KEYS:
        assert relerr(getattr(post, k), ref[k]) < 1e-7, k
    assert abs(post.marginal_loglik - ref["marginal_loglik"]) < 1e-7 * abs(ref["marginal_loglik"])
    ll = model.marginal_log_prob(params, y, t[:, None], cd.KFHyperParams(dt_final=1.0))
    assert abs(ll - post.marginal_loglik) < 1e-10 * abs(ll)
    sm = model.smoother(params, y, t[:, None], smoother_type="cd_smoother_2")
    assert np.isfinite(sm.smoothed_covariances).all() and relerr(sm.smoothed_means[-1], post.filtered_means[-1]) < 1e-12
    # the NumPy oracle needs 100 Dormand-Prince steps per interval and sweep: pinned on the first 60 observations (the CPU
    # share of this test was 25 s, several minutes on a loaded GPU box)
    W = 60
    smw = model.smoother(params, y[:W], t[:W, None], smoother_type="cd_smoother_2")
    oref = o.ekf_smoother(mdl, t[None, :W], y[None, :W], state_order="first")
    assert relerr(smw.smoothed_means, oref["smoothed_means"][0]) < 1e-9
    assert relerr(smw.smoothed_covariances, oref["smoothed_covariances"][0]) < 1e-9
    post32 = model.filter(params, y.astype(np.float32), filter_hyperparams=cd.KFHyperParams(dt_final=1.0))
    assert relerr(post32.filtered_means, ref["filtered_means"]) < 1e-5


def test_forecast_matches_repeated_predict(hip_lib):
    """cdnlgssm_forecast (reference: models.py:767-936 -> forecast_extended/unscented_kalman_filter): the moments pushed
    from an initial Gaussian over a forecast grid without updates, EKF and UKF, register and wavefront kernels."""
    from helpers import lorenz96_model
    rng = np.random.default_rng(11)
    for mdl in (o.lorenz63_model(1), lorenz96_model(6, 3)):
        d = mdl.d
        A = rng.standard_normal((d, d))
        m_init, P_init = rng.standard_normal(d) + 2.0, A @ A.T / d + 0.5 * np.eye(d)
        t_init = 0.37
        t_forecast = t_init + np.cumsum(rng.uniform(0.001, 0.03, size=25))
        P = params_from(mdl)
        ref_m, ref_P = o.forecast(mdl, m_init, P_init, np.array([t_init]), t_forecast[None], method="ekf")
        fc = cd.cdnlgssm_forecast(P, (m_init, P_init), np.array([[t_init]]), t_forecast[:, None])
        assert isinstance(fc, cd.GSSMForecast) and fc.forecasted_state_means.shape == (25, d)
        assert relerr(fc.forecasted_state_means, ref_m[0]) < 1e-9
        assert relerr(fc.forecasted_state_covariances, ref_P[0]) < 1e-9
        fc = cd.cdnlgssm_forecast(P, (m_init, P_init), np.array([[t_init]]), t_forecast[:, None],
                                  output_fields=["forecasted_state_means"])
        assert fc.forecasted_state_covariances is None
        if d == 3:  # UKF forecast (register kernels only)
            ref_m, ref_P = o.forecast(mdl, m_init, P_init, np.array([t_init]), t_forecast[None], method="ukf")
            fc = cd.cdnlgssm_forecast(P, (m_init, P_init), np.array([[t_init]]), t_forecast[:, None], cd.UKFHyperParams())
            assert relerr(fc.forecasted_state_means, ref_m[0]) < 1e-8
            assert relerr(fc.forecasted_state_covariances, ref_P[0]) < 1e-8
    with pytest.raises(ValueError, match="t_forecast"):
        cd.cdnlgssm_forecast(P, (m_init, P_init), np.array([[t_init]]))


def test_emission_moments(hip_lib):
    """cdnlgssm_emissions (reference: models.py:939-1047 -> emissions_extended_kalman_filter, inference_ekf.py:768-855):
    (H m + b, H P H^T + R) for every state marginal, small and large dimensions, with and without covariances."""
    rng = np.random.default_rng(21)
    for d, m in ((3, 1), (4, 2), (40, 17)):
        mdl = linear_model(rng, d, m)
        P = params_from(mdl)
        mu = rng.standard_normal((5, 7, d))
        A = rng.standard_normal((5, 7, d, d))
        cov = A @ np.swapaxes(A, -1, -2)
        em, ec = cd.cdnlgssm_emissions(P, np.zeros((7, 1)), mu, cov)
        np.testing.assert_allclose(em, mu @ mdl.H.T + mdl.bias, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(ec, mdl.H @ cov @ mdl.H.T + mdl.R, rtol=1e-11, atol=1e-11)
        em2, ec2 = cd.cdnlgssm_emissions(P, np.zeros((7, 1)), mu[0])
        assert ec2 is None
        np.testing.assert_allclose(em2, em[0], rtol=1e-12, atol=1e-12)
        em32, ec32 = cd.cdnlgssm_emissions(P, np.zeros((7, 1)), mu.astype(np.float32), cov.astype(np.float32))
        assert em32.dtype == np.float32 and relerr(ec32, ec) < 1e-5


@pytest.mark.parametrize("kind,d,m", [("lorenz63", 3, 3), ("lorenz63", 3, 1), ("lorenz63", 3, 2), ("linear", 2, 2),
                                      ("linear", 2, 1), ("linear", 1, 1), ("linear", 3, 3)])
def test_loglik_gradient(hip_lib, kind, d, m):
    """cdnlgssm_loglik_and_grad (the drift block of jax.value_and_grad(_loss_fn), ssm_temissions.py:550-568) against the
    oracle's forward-sensitivity gradient (itself pinned to finite differences in tests/test_oracle.py), irregular
    per-trajectory times, N not a multiple of the wavefront, and the log-likelihood against the plain filter."""
    rng = np.random.default_rng(77)
    mdl = o.lorenz63_model(m) if kind == "lorenz63" else linear_model(rng, d, m)
    N, T = 70, 40
    t = o.irregular_times(rng, N, T, 0.2)
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    ll_ref, g_ref = o.ekf_loglik_grad(mdl, t, y)
    ll, g = cd.cdnlgssm_loglik_and_grad(P, y, t[..., None])
    gflat = np.stack([g.sigma, g.rho, g.beta], -1) if kind == "lorenz63" else np.concatenate(
        [g.weights.reshape(N, -1), g.bias], -1)
    assert gflat.shape == g_ref.shape
    np.testing.assert_allclose(ll, ll_ref, rtol=1e-11)
    scale = np.abs(g_ref).max(axis=0, keepdims=True) + 1e-30
    assert np.max(np.abs(gflat - g_ref) / scale) < 1e-9
    post = cd.cdnlgssm_filter(P, y, t[..., None], output_fields=[])
    np.testing.assert_allclose(ll, post.marginal_loglik, rtol=1e-13)
    # one trajectory, unbatched call
    ll1, g1 = cd.cdnlgssm_loglik_and_grad(P, y[3], t[3][:, None])
    assert np.ndim(ll1) == 0 and abs(ll1 - ll[3]) <= 1e-12 * abs(ll[3])
    # fp32 kernels: same recursion in single precision
    ll32, g32 = cd.cdnlgssm_loglik_and_grad(P, y.astype(np.float32), t[..., None].astype(np.float32))
    g32 = np.stack([g32.sigma, g32.rho, g32.beta], -1) if kind == "lorenz63" else np.concatenate(
        [g32.weights.reshape(N, -1), g32.bias], -1)
    assert g32.dtype == np.float32
    assert np.max(np.abs(g32 - g_ref) / scale) < 5e-3


def test_loglik_gradient_reverse_sweep_on_the_lane_grid(hip_lib, tmp_path):
    """Small Lorenz-63 batches with H = I take the gradient w.r.t. (sigma, rho, beta) from the forward sweep on the sixteen-lane
    grid plus grad_lpe_l63_kernel, the reverse sweep on the same grid (ssm_temissions.py:550-568: jax.value_and_grad is reverse
    mode too).  Against the oracle's forward-sensitivity gradient: one Runge-Kutta step per interval, several, more than the 64
    step starts the kernel parks in LDS, N not a multiple of four, T = 1 and 2, shared times, a dense symmetric R, another
    (L, Qc), fp32; and against the forward-sensitivity kernel (CDKF_NO_LPE_GRAD=1, read once per process: a child process)."""
    import os, subprocess, sys
    rng = np.random.default_rng(314)
    mdl = o.lorenz63_model(3)
    P = params_from(mdl)

    def flat(g):
        return np.stack([g.sigma, g.rho, g.beta], -1)

    def check(mdl_, t, y, tol=1e-9, dtype=np.float64):
        ll_ref, g_ref = o.ekf_loglik_grad(mdl_, t, y)
        ll, g = cd.cdnlgssm_loglik_and_grad(params_from(mdl_), y.astype(dtype), t[..., None].astype(dtype))
        assert hip_lib.cdkf_last_kernel().startswith(b"grad_lpe_l63_kernel<" + (b"double" if dtype == np.float64 else b"float"))
        scale = np.abs(g_ref).max(axis=0, keepdims=True) + 1e-30
        assert np.max(np.abs(ll - ll_ref) / np.abs(ll_ref)) < max(tol * 1e-2, 3e-6 if dtype == np.float32 else 0)
        assert np.max(np.abs(flat(g) - g_ref) / scale) < tol
        return ll, flat(g)

    # (total time spans: mean gaps of 0.006, 0.05 and 0.8 = one, about five and about eighty steps of dt0 = 0.01 per interval)
    for N, T, span in ((13, 50, 0.3), (6, 12, 0.6), (5, 4, 2.4), (3, 1, 0.01), (2, 2, 0.3), (1, 9, 0.2)):
        t = o.irregular_times(rng, N, T, span)
        check(mdl, t, o.simulate(mdl, t, rng))
    # intervals of about a thousand steps (dt0 = 1e-3 over gaps around 1): beyond 48 x 16 steps the coarse step starts are spaced wider
    # than the window, and a segment is walked back in several refills of the window (two-level checkpoints, cdkf_lpe_grad_kernels.h)
    tl = o.irregular_times(rng, 3, 3, 2.0)
    assert np.diff(tl, axis=1).max() > 0.8
    yl = o.simulate(mdl, tl, rng)
    ll_ref, g_ref = o.ekf_loglik_grad(mdl, tl, yl, dt0=1e-3)
    ll_l, g_l = cd.cdnlgssm_loglik_and_grad(P, yl, tl[..., None], cd.EKFHyperParams(diffeqsolve_settings={"dt0": 1e-3}))
    assert hip_lib.cdkf_last_kernel().startswith(b"grad_lpe_l63_kernel<double")
    np.testing.assert_allclose(ll_l, ll_ref, rtol=1e-10)
    assert np.max(np.abs(flat(g_l) - g_ref) / (np.abs(g_ref).max(axis=0, keepdims=True) + 1e-30)) < 1e-8
    t = o.irregular_times(rng, 9, 30, 0.9)
    assert np.diff(t, axis=1).max() > 0.05
    y = o.simulate(mdl, t, rng)
    check(mdl, t, y, tol=5e-3, dtype=np.float32)
    # two observations at one instant: no predict between them, in either sweep
    t2 = t.copy()
    t2[:, 6] = t2[:, 5]
    check(mdl, t2, y)
    # one time grid shared by the batch
    ts = t[0]
    ll_s, g_s = cd.cdnlgssm_loglik_and_grad(P, y, ts[:, None])
    assert hip_lib.cdkf_last_kernel().startswith(b"grad_lpe_l63_kernel<double")
    _, g_ref = o.ekf_loglik_grad(mdl, np.broadcast_to(ts, t.shape), y)
    assert np.max(np.abs(flat(g_s) - g_ref) / (np.abs(g_ref).max(axis=0, keepdims=True) + 1e-30)) < 1e-9
    # dense symmetric R, non-trivial diffusion, off-centre prior
    A = rng.standard_normal((3, 3))
    R2 = 0.3 * np.eye(3) + 0.1 * A @ A.T
    R2 = 0.5 * (R2 + R2.T)  # bitwise symmetric (the in-grid update's condition; anything else takes the per-lane kernels)
    mdl2 = o.Model(mdl.drift, np.eye(3) + 0.2 * rng.standard_normal((3, 3)), np.diag([0.5, 1.5, 1.0]), mdl.H, mdl.bias,
                   R2, np.array([1.0, -2.0, 20.0]), 2.0 * np.eye(3) + 0.3 * A.T @ A)
    check(mdl2, t, o.simulate(mdl2, t, rng))
    # the same call in a process that keeps the forward-sensitivity kernel
    ll, g = check(mdl, t, y)
    # (and with the step count capped below what the longer intervals need: both sweeps stop there, the flag is the forward sweep's)
    capped = cd.EKFHyperParams(diffeqsolve_settings={"max_steps": 1})
    ll_c, g_c = cd.cdnlgssm_loglik_and_grad(P, y, t[..., None], capped)
    assert hip_lib.cdkf_last_kernel().startswith(b"grad_lpe_l63_kernel<double")
    np.savez(tmp_path / "in.npz", t=t, y=y)
    code = ("import sys, numpy as np; sys.path[:0] = [%r, %r, %r]\n"
            "import cd_dynamax_amd as cd, cdkf_oracle as o\nfrom cd_dynamax_amd import _ffi\nfrom helpers import params_from\n"
            "d = np.load(%r); ll, g = cd.cdnlgssm_loglik_and_grad(params_from(o.lorenz63_model(3)), d['y'], d['t'][..., None])\n"
            "assert _ffi.lib().cdkf_last_kernel().startswith(b'ekf_grad_reg_kernel'), _ffi.lib().cdkf_last_kernel()\n"
            "llc, gc = cd.cdnlgssm_loglik_and_grad(params_from(o.lorenz63_model(3)), d['y'], d['t'][..., None],\n"
            "                                      cd.EKFHyperParams(diffeqsolve_settings={'max_steps': 1}))\n"
            "np.savez(%r, ll=ll, g=np.stack([g.sigma, g.rho, g.beta], -1), llc=llc, gc=np.stack([gc.sigma, gc.rho, gc.beta], -1))\n"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(o.__file__)),
               os.path.dirname(os.path.abspath(__file__)), str(tmp_path / "in.npz"), str(tmp_path / "out.npz")))
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CDKF_NO_LPE_GRAD="1"), check=True, timeout=600)
    other = np.load(tmp_path / "out.npz")
    assert relerr(ll, other["ll"]) < 1e-12
    assert np.max(np.abs(g - other["g"]) / np.abs(other["g"]).max(axis=0, keepdims=True)) < 1e-10
    assert relerr(ll_c, other["llc"]) < 1e-12 and relerr(ll_c, ll) > 1e-6
    assert np.max(np.abs(flat(g_c) - other["gc"]) / np.abs(other["gc"]).max(axis=0, keepdims=True)) < 1e-10


def test_loglik_gradient_unsupported_raises(hip_lib):
    rng = np.random.default_rng(5)
    mdl = lorenz96_model(48, 4)           # state_dim 48 in fp64: beyond the LDS plan of the workgroup reverse sweep (and every other)
    t = o.irregular_times(rng, 2, 5, 0.1)
    y = o.simulate(mdl, t, rng)
    with pytest.raises(NotImplementedError, match="LDS plan"):   # (refused by the launch: the host gate is precision-agnostic)
        cd.cdnlgssm_loglik_and_grad(params_from(mdl), y, t[..., None])
    with pytest.raises(NotImplementedError, match="no gradient kernel"):
        from helpers import mlp_model                                    # an MLP drift with a hidden layer beyond 64: no reverse sweep
        cd.cdnlgssm_loglik_and_grad(params_from(mlp_model(rng, 20, 4, (80, 8))), np.zeros((2, 5, 4)), t[..., None])   # (d <= 16: the tangent sweep)
    with pytest.raises(NotImplementedError, match="no gradient kernel"):
        cd.cdnlgssm_loglik_and_grad(params_from(o.lorenz63_model(3)), y[..., :3], t[..., None],
                                    cd.EKFHyperParams(state_order="zeroth"))


@pytest.mark.parametrize("d,m", [(4, 2), (2, 6), (8, 3), (1, 1)])
def test_linear_smoother_type1(hip_lib, d, m):
    """cdlgssm_smoother(..., smoother_type='cd_smoother_1') -- the reference's default (inference.py:694-823): discrete RTS
    on the Dopri5-pushed-forward (A, Q), with smoothed_cross_covariances.  Against the oracle restatement (itself pinned to
    the exact matrix-exponential RTS smoother in tests/test_oracle.py); batched irregular times, then one trajectory on
    the regular grid (t_emissions=None: 100 Dormand-Prince steps per interval)."""
    rng = np.random.default_rng(40 + d)
    base = linear_model(rng, d, m)
    mdl = o.Model(o.LinearDrift(base.drift.W, np.zeros(d)), base.L, base.Qc, base.H, base.bias, base.R, base.m0, base.P0)
    N, T = 5, 25
    t = o.irregular_times(rng, N, T, 0.15)
    y = o.simulate(mdl, t, rng)
    model = cd.ContDiscreteLinearGaussianSSM(state_dim=d, emission_dim=m, has_emissions_bias=True)
    pp = cd.ParameterProperties()
    params, _ = model.initialize(
        initial_mean={"params": mdl.m0, "props": pp}, initial_cov={"params": mdl.P0, "props": pp},
        dynamics_weights={"params": mdl.drift.W, "props": pp}, dynamics_diffusion_coefficient={"params": mdl.L, "props": pp},
        dynamics_diffusion_cov={"params": mdl.Qc, "props": pp}, emission_weights={"params": mdl.H, "props": pp},
        emission_bias={"params": mdl.bias, "props": pp}, emission_cov={"params": mdl.R, "props": pp})
    ref = o.kf_smoother_type1(mdl, t, y)
    post = model.smoother(params, y, t[..., None])          # default smoother_type
    assert post.smoothed_cross_covariances.shape == (N, T - 1, d, d)
    for k in ("filtered_means", "filtered_covariances", "smoothed_means", "smoothed_covariances", "smoothed_cross_covariances"):
        assert relerr(getattr(post, k), ref[k]) < 1e-9, k
    np.testing.assert_allclose(post.marginal_loglik, ref["marginal_loglik"], rtol=1e-11)
    # type 2 (the reference freezes the filtered moments over each interval) approximates the same smoother
    post2 = model.smoother(params, y, t[..., None], smoother_type="cd_smoother_2")
    assert relerr(post.smoothed_means, post2.smoothed_means) < 5e-2
    # one trajectory, regular grid, single precision
    y1 = y[0]
    ref1 = o.kf_smoother_type1(mdl, np.arange(T, dtype=float)[None], y1[None], dt_final=1.0)
    p1 = model.smoother(params, y1, filter_hyperparams=cd.KFHyperParams(dt_final=1.0))
    assert p1.smoothed_means.shape == (T, d) and relerr(p1.smoothed_covariances, ref1["smoothed_covariances"][0]) < 1e-8
    p32 = model.smoother(params, y1.astype(np.float32), filter_hyperparams=cd.KFHyperParams(dt_final=1.0))
    assert p32.smoothed_means.dtype == np.float32 and relerr(p32.smoothed_means, ref1["smoothed_means"][0]) < 2e-3


@pytest.mark.parametrize("d,m", [(3, 2), (4, 2), (6, 3)])
def test_linear_front_end_dynamics_bias_and_inputs(hip_lib, d, m):
    """ContDiscreteLinearGaussianSSM.filter / marginal_log_prob with a dynamics bias, dynamics and emission input weights and an
    input series: the reference adds B u_k + b to the pushed-forward mean WITHOUT integrating it and D u_k + d to the emission mean
    (inference.py:185-205, 596-620).  Against the oracle running exactly that recursion (kf_filter_inputs); batched irregular
    times with per-trajectory inputs, then one trajectory on the regular grid (the last predict over dt_final = 1) with a shared
    input series, single precision, and the shapes that stay refused."""
    rng = np.random.default_rng(80 + d)
    base = linear_model(rng, d, m)
    mdl = o.Model(o.LinearDrift(base.drift.W, np.zeros(d)), base.L, base.Qc, base.H, base.bias, base.R, base.m0, base.P0)
    nu = 2
    b, B, D = 0.3 * rng.standard_normal(d), rng.standard_normal((d, nu)), rng.standard_normal((m, nu))
    N, T = 5, 14
    t = o.irregular_times(rng, N, T, 0.2)
    u = rng.standard_normal((N, T, nu))
    y = o.simulate(mdl, t, rng) + u @ D.T
    model = cd.ContDiscreteLinearGaussianSSM(state_dim=d, emission_dim=m, input_dim=nu, has_dynamics_bias=True, has_emissions_bias=True)
    pp = cd.ParameterProperties()
    params, _ = model.initialize(
        initial_mean={"params": mdl.m0, "props": pp}, initial_cov={"params": mdl.P0, "props": pp},
        dynamics_weights={"params": mdl.drift.W, "props": pp}, dynamics_bias={"params": b, "props": pp},
        dynamics_input_weights={"params": B, "props": pp}, dynamics_diffusion_coefficient={"params": mdl.L, "props": pp},
        dynamics_diffusion_cov={"params": mdl.Qc, "props": pp}, emission_weights={"params": mdl.H, "props": pp},
        emission_bias={"params": mdl.bias, "props": pp}, emission_input_weights={"params": D, "props": pp},
        emission_cov={"params": mdl.R, "props": pp})
    ref = o.kf_filter_inputs(mdl, t, y, b, B, D, u)
    post = model.filter(params, y, t[..., None], inputs=u)
    for k in FILTER_KEYS:
        assert relerr(getattr(post, k), ref[k]) < 1e-9, k
    np.testing.assert_allclose(post.marginal_loglik, ref["marginal_loglik"], rtol=1e-10)
    np.testing.assert_allclose(model.marginal_log_prob(params, y, t[..., None], inputs=u), ref["marginal_loglik"], rtol=1e-10)
    # the offsets matter: without them the means are elsewhere
    plain = o.ekf_filter(mdl, t, y, state_order="first")
    assert relerr(plain["filtered_means"], ref["filtered_means"]) > 1e-3
    # bias only, one trajectory, regular grid, shared (unbatched) inputs, fp32
    y1, u1 = y[0], u[0]
    ref1 = o.kf_filter_inputs(mdl, np.arange(T, dtype=float)[None], y1[None], b, B, D, u1[None], dt_final=1.0)
    p1 = model.filter(params, y1, inputs=u1, filter_hyperparams=cd.KFHyperParams(dt_final=1.0))
    assert p1.filtered_means.shape == (T, d) and relerr(p1.predicted_means, ref1["predicted_means"][0]) < 1e-9
    p32 = model.filter(params, y1.astype(np.float32), inputs=u1, filter_hyperparams=cd.KFHyperParams(dt_final=1.0))
    assert p32.filtered_means.dtype == np.float32 and relerr(p32.filtered_means, ref1["filtered_means"][0]) < 2e-3
    with pytest.raises(NotImplementedError, match="smoothers take no inputs|_predict"):
        model.smoother(params, y1, inputs=u1)


@pytest.mark.parametrize("d,m", [(4, 2), (3, 3), (8, 5)])
def test_linear_front_end_trains_dynamics_bias_and_input_weights(hip_lib, d, m):
    """VERDICT r3 "missing" 5: the reference's linear model makes the dynamics bias and the input weights ordinary trainable leaves
    (continuous_discrete_linear_gaussian_ssm/models.py:116-139, 167) although its predict adds B u_k + b un-integrated
    (inference.py:185-205).  marginal_log_prob_and_grad with a bias and inputs: EVERY leaf against central finite differences of the
    oracle's recursion (kf_filter_inputs) -- b, B, D through the per-step cotangents of the mean jumps / observations
    (cdkf_ekf_loglik_grad_jumps_*), the other leaves through the same reverse sweep in the presence of the jumps; then fit_sgd over
    (b, B, D, emission bias) from a perturbed start lowers the loss and its first step is the finite-difference gradient's."""
    rng = np.random.default_rng(300 + d)
    base = linear_model(rng, d, m)
    mdl = o.Model(o.LinearDrift(base.drift.W, np.zeros(d)), base.L, base.Qc, base.H, base.bias, base.R, base.m0, base.P0)
    nu = 2
    b, B, D = 0.3 * rng.standard_normal(d), 0.5 * rng.standard_normal((d, nu)), 0.5 * rng.standard_normal((m, nu))
    N, T = 3, 9
    t = o.irregular_times(rng, N, T, 0.25)
    u = rng.standard_normal((N, T, nu))
    y = o.simulate(mdl, t, rng) + u @ D.T
    model = cd.ContDiscreteLinearGaussianSSM(state_dim=d, emission_dim=m, input_dim=nu, has_dynamics_bias=True, has_emissions_bias=True)
    pp = cd.ParameterProperties()
    fz = cd.ParameterProperties(trainable=False)

    def make(b_, B_, D_, W_=mdl.drift.W, H_=mdl.H, hb_=mdl.bias, m0_=mdl.m0, props=None):
        pr = props or {}
        g = lambda k: pr.get(k, fz)
        return model.initialize(
            initial_mean={"params": m0_, "props": g("m0")}, initial_cov={"params": mdl.P0, "props": fz},
            dynamics_weights={"params": W_, "props": g("W")}, dynamics_bias={"params": b_, "props": g("b")},
            dynamics_input_weights={"params": B_, "props": g("B")}, dynamics_diffusion_coefficient={"params": mdl.L, "props": fz},
            dynamics_diffusion_cov={"params": mdl.Qc, "props": fz}, emission_weights={"params": H_, "props": g("H")},
            emission_bias={"params": hb_, "props": g("hb")}, emission_input_weights={"params": D_, "props": g("D")},
            emission_cov={"params": mdl.R, "props": fz})

    params, _ = make(b, B, D)
    ll, g = model.marginal_log_prob_and_grad(params, y, t[..., None], inputs=u)
    ref = o.kf_filter_inputs(mdl, t, y, b, B, D, u)
    np.testing.assert_allclose(ll, ref["marginal_loglik"], rtol=1e-10)

    def total(b_=b, B_=B, D_=D, W_=None, H_=None, hb_=None, m0_=None):
        mm = o.Model(o.LinearDrift(mdl.drift.W if W_ is None else W_, np.zeros(d)), mdl.L, mdl.Qc, mdl.H if H_ is None else H_,
                     mdl.bias if hb_ is None else hb_, mdl.R, mdl.m0 if m0_ is None else m0_, mdl.P0)
        return o.kf_filter_inputs(mm, t, y, b_, B_, D_, u)["marginal_loglik"]

    def fd(arr, key, h=1e-6):
        out = np.zeros((N,) + arr.shape)
        for idx in np.ndindex(arr.shape):
            ap, am = arr.copy(), arr.copy()
            ap[idx] += h
            am[idx] -= h
            out[(slice(None),) + idx] = (total(**{key: ap}) - total(**{key: am})) / (2 * h)
        return out

    checks = [("dynamics.bias", g.dynamics.bias, fd(b, "b_")), ("dynamics.input_weights", g.dynamics.input_weights, fd(B, "B_")),
              ("emissions.input_weights", g.emissions.input_weights, fd(D, "D_")), ("dynamics.weights", g.dynamics.weights, fd(mdl.drift.W, "W_")),
              ("emissions.weights", g.emissions.weights, fd(mdl.H, "H_")), ("emissions.bias", g.emissions.bias, fd(mdl.bias, "hb_")),
              ("initial.mean", g.initial.mean, fd(mdl.m0, "m0_"))]
    for name, got, want in checks:
        assert np.asarray(got).shape == want.shape, name
        assert np.abs(np.asarray(got) - want).max() < 2e-6 * max(1.0, np.abs(want).max()), (name, np.abs(np.asarray(got) - want).max(), np.abs(want).max())
    # one trajectory, unbatched, bias only (no inputs given): leaves without a leading axis
    ll1, g1 = model.marginal_log_prob_and_grad(make(b, np.zeros((d, nu)), np.zeros((m, nu)))[0], y[0], t[0][:, None])
    assert np.ndim(ll1) == 0 and g1.dynamics.bias.shape == (d,)
    ref1 = o.kf_filter_inputs(mdl, t[:1], y[:1], b, np.zeros((d, nu)), np.zeros((m, nu)), np.zeros((1, T, nu)))
    np.testing.assert_allclose(ll1, ref1["marginal_loglik"][0], rtol=1e-10)
    # fit_sgd over the offsets' leaves from a perturbed start
    tr = cd.ParameterProperties()
    p0, props = make(b + 0.3, B * 0.5, D * 0.5, props={"b": tr, "B": tr, "D": tr, "hb": tr})
    from cd_dynamax_amd.fit import SGD
    lr = 1e-2
    fitted, losses = model.fit_sgd(p0, props, y, t[..., None], inputs=u, optimizer=SGD(lr), batch_size=N, num_epochs=1)
    _, g0 = model.marginal_log_prob_and_grad(p0, y, t[..., None], inputs=u)
    step_b = lr * np.asarray(g0.dynamics.bias).sum(0) / y.size     # loss = -sum ll / emissions.size: one SGD step adds lr * grad ll / size
    np.testing.assert_allclose(np.asarray(fitted.dynamics.bias) - (b + 0.3), step_b, rtol=1e-8, atol=1e-12)
    fitted2, losses2 = model.fit_sgd(p0, props, y, t[..., None], inputs=u, optimizer=cd.fit.Adam(2e-2),
                                     batch_size=2, num_epochs=40, shuffle=True, key=1)
    assert losses2[-1] < losses2[0] - 1e-3, (losses2[0], losses2[-1])
    # what stays refused says so: a non-default solver with offsets
    with pytest.raises(NotImplementedError, match="default solver|Dopri5|dopri5"):
        model.marginal_log_prob_and_grad(params, y, t[..., None], inputs=u, filter_hyperparams=cd.KFHyperParams(diffeqsolve_settings={"solver": "tsit5"}))


@pytest.mark.parametrize("d,m,solver,ctrl", [(3, 2, "tsit5", None), (6, 3, "bosh3", None), (3, 2, "dopri5", dict(rtol=1e-6, atol=1e-8)),
                                             (6, 3, "tsit5", dict(rtol=1e-5, atol=1e-7))])
def test_linear_smoother_type1_solver_settings(hip_lib, d, m, solver, ctrl):
    """cd_smoother_1 with diffeqsolve_settings: the pushed-forward (A, Q) integrate with the chosen method, with fixed steps or
    under diffrax.PIDController (error norm over the (A, Q) pytree), the forward pass likewise -- against the oracle under use_solver."""
    rng = np.random.default_rng(70 + d)
    base = linear_model(rng, d, m)
    mdl = o.Model(o.LinearDrift(base.drift.W, np.zeros(d)), base.L, base.Qc, base.H, base.bias, base.R, base.m0, base.P0)
    N, T = 4, 12
    t = o.irregular_times(rng, N, T, 0.3)
    t[:, 6:] += 0.2
    y = o.simulate(mdl, t, rng)
    model = cd.ContDiscreteLinearGaussianSSM(state_dim=d, emission_dim=m, has_emissions_bias=True)
    pp = cd.ParameterProperties()
    params, _ = model.initialize(
        initial_mean={"params": mdl.m0, "props": pp}, initial_cov={"params": mdl.P0, "props": pp},
        dynamics_weights={"params": mdl.drift.W, "props": pp}, dynamics_diffusion_coefficient={"params": mdl.L, "props": pp},
        dynamics_diffusion_cov={"params": mdl.Qc, "props": pp}, emission_weights={"params": mdl.H, "props": pp},
        emission_bias={"params": mdl.bias, "props": pp}, emission_cov={"params": mdl.R, "props": pp})
    settings = {"solver": solver, "dt0": 0.05}
    if ctrl is not None:
        settings["stepsize_controller"] = cd.PIDController(**ctrl)
    with o.use_solver(solver, adaptive=ctrl):
        ref = o.kf_smoother_type1(mdl, t, y, dt0=0.05)
    post = model.smoother(params, y, t[..., None], filter_hyperparams=cd.KFHyperParams(diffeqsolve_settings=settings))
    for k in ("filtered_covariances", "smoothed_means", "smoothed_covariances", "smoothed_cross_covariances"):
        assert relerr(getattr(post, k), ref[k]) < 1e-9, k
    np.testing.assert_allclose(post.marginal_loglik, ref["marginal_loglik"], rtol=1e-10)


def test_linear_smoother_type1_refusals(hip_lib):
    rng = np.random.default_rng(3)
    mdl = linear_model(rng, 12, 3)
    model = cd.ContDiscreteLinearGaussianSSM(state_dim=12, emission_dim=3)
    pp = cd.ParameterProperties()
    params, _ = model.initialize(dynamics_weights={"params": mdl.drift.W, "props": pp})
    y = rng.standard_normal((6, 3))
    with pytest.raises(NotImplementedError, match="state_dim <= 8"):
        model.smoother(params, y)
    assert model.smoother(params, y, smoother_type="cd_smoother_2").smoothed_means.shape == (6, 12)
    with pytest.raises(ValueError, match="unknown smoother_type"):
        model.smoother(params, y, smoother_type="cd_smoother_3")


@pytest.mark.parametrize("solver", ["tsit5", "bosh3", "heun", "midpoint", "ralston", "euler"])
def test_runge_kutta_solver_choice(hip_lib, solver):
    """diffeqsolve_settings={'solver': ...} (the reference forwards a diffrax solver object to dfx.diffeqsolve,
    src/utils/diffrax_utils.py:40-57, 150-163; fixed steps of dt0): EKF second / zeroth order, UKF, EKF smoother and
    forecast on Lorenz-63, EKF on a linear model, against the oracle integrating with the same tableau."""
    rng = np.random.default_rng(90)
    settings = {"solver": solver, "dt0": 0.01}
    for mdl, name in ((o.lorenz63_model(2), "l63"), (linear_model(rng, 2, 2), "lin")):
        N, T = 9, 25
        t = o.irregular_times(rng, N, T, 0.04)
        y = o.simulate(mdl, t, rng)
        P = params_from(mdl)
        with o.use_solver(solver):
            ref = o.ekf_filter(mdl, t, y)
            ref0 = o.ekf_filter(mdl, t, y, state_order="zeroth")
            refu = o.ukf_filter(mdl, t, y)
            refs = o.ekf_smoother(mdl, t, y)
        post = cd.cdnlgssm_filter(P, y, t[..., None], cd.EKFHyperParams(diffeqsolve_settings=settings))
        for k in FILTER_KEYS:
            assert relerr(getattr(post, k), ref[k]) < 1e-11, (name, k)
        np.testing.assert_allclose(post.marginal_loglik, ref["marginal_loglik"], rtol=1e-11)
        post0 = cd.cdnlgssm_filter(P, y, t[..., None], cd.EKFHyperParams(state_order="zeroth", diffeqsolve_settings=settings))
        assert relerr(post0.filtered_means, ref0["filtered_means"]) < 1e-11
        postu = cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams(diffeqsolve_settings=settings))
        assert relerr(postu.filtered_covariances, refu["filtered_covariances"]) < 1e-10
        sm = cd.cdnlgssm_smoother(P, y, t[..., None], cd.EKFHyperParams(diffeqsolve_settings=settings))
        assert relerr(sm.smoothed_means, refs["smoothed_means"]) < 1e-10
        assert relerr(sm.smoothed_covariances, refs["smoothed_covariances"]) < 1e-10
        if name == "l63":  # the method matters (this is not the default tableau under another name)
            dflt = cd.cdnlgssm_filter(P, y, t[..., None], cd.EKFHyperParams())
            diff = relerr(post.filtered_covariances, dflt.filtered_covariances)
            assert 1e-14 < diff < 1e-6 if solver == "tsit5" else diff > 1e-10, diff
    p32 = cd.cdnlgssm_filter(P, y.astype(np.float32), t[..., None].astype(np.float32), cd.EKFHyperParams(diffeqsolve_settings=settings))
    assert relerr(p32.filtered_means, ref["filtered_means"]) < 1e-3
    fc = cd.cdnlgssm_forecast(P, (mdl.m0, mdl.P0), np.array([[0.0]]), np.linspace(0.05, 0.5, 5)[:, None],
                              cd.EKFHyperParams(diffeqsolve_settings=settings))
    with o.use_solver(solver):
        rm, rP = o.forecast(mdl, mdl.m0[None], mdl.P0[None], np.array([0.0]), np.linspace(0.05, 0.5, 5)[None], "ekf")
    assert relerr(fc.forecasted_state_means, rm[0]) < 1e-11


def test_solver_choice_refusals(hip_lib):
    rng = np.random.default_rng(1)
    mdl = lorenz96_model(8, 4)
    t = o.irregular_times(rng, 2, 5, 0.1)
    y = o.simulate(mdl, t, rng)
    with pytest.raises(NotImplementedError, match="choose from"):
        cd.cdnlgssm_filter(params_from(mdl), y, t[..., None], cd.EKFHyperParams(diffeqsolve_settings={"solver": "kvaerno5"}))
    with pytest.raises(NotImplementedError, match="stepsize_controller"):
        cd.cdnlgssm_filter(params_from(mdl), y, t[..., None], cd.EKFHyperParams(diffeqsolve_settings={"stepsize_controller": object()}))
    l63 = o.lorenz63_model(3)
    y3 = o.simulate(l63, t, rng)
    with o.use_solver("heun"):                                                  # the all-parameter sweep: any of the methods
        ll_ref, g_ref = o.ekf_loglik_grad_adjoint(l63, t, y3)
    ll, g = cd.cdnlgssm_loglik_and_grad_all(params_from(l63), y3, t[..., None], cd.EKFHyperParams(diffeqsolve_settings={"solver": "heun"}))
    np.testing.assert_allclose(ll, ll_ref, rtol=1e-10)
    flat = np.concatenate([np.asarray(a).reshape(2, -1) for a in g.dynamics.drift], axis=-1)
    assert np.abs(flat - g_ref).max() < 1e-8 * np.abs(g_ref).max()


@pytest.mark.parametrize("solver,ctrl", [("tsit5", dict(rtol=1e-6, atol=1e-8)), ("dopri5", dict(rtol=1e-5, atol=1e-7, pcoeff=0.1, icoeff=0.3)),
                                         ("bosh3", dict(rtol=1e-4, atol=1e-6)), ("heun", dict(rtol=1e-3, atol=1e-5, dcoeff=0.05))])
def test_adaptive_step_size_control(hip_lib, solver, ctrl):
    """diffeqsolve_settings={'solver': ..., 'stepsize_controller': PIDController(rtol, atol, ...)} (the reference's tutorial
    uses Tsit5 + PIDController for its high-fidelity log-likelihood): every trajectory adapts on its own.  Against the oracle's
    restatement of the controller -- identical accept / reject decisions give agreement to rounding; intervals of very
    different lengths, first step dt0 = 0.05 far above what the tolerance allows (forces rejections)."""
    rng = np.random.default_rng(100)
    mdl = o.lorenz63_model(2)
    N, T = 9, 20
    t = o.irregular_times(rng, N, T, 0.05)
    t[:, 10:] += 0.4
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    settings = {"solver": solver, "dt0": 0.05, "stepsize_controller": cd.PIDController(**ctrl)}
    with o.use_solver(solver, adaptive=ctrl):
        ref = o.ekf_filter(mdl, t, y, dt0=0.05)
        refu = o.ukf_filter(mdl, t, y, dt0=0.05)
        refs = o.ekf_smoother(mdl, t, y, dt0=0.05)
        ref0 = o.ekf_filter(mdl, t, y, dt0=0.05, state_order="zeroth")
    post = cd.cdnlgssm_filter(P, y, t[..., None], cd.EKFHyperParams(diffeqsolve_settings=settings))
    for k in FILTER_KEYS:
        assert relerr(getattr(post, k), ref[k]) < 1e-10, k
    np.testing.assert_allclose(post.marginal_loglik, ref["marginal_loglik"], rtol=1e-10)
    postu = cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams(diffeqsolve_settings=settings))
    assert relerr(postu.filtered_means, refu["filtered_means"]) < 1e-9
    sm = cd.cdnlgssm_smoother(P, y, t[..., None], cd.EKFHyperParams(diffeqsolve_settings=settings))
    assert relerr(sm.smoothed_covariances, refs["smoothed_covariances"]) < 1e-9
    post0 = cd.cdnlgssm_filter(P, y, t[..., None], cd.EKFHyperParams(state_order="zeroth", diffeqsolve_settings=settings))
    assert relerr(post0.filtered_means, ref0["filtered_means"]) < 1e-10
    # the adaptive solve tracks the fixed-step reference solution to about its tolerance
    fine = cd.cdnlgssm_filter(P, y, t[..., None], cd.EKFHyperParams(diffeqsolve_settings={"dt0": 0.001}))
    assert relerr(post.filtered_means, fine.filtered_means) < 300 * ctrl["rtol"]
    # a step budget that cannot be met raises the MAX_STEPS status, as diffrax raises on max_steps
    tight = dict(settings, max_steps=3)
    mb = models._model_block(P)
    op = models._opts(cd.EKFHyperParams(diffeqsolve_settings=tight), 1)
    _, _, status = _ffi.run_host("ekf_filter", mb, op, t, y, [False] * 4, np.float64)
    assert (status & 4).any()


@pytest.mark.parametrize("ctrl", [dict(rtol=1e-6, atol=1e-8, dtmax=0.004), dict(rtol=1e-12, atol=1e-14, dtmin=0.003),
                                  dict(rtol=1e-5, atol=1e-7, pcoeff=0.1, icoeff=0.3, dtmin=0.002, dtmax=0.01),
                                  dict(rtol=1e-6, atol=1e-8, safety=0.7, factormin=0.5, factormax=1.5),      # (ABI 110: the controller's clip)
                                  dict(rtol=1e-7, atol=1e-9, pcoeff=0.2, icoeff=0.4, safety=0.95, factormin=0.1, factormax=3.0, dtmax=0.02)])
def test_adaptive_step_size_bounds(hip_lib, ctrl):
    """PIDController(dtmin=, dtmax=) and (round 5, ABI 110) PIDController(safety=, factormin=, factormax=) -- the
    reference forwards any controller, src/utils/diffrax_utils.py:40-57: the
    three kernel families that adapt -- register-resident (Lorenz-63), workgroup (Lorenz-96 d = 12), the type-1 smoother's pushed-forward
    (A, Q) of the linear front-end -- against the oracle's controller with the same bounds: a cap that binds on every step, a floor that
    forces steps the tolerance would reject (kept by force_dtmin), both at once."""
    rng = np.random.default_rng(101)
    settings = {"solver": "dopri5", "dt0": 0.05, "stepsize_controller": cd.PIDController(**ctrl)}
    for mdl, (N, T) in ((o.lorenz63_model(2), (7, 12)), (lorenz96_model(12, 5), (3, 6))):
        t = o.irregular_times(rng, N, T, 0.05)
        t[:, T // 2:] += 0.1
        y = o.simulate(mdl, t, rng)
        P = params_from(mdl)
        with o.use_solver("dopri5", adaptive=ctrl):
            ref = o.ekf_filter(mdl, t, y, dt0=0.05, state_order="first")
            refs = o.ekf_smoother(mdl, t, y, dt0=0.05, state_order="first")
        post = cd.cdnlgssm_filter(P, y, t[..., None], cd.EKFHyperParams(state_order="first", diffeqsolve_settings=settings))
        for k in FILTER_KEYS:
            assert relerr(getattr(post, k), ref[k]) < 1e-9, (mdl.d, k)
        np.testing.assert_allclose(post.marginal_loglik, ref["marginal_loglik"], rtol=1e-9)
        sm = cd.cdnlgssm_smoother(P, y, t[..., None], cd.EKFHyperParams(state_order="first", diffeqsolve_settings=settings))
        assert relerr(sm.smoothed_covariances, refs["smoothed_covariances"]) < 1e-8, mdl.d
    # the bounds really bind: without them the same tolerances give other numbers
    free = {k: v for k, v in ctrl.items() if k not in ("dtmin", "dtmax", "safety", "factormin", "factormax")}
    postf = cd.cdnlgssm_filter(P, y, t[..., None], cd.EKFHyperParams(state_order="first", diffeqsolve_settings=dict(
        settings, stepsize_controller=cd.PIDController(**free), max_steps=400)))
    assert not np.array_equal(np.nan_to_num(postf.filtered_means), post.filtered_means)
    # linear front-end, smoother type 1: the (A, Q) pairs of every interval integrate under the same controller
    d, m = 4, 2
    base = linear_model(rng, d, m)
    lm = o.Model(o.LinearDrift(base.drift.W, np.zeros(d)), base.L, base.Qc, base.H, base.bias, base.R, base.m0, base.P0)
    Nl, Tl = 3, 10
    tl = o.irregular_times(rng, Nl, Tl, 0.15)
    tl[:, 6:] += 0.2
    yl = o.simulate(lm, tl, rng)
    model = cd.ContDiscreteLinearGaussianSSM(state_dim=d, emission_dim=m, has_emissions_bias=True)
    pp = cd.ParameterProperties()
    lp, _ = model.initialize(
        initial_mean={"params": lm.m0, "props": pp}, initial_cov={"params": lm.P0, "props": pp},
        dynamics_weights={"params": lm.drift.W, "props": pp}, dynamics_diffusion_coefficient={"params": lm.L, "props": pp},
        dynamics_diffusion_cov={"params": lm.Qc, "props": pp}, emission_weights={"params": lm.H, "props": pp},
        emission_bias={"params": lm.bias, "props": pp}, emission_cov={"params": lm.R, "props": pp})
    with o.use_solver("dopri5", adaptive=ctrl):
        ref1 = o.kf_smoother_type1(lm, tl, yl, dt0=0.05)
    sm1 = model.smoother(lp, yl, tl[..., None], filter_hyperparams=cd.KFHyperParams(diffeqsolve_settings=settings))
    for k in ("smoothed_means", "smoothed_covariances", "smoothed_cross_covariances"):
        assert relerr(getattr(sm1, k), ref1[k]) < 1e-9, k
    # and the library refuses bounds that make no sense
    op = models._opts(cd.EKFHyperParams(diffeqsolve_settings=dict(settings, stepsize_controller=cd.PIDController(1e-3, 1e-6, dtmin=0.1, dtmax=0.01))), 1)
    with pytest.raises(_ffi.CdkfError, match="dtmin"):
        _ffi.run_host("ekf_filter", models._model_block(P), op, t, y, [False] * 4, np.float64)


def test_notebook_pin_default_vs_tsit5_pid_loglik(hip_lib):
    """The reference-recorded statement about the adaptive path (tutorial diffeqsolve_settings_analysis.ipynb:385-386; see
    tests/test_oracle.py::test_notebook_pin_default_vs_tsit5_pid_loglik): default Dopri5 and Tsit5 + PIDController(1e-9, 1e-9)
    give the same marginal log-likelihood to float32 resolution.  The HIP path on the notebook's model at its time density:
    fp64 within 6.7e-8 relative (one float32 ulp of the notebook's value); fp32, the reference's precision, within 16 ulps
    (the notebook saw 0 at 10 000 steps; rounding accumulates differently here) with the adaptive solve under a tolerance
    BELOW float32 resolution still terminating; each against the oracle under the same settings."""
    rng = np.random.default_rng(2024)
    mdl = o.lorenz63_model(1)
    P = params_from(mdl)
    N, T = 6, 2000
    t = o.irregular_times(rng, N, T, 0.005 * T)
    y = o.simulate(mdl, t, rng)
    hifi_settings = {"solver": "tsit5", "stepsize_controller": cd.PIDController(rtol=1e-9, atol=1e-9), "max_steps": 10 ** 7}
    d64 = cd.cdnlgssm_filter(P, y, t[..., None], cd.EKFHyperParams(), output_fields=[]).marginal_loglik
    h64 = cd.cdnlgssm_filter(P, y, t[..., None], cd.EKFHyperParams(diffeqsolve_settings=hifi_settings), output_fields=[]).marginal_loglik
    assert np.all(np.abs(d64 - h64) <= 6.7e-8 * np.abs(h64)), (d64, h64)
    y32 = y.astype(np.float32)
    d32 = np.asarray(cd.cdnlgssm_filter(P, y32, t[..., None], cd.EKFHyperParams(), output_fields=[]).marginal_loglik, np.float64)
    h32 = np.asarray(cd.cdnlgssm_filter(P, y32, t[..., None], cd.EKFHyperParams(diffeqsolve_settings=hifi_settings), output_fields=[]).marginal_loglik,
                     np.float64)
    assert np.all(np.isfinite(h32))
    assert np.all(np.abs(d32 - h32) <= 16 * 6.0e-8 * np.abs(h64)), (d32, h32)
    assert np.all(np.abs(d32 - d64) <= 3e-5 * np.abs(d64))
    # the oracle under the same two settings (a short prefix: the adaptive NumPy loop is slow)
    Ts = 150
    with o.use_solver("tsit5", adaptive=dict(rtol=1e-9, atol=1e-9)):
        ref_h = o.ekf_filter(mdl, t[:2, :Ts], y[:2, :Ts])["marginal_loglik"]
    got_h = cd.cdnlgssm_filter(P, y[:2, :Ts], t[:2, :Ts, None], cd.EKFHyperParams(diffeqsolve_settings=hifi_settings), output_fields=[]).marginal_loglik
    assert relerr(got_h, ref_h) < 1e-9


def test_notebook_pins_lower_fidelity_settings_by_magnitude(hip_lib):
    """The same notebook's other recorded log-likelihoods (tests/test_oracle.py::test_notebook_pins_lower_fidelity_settings_by_magnitude:
    Tsit5 + PIDController(1e-3, 1e-6) from dt0 = 0.1 two float32 ulps off the high-fidelity value, Heun dt0 = 1e-3 off by 2.5e-2, Euler
    dt0 = 1e-4 by 5.3e-1, per 1e4 observations) through the HIP path: on the same seeded problem the fp64 sweeps reproduce the
    oracle's sums to 1e-9 under each of the four settings -- so they sit at the same magnitudes -- and in fp32, the reference's
    precision, the loose controller stays within 16 ulps per sequence of the high-fidelity solve, as the notebook saw (2)."""
    from test_oracle import _notebook_problem
    mdl, t, y = _notebook_problem()
    P = params_from(mdl)
    settings = {
        "hifi": ({"solver": "tsit5", "stepsize_controller": cd.PIDController(rtol=1e-9, atol=1e-9), "max_steps": 10 ** 7},
                 dict(solver="tsit5", adaptive=dict(rtol=1e-9, atol=1e-9)), {}),
        "loose": ({"solver": "tsit5", "dt0": 0.1, "stepsize_controller": cd.PIDController(rtol=1e-3, atol=1e-6), "max_steps": 100},
                  dict(solver="tsit5", adaptive=dict(rtol=1e-3, atol=1e-6)), dict(dt0=0.1, max_steps=100)),
        "heun": ({"solver": "heun", "dt0": 1e-3, "max_steps": 10 ** 4}, dict(solver="heun"), dict(dt0=1e-3, max_steps=10 ** 4)),
        "euler": ({"solver": "euler", "dt0": 1e-4, "max_steps": 10 ** 3}, dict(solver="euler"), dict(dt0=1e-4, max_steps=10 ** 3)),
    }
    got, got32 = {}, {}
    for name, (hip, orc, kw) in settings.items():
        ll = cd.cdnlgssm_filter(P, y, t[..., None], cd.EKFHyperParams(diffeqsolve_settings=hip), output_fields=[]).marginal_loglik
        with o.use_solver(orc["solver"], **({"adaptive": orc["adaptive"]} if "adaptive" in orc else {})):
            ref = o.ekf_filter(mdl, t, y, **kw)["marginal_loglik"]
        assert relerr(ll, ref) < 1e-9, name
        got[name] = float(np.sum(ll))
        if name in ("hifi", "loose"):
            got32[name] = np.asarray(cd.cdnlgssm_filter(P, y.astype(np.float32), t[..., None], cd.EKFHyperParams(diffeqsolve_settings=hip),
                                                        output_fields=[]).marginal_loglik, np.float64)
    ulp32 = float(np.spacing(np.float32(abs(got["hifi"]))))
    assert abs(got["loose"] - got["hifi"]) < 3 * ulp32
    assert 2.54e-2 / 5 < abs(got["heun"] - got["hifi"]) < 2.54e-2 * 5
    assert 5.3e-1 / 5 < abs(got["euler"] - got["hifi"]) < 5.3e-1 * 5
    per_seq_ulp = np.spacing(np.abs(got32["hifi"]).astype(np.float32)).astype(np.float64)
    assert np.all(np.abs(got32["loose"] - got32["hifi"]) <= 16 * per_seq_ulp), (got32["loose"] - got32["hifi"]) / per_seq_ulp


def test_adaptive_refusals(hip_lib):
    rng = np.random.default_rng(1)
    l63 = o.lorenz63_model(3)
    t = o.irregular_times(rng, 2, 5, 0.1)
    y = o.simulate(l63, t, rng)
    pid = cd.PIDController(1e-3, 1e-6)
    with pytest.raises(NotImplementedError, match="embedded error estimate"):
        cd.cdnlgssm_filter(params_from(l63), y, t[..., None], cd.EKFHyperParams(diffeqsolve_settings={"solver": "euler", "stepsize_controller": pid}))
    mdl = lorenz96_model(8, 4)   # beyond the register-resident shapes: the workgroup kernels adapt too (tests/test_gpu_wg.py)
    y8 = o.simulate(mdl, t, rng)
    with o.use_solver("dopri5", adaptive=dict(rtol=1e-3, atol=1e-6)):
        ref8 = o.ekf_filter(mdl, t, y8)
    post8 = cd.cdnlgssm_filter(params_from(mdl), y8, t[..., None], cd.EKFHyperParams(diffeqsolve_settings={"stepsize_controller": pid}))
    assert relerr(post8.filtered_covariances, ref8["filtered_covariances"]) < 1e-9
    assert cd.cdnlgssm_filter(params_from(l63), y, t[..., None], cd.EKFHyperParams(
        diffeqsolve_settings={"stepsize_controller": cd.ConstantStepSize(), "tol_vbt": 1e-5})).filtered_means.shape == (2, 5, 3)


@pytest.mark.parametrize("settings,ctx", [({"solver": "tsit5"}, ("tsit5", None)), ({"solver": "heun", "dt0": 0.002}, ("heun", None)),
                                          ({"solver": "tsit5", "dt0": 0.05, "stepsize_controller": ("pid", 1e-6, 1e-8)},
                                           ("tsit5", dict(rtol=1e-6, atol=1e-8)))])
def test_loglik_gradient_under_solver_settings(hip_lib, settings, ctx):
    """The reference's tutorial differentiates the log-likelihood under Tsit5 + PIDController and under cheap fixed-step
    methods (notebooks/tutorial/diffeqsolve_settings_analysis.ipynb).  Forward sensitivities ride on the primal's steps --
    with adaptive control the error norm sees the primal state only and the controller's factor carries no derivative, which
    is how JAX differentiates the solve -- against the oracle doing the same."""
    rng = np.random.default_rng(33)
    mdl = o.lorenz63_model(1)
    N, T = 20, 15
    t = o.irregular_times(rng, N, T, 0.06)
    y = o.simulate(mdl, t, rng)
    st = dict(settings)
    if "stepsize_controller" in st:
        st["stepsize_controller"] = cd.PIDController(*st["stepsize_controller"][1:])
    dt0 = st.get("dt0", 0.01)
    with o.use_solver(ctx[0], adaptive=ctx[1]):
        ll_ref, g_ref = o.ekf_loglik_grad(mdl, t, y, dt0=dt0)
    ll, g = cd.cdnlgssm_loglik_and_grad(params_from(mdl), y, t[..., None], cd.EKFHyperParams(diffeqsolve_settings=st))
    np.testing.assert_allclose(ll, ll_ref, rtol=1e-10)
    got = np.stack([g.sigma, g.rho, g.beta], -1)
    assert relerr(got, g_ref) < 1e-8


def test_integration_md_binding_stub_runs(hip_lib):
    """The ctypes stub INTEGRATION.md shows a maintainer (section 2) is executed as written -- only the library path and the
    posterior class are supplied -- and must agree with the package's own filter: the documented struct layouts and call
    sequence are the real ones."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    block = re.search(r"```python\nimport ctypes as C, numpy as np\n(.*?)```", text, re.S).group(0)
    code = block.strip("`").replace("python\n", "", 1)
    code = code.replace('C.CDLL("libcdkf_hip.so")', f'C.CDLL({_ffi.LIB_PATH!r})')
    ns = {"PosteriorGSSMFiltered": cd.PosteriorGSSMFiltered}
    exec(code, ns)
    ns["_lib"].cdkf_last_error.restype = C.c_char_p
    rng = np.random.default_rng(5)
    mdl = o.lorenz63_model(2)
    T = 30
    t = o.irregular_times(rng, 1, T, 0.3)
    y = o.simulate(mdl, t, rng)[0]
    P = params_from(mdl)
    hyp = cd.EKFHyperParams()
    got = ns["_hip_ekf_filter"](P, y, t[0][:, None], hyp, 1)
    ref = cd.cdnlgssm_filter(P, y, t[0][:, None], hyp)
    for k in FILTER_KEYS:
        assert relerr(getattr(got, k), getattr(ref, k)) < 1e-13, k
    assert abs(got.marginal_loglik - ref.marginal_loglik) < 1e-12 * abs(ref.marginal_loglik)


def test_device_resident_torch_tensors(hip_lib):
    """emissions / t_emissions as torch tensors that already live on the GPU: the sweeps run on them in place (the `_dev` entry
    points on torch's current stream) and return device tensors equal to the host path's arrays -- filter, smoother,
    log-likelihood, gradients; fp64 and fp32; register and wavefront kernels; batched and single trajectory."""
    import torch
    rng = np.random.default_rng(55)
    for mdl, order in ((o.lorenz63_model(2), "second"), (lorenz96_model(6, 3), "first")):
        N, T = 37, 12
        t = o.irregular_times(rng, N, T, 0.05)
        y = o.simulate(mdl, t, rng)
        P = params_from(mdl)
        hyp = cd.EKFHyperParams(state_order=order)
        yd, td = torch.from_numpy(y).cuda(), torch.from_numpy(t[..., None]).cuda()
        host = cd.cdnlgssm_filter(P, y, t[..., None], hyp)
        dev = cd.cdnlgssm_filter(P, yd, td, hyp)
        assert dev.filtered_means.is_cuda and tuple(dev.filtered_covariances.shape) == host.filtered_covariances.shape
        for k in list(FILTER_KEYS) + ["marginal_loglik"]:
            np.testing.assert_array_equal(getattr(dev, k).cpu().numpy(), getattr(host, k))
        sm_h = cd.cdnlgssm_smoother(P, y, t[..., None], hyp)
        sm_d = cd.cdnlgssm_smoother(P, yd, td, hyp)
        np.testing.assert_array_equal(sm_d.smoothed_covariances.cpu().numpy(), sm_h.smoothed_covariances)
        ll_d = cd.ContDiscreteNonlinearGaussianSSM(mdl.d, mdl.m).marginal_log_prob(P, yd, td, cd.UKFHyperParams())
        ll_h = cd.ContDiscreteNonlinearGaussianSSM(mdl.d, mdl.m).marginal_log_prob(P, y, t[..., None], cd.UKFHyperParams())
        np.testing.assert_array_equal(ll_d.cpu().numpy(), ll_h)
        one = cd.cdnlgssm_filter(P, yd[3], td[3], hyp)                       # single trajectory
        assert tuple(one.filtered_means.shape) == (T, mdl.d) and one.marginal_loglik.ndim == 0
        np.testing.assert_array_equal(one.filtered_means.cpu().numpy(), host.filtered_means[3])
        f32 = cd.cdnlgssm_filter(P, yd.float(), td.float(), hyp)
        assert f32.filtered_means.dtype == torch.float32 and relerr(f32.filtered_means.cpu().numpy(), host.filtered_means) < 1e-3
        gh = cd.cdnlgssm_loglik_and_grad(P, y, t[..., None], cd.EKFHyperParams(state_order="first"))
        gd = cd.cdnlgssm_loglik_and_grad(P, yd, td, cd.EKFHyperParams(state_order="first"))
        for a, b in zip(gd[1], gh[1]):
            np.testing.assert_array_equal(a.cpu().numpy(), b)
        ga_h = cd.cdnlgssm_loglik_and_grad_all(P, y, t[..., None], cd.EKFHyperParams(state_order="first"))
        ga_d = cd.cdnlgssm_loglik_and_grad_all(P, yd, td, cd.EKFHyperParams(state_order="first"))
        np.testing.assert_array_equal(ga_d[1].emissions.emission_cov.params, ga_h[1].emissions.emission_cov.params)
