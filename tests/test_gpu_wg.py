"""Parity of the workgroup-per-trajectory kernels (state dimensions beyond the register kernels: Lorenz-96,
MLP drift, larger linear models) with the oracle.  GPU only."""
import os
import numpy as np
import pytest

import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi
import cdkf_oracle as o
from helpers import (FILTER_KEYS, GOLDEN_WIDE, linear_model, load_golden, lorenz96_model, mlp_model, model_from_fixture,
                     params_from, relerr)

pytestmark = pytest.mark.gpu


def _check(post, ref, tol, keys=FILTER_KEYS):
    assert relerr(post.marginal_loglik, ref["marginal_loglik"]) < tol
    for k in keys:
        assert relerr(getattr(post, k), ref[k]) < tol, k


@pytest.mark.parametrize("d,m", [(6, 3), (12, 12), (40, 40)])
def test_lorenz96_filter_and_smoother(hip_lib, d, m):
    rng = np.random.default_rng(d)
    mdl = lorenz96_model(d, m)
    N, T = (3, 25) if d == 40 else (5, 40)
    t = o.irregular_times(rng, N, T, 0.012 * T)
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    ref = o.ekf_smoother(mdl, t, y)
    post = cd.cdnlgssm_smoother(P, y, t[..., None])
    assert relerr(post.filtered_means, ref["filtered_means"]) < 1e-9
    assert relerr(post.filtered_covariances, ref["filtered_covariances"]) < 1e-9
    assert relerr(post.smoothed_means, ref["smoothed_means"]) < 1e-8
    assert relerr(post.smoothed_covariances, ref["smoothed_covariances"]) < 1e-8
    _check(cd.cdnlgssm_filter(P, y, t[..., None]), o.ekf_filter(mdl, t, y), 1e-9)
    # fp32 engine vs the fp64 oracle
    post32 = cd.cdnlgssm_filter(P, y.astype(np.float32), t[..., None])
    assert post32.filtered_means.dtype == np.float32
    assert relerr(post32.filtered_means, ref["filtered_means"]) < 2e-4


def test_lorenz96_d40_dense_noise_matrices(hip_lib):
    """Config 4's kernels own the packed upper triangle and use symmetric images throughout: dense (non-diagonal) L, Qc, R, P0 and a
    non-zero initial mean must still reproduce the oracle (filter: all four moment arrays; smoother)."""
    rng = np.random.default_rng(123)
    d = 40

    def spd(n, s):
        A = rng.standard_normal((n, n))
        return A @ A.T / n * s + 0.3 * np.eye(n)

    mdl = o.Model(o.Lorenz96Drift(8.0), np.eye(d) + 0.1 * rng.standard_normal((d, d)), spd(d, 0.5), np.eye(d), np.zeros(d), spd(d, 0.7),
                  8.0 + rng.standard_normal(d), spd(d, 1.0))
    N, T = 5, 10
    t = o.irregular_times(rng, N, T, 0.015 * T)
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    ref = o.ekf_smoother(mdl, t, y)
    post = cd.cdnlgssm_smoother(P, y, t[..., None])
    assert _ffi.lib().cdkf_last_kernel().decode().startswith("ekf_smoother_wave_l96_kernel<double")
    for k in ("filtered_means", "filtered_covariances", "smoothed_means", "smoothed_covariances"):
        assert relerr(getattr(post, k), ref[k]) < 1e-10, k
    assert relerr(post.marginal_loglik, ref["marginal_loglik"]) < 1e-11
    flt = cd.cdnlgssm_filter(P, y, t[..., None])
    assert _ffi.lib().cdkf_last_kernel().decode().startswith("ekf_filter_wave_l96_kernel<double")
    _check(flt, o.ekf_filter(mdl, t, y), 1e-10)


def test_c4_full_length_against_the_oracle(hip_lib):
    """BASELINE config 4 at its full length (500 irregular observations at the benchmark's time density, d = m = 40): both
    wavefront-per-trajectory sweeps against the oracle on the first two of eight trajectories, every step -- rounding must not
    accumulate over the scan (filtered / smoothed moments 1e-9, log-likelihood 1e-11)."""
    rng = np.random.default_rng(4)
    mdl = lorenz96_model(40, 40)
    N, T = 8, 500
    u = rng.uniform(0.0, 1.0, size=(N, T))
    cs = np.cumsum(u, axis=1)
    t = cs / cs[:, -1:] * (0.005 * T)
    y = 8.0 + rng.standard_normal((N, T, 40))
    post = cd.cdnlgssm_smoother(params_from(mdl), y, t[..., None])
    assert _ffi.lib().cdkf_last_kernel().decode().startswith("ekf_smoother_wave_l96_kernel<double")
    ref = o.ekf_smoother(mdl, t[:2], y[:2])
    for k in ("filtered_means", "filtered_covariances", "smoothed_means", "smoothed_covariances"):
        assert relerr(getattr(post, k)[:2], ref[k]) < 1e-9, k
    np.testing.assert_allclose(post.marginal_loglik[:2], ref["marginal_loglik"], rtol=1e-11)
    assert np.all(np.isfinite(post.smoothed_covariances))


def test_c4_full_slice_properties(hip_lib):
    """BASELINE config 4's per-GPU slice at FULL size (2048 trajectories x 500 irregular observations, Lorenz-96 d = m = 40, fp64,
    filter + smoother; extended_kalman_smoother, inference_ekf.py:450-539), device-resident as bench.py runs it: four 13 GB
    covariance arrays, element offsets beyond 4 GiB.  Checked through size-independent properties: no status flag, the device
    log-likelihood sum equals the host sum, a random subset (the LAST trajectory included: the largest offsets) re-run through the
    ORACLE matches on every step (1e-9), the same subset as a batch of its own gives bitwise the same numbers, symmetric outputs."""
    import ctypes as C
    from cd_dynamax_amd.models import _model_block
    from cd_dynamax_amd._ffi import DeviceArray
    L = hip_lib
    rng = np.random.default_rng(41)
    d = 40
    mdl = lorenz96_model(d, d)
    N, T = 2048, 500
    u = rng.uniform(0.0, 1.0, size=(N, T))
    cs = np.cumsum(u, axis=1)
    t = cs / cs[:, -1:] * (0.005 * T)
    y = 8.0 + rng.standard_normal((N, T, d))
    sub = np.sort(np.concatenate([rng.choice(N - 1, size=2, replace=False), [N - 1]]))
    y[sub] = o.simulate(mdl, t[sub], rng)
    blk = _model_block(params_from(mdl))
    opts = _ffi.default_opts()
    opts.layout = _ffi.LAYOUT_TN

    def run(tt, yy):
        n = tt.shape[0]
        t_d = DeviceArray.from_numpy(np.ascontiguousarray(tt.T))
        y_d = DeviceArray.from_numpy(np.ascontiguousarray(yy.transpose(1, 0, 2)))
        ll, st, llsum = DeviceArray((n,), np.float64), DeviceArray.from_numpy(np.zeros(n, np.int32)), DeviceArray((1,), np.float64)
        bufs = [DeviceArray((T, n) + w, np.float64) for w in ((d,), (d, d), (d,), (d, d))]
        _ffi.check(L.cdkf_ekf_smoother_f64_dev(C.byref(blk.c), C.byref(opts), n, T, t_d.ptr, y_d.ptr, ll.ptr, *[b.ptr for b in bufs],
                                               st.ptr, None))
        kern = L.cdkf_last_kernel().decode()
        _ffi.check(L.cdkf_ll_sum_f64_dev(ll.ptr, n, llsum.ptr, None))
        _ffi.check(L.cdkf_synchronize(None))

        def rows(buf, w, which):  # [len(which), T] + w from the [T, n] + w device array: one small copy per (k, trajectory)
            out = np.empty((len(which), T) + w)
            e = int(np.prod(w)) * 8
            for a, n_ in enumerate(which):
                for k in range(T):
                    _ffi.check(L.cdkf_memcpy_d2h(out[a, k].ctypes.data_as(C.c_void_p), C.c_void_p(buf.ptr.value + (k * n + int(n_)) * e), e))
            return out
        res = dict(ll=ll.numpy(), st=st.numpy(), llsum=float(llsum.numpy()[0]), kern=kern, rows=rows, bufs=bufs)
        for a in (t_d, y_d, ll, st, llsum):
            a.free()
        return res

    full = run(t, y)
    assert full["kern"].startswith("ekf_smoother_wave_l96_kernel<double"), full["kern"]
    assert (full["st"] == 0).all() and np.isfinite(full["ll"]).all()
    assert abs(full["llsum"] - full["ll"].sum()) <= 1e-11 * abs(full["llsum"])
    got = {k: full["rows"](b, w, sub) for k, b, w in zip(("filtered_means", "filtered_covariances", "smoothed_means", "smoothed_covariances"),
                                                        full["bufs"], ((d,), (d, d), (d,), (d, d)))}
    for b in full["bufs"]:
        b.free()
    ref = o.ekf_smoother(mdl, t[sub], y[sub])
    np.testing.assert_allclose(full["ll"][sub], ref["marginal_loglik"], rtol=1e-11)
    for k, v in got.items():
        assert relerr(v, ref[k]) < 1e-9, k
    assert np.array_equal(got["smoothed_covariances"], np.swapaxes(got["smoothed_covariances"], -1, -2))
    small = run(t[sub], y[sub])
    np.testing.assert_array_equal(small["ll"], full["ll"][sub])
    for (k, v), b, w in zip(got.items(), small["bufs"], ((d,), (d, d), (d,), (d, d))):
        np.testing.assert_array_equal(small["rows"](b, w, range(len(sub))), v, err_msg=k)
    for b in small["bufs"]:
        b.free()


def test_c5_full_slice_properties(hip_lib):
    """BASELINE config 5's per-GPU slice at FULL size (1024 trajectories x 1000 irregular observations, 8 -> 64 -> 64 -> 8 tanh MLP,
    d = 8, m = 4, the reference's default state_order='second', fp64): the SGD objective's value and all 5 256 weight gradients per
    trajectory (value_and_grad of ssm_temissions.py:550-568) through the forward sweep + reverse sweep with the 8 GB checkpoint
    workspace.  Properties: no status flag, every value finite, the log-likelihood is the LL-only filter sweep's, a subset (first,
    last and a random trajectory) re-run through the ORACLE's discrete adjoint matches (1e-8 of the gradient scale over a
    thousand steps), the subset as a batch of its own gives bitwise the same numbers, the device-side sums equal the host sums."""
    import ctypes as C
    from cd_dynamax_amd.models import _model_block
    from cd_dynamax_amd._ffi import DeviceArray
    L = hip_lib
    rng = np.random.default_rng(52)
    mdl = mlp_model(rng)
    N, T = 1024, 1000
    u = rng.uniform(0.0, 1.0, size=(N, T))
    cs = np.cumsum(u, axis=1)
    t = cs / cs[:, -1:] * (0.005 * T)
    y = rng.standard_normal((N, T, 4))
    sub = np.array([0, int(rng.integers(1, N - 1)), N - 1])
    P = params_from(mdl)
    ll, g = cd.cdnlgssm_loglik_and_grad(P, y, t[..., None])
    assert L.cdkf_last_kernel().decode().startswith("ekf_adjoint_wave8_kernel<double, true, false>")
    flat = np.concatenate([np.asarray(a).reshape(N, -1) for a in g], axis=-1)
    assert flat.shape == (N, 5256) and np.isfinite(flat).all() and np.isfinite(ll).all()
    post = cd.cdnlgssm_filter(P, y, t[..., None], output_fields=[])
    np.testing.assert_allclose(ll, post.marginal_loglik, rtol=1e-12)
    ll_ref, g_ref = o.ekf_loglik_grad_adjoint(mdl, t[sub], y[sub], state_order="second")
    np.testing.assert_allclose(ll[sub], ll_ref, rtol=1e-11)
    assert np.abs(flat[sub] - g_ref).max() < 1e-8 * np.abs(g_ref).max()
    ll3, g3 = cd.cdnlgssm_loglik_and_grad(P, y[sub], t[sub][..., None])
    np.testing.assert_array_equal(ll3, ll[sub])
    np.testing.assert_array_equal(np.concatenate([np.asarray(a).reshape(3, -1) for a in g3], axis=-1), flat[sub])
    # the device-resident composition fit_sgd uses: sweeps -> cdkf_ll_sum / cdkf_grad_sum, status flags
    blk = _model_block(P)
    opts = _ffi.default_opts()
    opts.layout = _ffi.LAYOUT_TCN
    t_d = DeviceArray.from_numpy(np.ascontiguousarray(t.T))
    y_d = DeviceArray.from_numpy(np.ascontiguousarray(y.transpose(1, 2, 0)))
    ll_d, st, gd = DeviceArray((N,), np.float64), DeviceArray.from_numpy(np.zeros(N, np.int32)), DeviceArray((N, 5256), np.float64)
    sums = DeviceArray((1 + 5256,), np.float64)
    _ffi.check(L.cdkf_ekf_loglik_grad_f64_dev(C.byref(blk.c), C.byref(opts), N, T, t_d.ptr, y_d.ptr, ll_d.ptr, gd.ptr, st.ptr, None))
    _ffi.check(L.cdkf_ll_sum_f64_dev(ll_d.ptr, N, sums.ptr, None))
    _ffi.check(L.cdkf_grad_sum_f64_dev(gd.ptr, N, 5256, C.c_void_p(sums.ptr.value + 8), None))
    _ffi.check(L.cdkf_synchronize(None))
    s = sums.numpy()
    assert (st.numpy() == 0).all()
    assert abs(s[0] - ll.sum()) <= 1e-11 * abs(ll.sum())
    assert np.abs(s[1:] - flat.sum(0)).max() <= 1e-10 * np.abs(flat.sum(0)).max()
    for a in (t_d, y_d, ll_d, st, gd, sums):
        a.free()


@pytest.mark.parametrize("d", [12, 16, 20, 24, 28, 32, 36])
def test_lorenz96_wavefront_kernels_other_state_dimensions(hip_lib, d, monkeypatch):
    """The wavefront-per-trajectory Lorenz-96 sweeps (config 4's kernels) at the other state dimensions they are instantiated for --
    every multiple of four from 12 to 40: the sixteen-wide panels of the factorisation and the triangular solves end in a panel of
    4, 8 or 12 columns (extended_kalman_filter / _smoother are shape-generic, inference_ekf.py:202-326, 450-539).  Filter (all four
    moment arrays) and smoother against the oracle, dense noise matrices, intervals of one to three steps; against the workgroup
    kernels they replace (CDKF_NO_WAVE40=1); fp32."""
    rng = np.random.default_rng(100 + d)

    def spd(n, s):
        A = rng.standard_normal((n, n))
        return A @ A.T / n * s + 0.3 * np.eye(n)

    mdl = o.Model(o.Lorenz96Drift(8.0), np.eye(d) + 0.1 * rng.standard_normal((d, d)), spd(d, 0.5), np.eye(d), np.zeros(d), spd(d, 0.7),
                  8.0 + rng.standard_normal(d), spd(d, 1.0))
    N, T = 5, 12
    t = o.irregular_times(rng, N, T, 0.015 * T)
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    ref = o.ekf_smoother(mdl, t, y)
    post = cd.cdnlgssm_smoother(P, y, t[..., None])
    assert _ffi.lib().cdkf_last_kernel().decode().startswith("ekf_smoother_wave_l96_kernel<double, %d>" % d)
    for k in ("filtered_means", "filtered_covariances", "smoothed_means", "smoothed_covariances"):
        assert relerr(getattr(post, k), ref[k]) < 1e-9, k
    assert relerr(post.marginal_loglik, ref["marginal_loglik"]) < 1e-10
    flt = cd.cdnlgssm_filter(P, y, t[..., None])
    assert _ffi.lib().cdkf_last_kernel().decode().startswith("ekf_filter_wave_l96_kernel<double, %d" % d)
    _check(flt, o.ekf_filter(mdl, t, y), 1e-9)
    flt32 = cd.cdnlgssm_filter(P, y.astype(np.float32), t[..., None])
    assert relerr(flt32.filtered_means, ref["filtered_means"]) < 5e-4
    monkeypatch.setenv("CDKF_NO_WAVE40", "1")
    wg = cd.cdnlgssm_smoother(P, y, t[..., None])
    assert _ffi.lib().cdkf_last_kernel().decode().startswith("ekf_smoother_wg_kernel")
    monkeypatch.delenv("CDKF_NO_WAVE40")
    assert relerr(post.smoothed_covariances, wg.smoothed_covariances) < 1e-9


@pytest.mark.parametrize("d", [46, 48])
def test_workgroup_kernels_eight_entries_per_thread(hip_lib, d, monkeypatch):
    """State dimensions 46 .. 64: the workgroup kernels own eight covariance entries per thread.  The -O3 build of that fp64
    instantiation returned NaN from the second observation on (ROCm 7.2; launch_wg8.hip is built at -O1 since): filter and smoother
    against the oracle, both precisions."""
    monkeypatch.setenv("CDKF_NO_WAVE40", "1")
    rng = np.random.default_rng(d)
    mdl = lorenz96_model(d, d)
    N, T = 2, 5
    t = o.irregular_times(rng, N, T, 0.012 * T)
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    ref = o.ekf_smoother(mdl, t, y)
    post = cd.cdnlgssm_smoother(P, y, t[..., None])
    assert _ffi.lib().cdkf_last_kernel().decode().startswith("ekf_smoother_wg_kernel<double, 8>")
    for k in ("filtered_means", "filtered_covariances", "smoothed_means", "smoothed_covariances"):
        assert relerr(getattr(post, k), ref[k]) < 1e-9, k
    assert relerr(post.marginal_loglik, ref["marginal_loglik"]) < 1e-10
    flt32 = cd.cdnlgssm_filter(P, y.astype(np.float32), t[..., None])
    assert relerr(flt32.filtered_means, ref["filtered_means"]) < 5e-4


@pytest.mark.parametrize("d,sel", [(36, "first20"), (40, "every_other"), (12, "shuffled5"), (24, "single"), (32, "first20"), (32, "every_other")])
def test_lorenz96_wavefront_kernels_partial_observation(hip_lib, d, sel, monkeypatch):
    """The wavefront-per-trajectory Lorenz-96 sweeps when the emission observes only SOME state components (_condition_on is shape-
    generic, inference_ekf.py:153-199): H = I[:m], every other component, a shuffled handful, a single one -- every row of H a unit
    vector.  The kernel runs the update in state coordinates on the order-d system [P + R on the observed block; identity elsewhere];
    the numbers must be those of the m x m system: filter (all four moment arrays, log-likelihood) and smoother against the oracle
    (1e-9) with a dense symmetric R, against the workgroup kernels (CDKF_NO_WAVE40=1), and in fp32."""
    rng = np.random.default_rng(300 + d)
    cols = {"first20": np.arange(20), "every_other": np.arange(0, d, 2), "shuffled5": rng.permutation(d)[:5], "single": np.array([7])}[sel]
    m = len(cols)
    H = np.eye(d)[cols]

    def spd(n, s):
        A = rng.standard_normal((n, n))
        return A @ A.T / n * s + 0.3 * np.eye(n)

    mdl = o.Model(o.Lorenz96Drift(8.0), np.eye(d), spd(d, 0.5), H, np.zeros(m), spd(m, 0.7), 8.0 + rng.standard_normal(d), spd(d, 1.0))
    N, T = 5, 14
    t = o.irregular_times(rng, N, T, 0.015 * T)
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    ref = o.ekf_smoother(mdl, t, y)
    post = cd.cdnlgssm_smoother(P, y, t[..., None])
    assert _ffi.lib().cdkf_last_kernel().decode().startswith("ekf_smoother_wave_l96_kernel<double, %d>" % d)
    for k in ("filtered_means", "filtered_covariances", "smoothed_means", "smoothed_covariances"):
        assert relerr(getattr(post, k), ref[k]) < 1e-9, k
    assert relerr(post.marginal_loglik, ref["marginal_loglik"]) < 1e-10
    flt = cd.cdnlgssm_filter(P, y, t[..., None])
    assert _ffi.lib().cdkf_last_kernel().decode().startswith("ekf_filter_wave_l96_kernel<double, %d" % d)
    _check(flt, o.ekf_filter(mdl, t, y), 1e-9)
    flt32 = cd.cdnlgssm_filter(P, y.astype(np.float32), t[..., None])
    assert _ffi.lib().cdkf_last_kernel().decode().startswith("ekf_filter_wave_l96_kernel<float, %d" % d)
    assert relerr(flt32.filtered_means, ref["filtered_means"]) < 5e-4
    monkeypatch.setenv("CDKF_NO_WAVE40", "1")
    wg = cd.cdnlgssm_filter(P, y, t[..., None])
    assert _ffi.lib().cdkf_last_kernel().decode().startswith("ekf_filter_wg_kernel")
    monkeypatch.delenv("CDKF_NO_WAVE40")
    assert relerr(flt.filtered_covariances, wg.filtered_covariances) < 1e-9
    np.testing.assert_allclose(flt.marginal_loglik, wg.marginal_loglik, rtol=1e-10)
    # a non-symmetric R, a bias or a row of H that is not a unit vector keep the workgroup kernels
    Rn = spd(m, 0.7)
    if m > 1:
        Rn[0, 1] += 0.05
        cd.cdnlgssm_filter(params_from(o.Model(mdl.drift, mdl.L, mdl.Qc, H, np.zeros(m), Rn, mdl.m0, mdl.P0)), y, t[..., None])
        assert _ffi.lib().cdkf_last_kernel().decode().startswith("ekf_filter_wg_kernel")
    H2 = H.copy()
    H2[0, cols[0]] = 0.5
    cd.cdnlgssm_filter(params_from(o.Model(mdl.drift, mdl.L, mdl.Qc, H2, np.zeros(m), mdl.R, mdl.m0, mdl.P0)), y, t[..., None])
    assert _ffi.lib().cdkf_last_kernel().decode().startswith("ekf_filter_wg_kernel")


def test_lorenz96_d40_backward_sweep_kernels_agree(hip_lib, monkeypatch):
    """Config 4's smoother: the wavefront-per-trajectory backward sweep (ekf_smoother_wave_l96_kernel) against the oracle at a
    batch that does not fill its last workgroup, against the workgroup kernel it replaced (CDKF_WG_BACKWARD=1), in fp32, and at
    the degenerate lengths T = 1 (nothing to integrate) and T = 2."""
    rng = np.random.default_rng(40)
    mdl = lorenz96_model(40, 40)
    P = params_from(mdl)
    N, T = 6, 12
    t = o.irregular_times(rng, N, T, 0.02 * T)  # some intervals take two or three Dormand-Prince steps
    y = o.simulate(mdl, t, rng)
    ref = o.ekf_smoother(mdl, t, y)
    post = cd.cdnlgssm_smoother(P, y, t[..., None])
    assert _ffi.lib().cdkf_last_kernel().decode().startswith("ekf_smoother_wave_l96_kernel<double")
    assert relerr(post.smoothed_means, ref["smoothed_means"]) < 1e-8
    assert relerr(post.smoothed_covariances, ref["smoothed_covariances"]) < 1e-8
    assert np.array_equal(post.smoothed_covariances, np.swapaxes(post.smoothed_covariances, -1, -2))
    monkeypatch.setenv("CDKF_WG_BACKWARD", "1")
    wg = cd.cdnlgssm_smoother(P, y, t[..., None])
    assert _ffi.lib().cdkf_last_kernel().decode().startswith("ekf_smoother_wg_kernel")
    monkeypatch.delenv("CDKF_WG_BACKWARD")
    assert relerr(post.smoothed_covariances, wg.smoothed_covariances) < 1e-9
    post32 = cd.cdnlgssm_smoother(P, y.astype(np.float32), t[..., None])
    assert post32.smoothed_means.dtype == np.float32
    assert relerr(post32.smoothed_means, ref["smoothed_means"]) < 5e-4
    for T_short in (1, 2):
        ps = cd.cdnlgssm_smoother(P, y[:2, :T_short], t[:2, :T_short, None])
        rs = o.ekf_smoother(mdl, t[:2, :T_short], y[:2, :T_short])
        assert relerr(ps.smoothed_covariances, rs["smoothed_covariances"]) < 1e-9
        assert relerr(ps.smoothed_means, rs["smoothed_means"]) < 1e-9


@pytest.mark.parametrize("order", ["first", "second", "zeroth"])
def test_mlp_drift_orders(hip_lib, order):
    """Config C5 shape (d=8, m=4, 2x64 tanh MLP).  'second' exercises the reference's 0.5*trace(H_t @ P) quirk,
    i.e. 0.5 * P grad(div f), which is non-zero for an MLP."""
    rng = np.random.default_rng(42)
    mdl = mlp_model(rng)
    N, T = 4, 30
    t = o.irregular_times(rng, N, T, 0.02 * T)
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    ref = o.ekf_filter(mdl, t, y, state_order=order, cov_rescaling=0.9)
    post = cd.cdnlgssm_filter(P, y, t[..., None], cd.EKFHyperParams(state_order=order, cov_rescaling=0.9))
    _check(post, ref, 1e-9)
    if order == "second":
        first = o.ekf_filter(mdl, t, y, state_order="first")
        assert relerr(first["filtered_means"], ref["filtered_means"]) > 1e-6  # the term is really there


def test_mlp_smoother_and_num_iter(hip_lib):
    rng = np.random.default_rng(43)
    mdl = mlp_model(rng, d=8, m_obs=4, h=32)
    N, T = 3, 20
    t = o.irregular_times(rng, N, T, 0.02 * T)
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    ref = o.ekf_smoother(mdl, t, y)
    post = cd.cdnlgssm_smoother(P, y, t[..., None])
    assert relerr(post.smoothed_means, ref["smoothed_means"]) < 1e-8
    assert relerr(post.smoothed_covariances, ref["smoothed_covariances"]) < 1e-8
    _check(cd.cdnlgssm_filter(P, y, t[..., None], num_iter=2), o.ekf_filter(mdl, t, y, num_iter=2), 1e-9)


@pytest.mark.parametrize("d,m", [(5, 3), (6, 7), (16, 4)])
def test_linear_models_beyond_register_shapes(hip_lib, d, m):
    rng = np.random.default_rng(100 + d)
    mdl = linear_model(rng, d, m)
    N, T = 4, 30
    t = o.irregular_times(rng, N, T, 1.5)
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    _check(cd.cdnlgssm_filter(P, y, t[..., None]), o.ekf_filter(mdl, t, y), 1e-9)
    ref = o.ekf_smoother(mdl, t, y)
    assert relerr(cd.cdnlgssm_smoother(P, y, t[..., None]).smoothed_covariances, ref["smoothed_covariances"]) < 1e-8
    _check(cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams()), o.ukf_filter(mdl, t, y), 1e-8)


def test_wg_and_reg_kernels_agree_on_shared_shape(hip_lib):
    """Lorenz-96 with d = 4, m = 4 has no register instantiation -> workgroup kernel; a linear (4,4) model runs on the
    register kernel.  Both must reproduce the oracle; this guards the dispatch boundary."""
    rng = np.random.default_rng(7)
    mdl = lorenz96_model(4, 4)
    t = o.irregular_times(rng, 3, 20, 0.3)
    y = o.simulate(mdl, t, rng)
    _check(cd.cdnlgssm_filter(params_from(mdl), y, t[..., None]), o.ekf_filter(mdl, t, y), 1e-9)


@pytest.mark.parametrize("name", GOLDEN_WIDE)
def test_wide_golden_vectors(hip_lib, name):
    """Committed fixtures at the BASELINE config 4 / 5 shapes (Lorenz-96 d=40; MLP d=8, first and second order)."""
    g = load_golden(name)
    mdl = model_from_fixture(g)
    P = params_from(mdl)
    y, t = g["y"], g["t"][..., None]
    orders = [k[4:-3] for k in g.files if k.startswith("ekf_") and k.endswith("_ll")]
    for order in orders:
        post = cd.cdnlgssm_filter(P, y, t, cd.EKFHyperParams(state_order=order))
        assert relerr(post.marginal_loglik, g[f"ekf_{order}_ll"]) < 1e-9
        assert relerr(post.filtered_means, g[f"ekf_{order}_filtered_means"]) < 1e-9
        assert relerr(post.filtered_covariances[:, -1], g[f"ekf_{order}_filtered_cov_last"]) < 1e-9
        assert relerr(post.predicted_covariances[:, -1], g[f"ekf_{order}_predicted_cov_last"]) < 1e-9
    sm = cd.cdnlgssm_smoother(P, y, t, cd.EKFHyperParams(state_order=orders[-1]))
    assert relerr(sm.smoothed_means, g["eks_smoothed_means"]) < 1e-8
    assert relerr(sm.smoothed_covariances[:, 0], g["eks_smoothed_cov_first"]) < 1e-8


@pytest.mark.parametrize("case", ["l96_6_3", "l96_12_12", "l96_40_40", "mlp_8_4"])
def test_unscented_filter_on_workgroup_kernels(hip_lib, case):
    """unscented_kalman_filter (inference_ukf.py:206-308) beyond the register shapes: banded Lorenz-96 fast path, the
    generic one-drift-per-sigma-point path (MLP), general and selection emissions, fp64 and fp32."""
    rng = np.random.default_rng(abs(hash(case)) % 997)
    if case.startswith("l96"):
        _, d, m = case.split("_")
        mdl = lorenz96_model(int(d), int(m))
        N, T = (2, 12) if int(d) == 40 else (4, 25)
    else:
        mdl = mlp_model(rng, 8, 4, 16)
        N, T = 3, 15
    t = o.irregular_times(rng, N, T, 0.012 * T)
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    ref = o.ukf_filter(mdl, t, y)
    _check(cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams()), ref, 1e-8)
    ref2 = o.ukf_filter(mdl, t, y, alpha=1.1, beta=1.0, kappa=0.5)
    _check(cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams(alpha=1.1, beta=1.0, kappa=0.5)), ref2, 1e-8)
    post32 = cd.cdnlgssm_filter(P, y.astype(np.float32), t[..., None], cd.UKFHyperParams())
    assert relerr(post32.filtered_means, ref["filtered_means"]) < 5e-4


@pytest.mark.parametrize("d,m,h", [(8, 4, (64, 64)), (4, 2, (7, 5)), (8, 8, (32, 16)), (3, 1, (64, 9))])
def test_mlp_loglik_gradient_reverse_sweep(hip_lib, d, m, h):
    """Reverse-sweep gradient w.r.t. every weight and bias of the MLP drift (BASELINE config 5's SGD objective; reference:
    jax.value_and_grad of the fit_sgd loss, ssm_temissions.py:550-568) against the oracle's discrete adjoint (pinned to
    finite differences in tests/test_oracle.py).  Intervals of 1..30 Dormand-Prince steps exercise the chunked replay."""
    rng = np.random.default_rng(9)
    mdl = mlp_model(rng, d, m, h)
    N, T = 5, 9
    t = o.irregular_times(rng, N, T, 0.02)
    t[:, 5:] += 0.25        # one long interval (more steps than one replay chunk holds)
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    hyp = cd.EKFHyperParams(state_order="first")
    ll_ref, g_ref = o.ekf_loglik_grad_adjoint(mdl, t, y)
    ll, g = cd.cdnlgssm_loglik_and_grad(P, y, t[..., None], hyp)
    np.testing.assert_allclose(ll, ll_ref, rtol=1e-10)
    flat = np.concatenate([np.asarray(a).reshape(N, -1) for a in g], axis=-1)
    assert flat.shape == g_ref.shape
    scale = np.abs(g_ref).max()
    assert np.abs(flat - g_ref).max() < 1e-8 * scale, np.abs(flat - g_ref).max() / scale
    assert g.W2.shape == (N,) + mdl.drift.W2.shape and g.b3.shape == (N, d)
    # fp32 kernels
    ll32, g32 = cd.cdnlgssm_loglik_and_grad(P, y.astype(np.float32), t[..., None].astype(np.float32), hyp)
    flat32 = np.concatenate([np.asarray(a).reshape(N, -1) for a in g32], axis=-1)
    assert flat32.dtype == np.float32 and np.abs(flat32 - g_ref).max() < 2e-2 * scale
    # the reference's default, state_order='second': the mean also moves with 0.5 P grad(div f), whose reverse needs third
    # derivatives of the drift (oracle: divgrad_vjp, FD-pinned in tests/test_oracle.py)
    ll_ref2, g_ref2 = o.ekf_loglik_grad_adjoint(mdl, t, y, state_order="second")
    ll2, g2 = cd.cdnlgssm_loglik_and_grad(P, y, t[..., None])
    np.testing.assert_allclose(ll2, ll_ref2, rtol=1e-10)
    flat2 = np.concatenate([np.asarray(a).reshape(N, -1) for a in g2], axis=-1)
    scale2 = np.abs(g_ref2).max()
    assert np.abs(flat2 - g_ref2).max() < 1e-8 * scale2, np.abs(flat2 - g_ref2).max() / scale2
    assert np.abs(g_ref2 - g_ref).max() > 1e-6 * scale2  # the term is really there
    ll32b, g32b = cd.cdnlgssm_loglik_and_grad(P, y.astype(np.float32), t[..., None].astype(np.float32))
    flat32b = np.concatenate([np.asarray(a).reshape(N, -1) for a in g32b], axis=-1)
    # (fp32 through up to 30 steps per interval and third derivatives: the error is rounding amplified by the flow, and moves by a factor
    #  of two with the order of the sums -- 1.2e-2 .. 1.9e-2 with the round-2 kernels, 3.4e-2 .. 4.2e-2 with the DPP reductions of round 3)
    assert np.abs(flat32b - g_ref2).max() < 8e-2 * scale2


def test_c5_long_scan_value_and_gradient_against_the_oracle(hip_lib):
    """BASELINE config 5's shape (d = 8, m = 4, 2 x 64 tanh MLP, the reference's default state_order='second') over a long scan:
    log-likelihood and all 5 256 weight gradients of the first of four trajectories against the oracle's discrete adjoint, 400
    irregular observations at the benchmark's time density -- the reverse sweep's accumulators must not lose digits over the scan."""
    rng = np.random.default_rng(55)
    mdl = mlp_model(rng)
    N, T = 4, 400
    u = rng.uniform(0.0, 1.0, size=(N, T))
    cs = np.cumsum(u, axis=1)
    t = cs / cs[:, -1:] * (0.005 * T)
    y = rng.standard_normal((N, T, 4))
    ll, g = cd.cdnlgssm_loglik_and_grad(params_from(mdl), y, t[..., None])
    ll_ref, g_ref = o.ekf_loglik_grad_adjoint(mdl, t[:1], y[:1], state_order="second")
    np.testing.assert_allclose(ll[:1], ll_ref, rtol=1e-11)
    flat = np.concatenate([np.asarray(a).reshape(N, -1) for a in g], axis=-1)[:1]
    assert np.abs(flat - g_ref).max() < 1e-9 * np.abs(g_ref).max()


def test_reverse_sweep_slope_checkpoints_do_not_change_the_gradient(hip_lib, monkeypatch):
    """The forward sweep's stage-slope checkpoints (first two steps of every interval) against re-integration
    (CDKF_ADJ_CKPT_STEPS=0) and against four checkpointed steps: same gradient to rounding, on intervals of 1, 2, 3 and more
    steps (both paths in one sweep)."""
    rng = np.random.default_rng(21)
    mdl = mlp_model(rng, 6, 3, (20, 12))
    N, T = 5, 14
    t = o.irregular_times(rng, N, T, 0.012 * T)   # gaps around dt0 = 0.01: one to three steps
    t[:, 9:] += 0.07                              # and one of eight
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    out = {}
    for steps in ("2", "0", "4"):
        monkeypatch.setenv("CDKF_ADJ_CKPT_STEPS", steps)
        ll, g = cd.cdnlgssm_loglik_and_grad(P, y, t[..., None])
        out[steps] = np.concatenate([np.asarray(a).reshape(N, -1) for a in g], axis=-1)
    monkeypatch.delenv("CDKF_ADJ_CKPT_STEPS")
    scale = np.abs(out["0"]).max()
    assert np.abs(out["2"] - out["0"]).max() < 1e-11 * scale
    assert np.abs(out["4"] - out["0"]).max() < 1e-11 * scale
    _, g_ref = o.ekf_loglik_grad_adjoint(mdl, t, y, state_order="second")
    assert np.abs(out["2"] - g_ref).max() < 1e-8 * scale


@pytest.mark.parametrize("solver", ["tsit5", "bosh3", "euler"])
def test_reverse_sweep_other_runge_kutta_methods(hip_lib, solver):
    """cdnlgssm_loglik_and_grad[_all] with diffeqsolve_settings={'solver': ...}: the reverse sweep reads the tableau the forward sweep
    (the workgroup kernel for non-default methods) integrated with -- MLP d = 5 (both state orders) and Lorenz-96 d = 6."""
    rng = np.random.default_rng(41)
    settings = {"solver": solver}
    for mdl, orders in ((mlp_model(rng, 5, 2, (9, 7)), ("first", "second")), (lorenz96_model(6, 3), ("second",))):
        N, T = 4, 9
        t = o.irregular_times(rng, N, T, 0.025)
        t[:, 5:] += 0.06
        y = o.simulate(mdl, t, rng)
        P = params_from(mdl)
        for order in orders:
            with o.use_solver(solver):
                ll_ref, g_ref, ex = o.ekf_loglik_grad_adjoint(mdl, t, y, full=True, state_order=order)
            hyp = cd.EKFHyperParams(state_order=order, diffeqsolve_settings=settings)
            ll, g = cd.cdnlgssm_loglik_and_grad_all(P, y, t[..., None], hyp)
            np.testing.assert_allclose(ll, ll_ref, rtol=1e-10)
            flat = np.concatenate([np.asarray(a_).reshape(N, -1) for a_ in g.dynamics.drift], axis=-1)
            scale = np.abs(g_ref).max()
            assert np.abs(flat - g_ref).max() < 1e-8 * scale, (order, np.abs(flat - g_ref).max() / scale)
            assert np.abs(np.asarray(g.emissions.emission_cov.params) - ex["R"]).max() < 1e-8 * np.abs(ex["R"]).max()


@pytest.mark.parametrize("solver,ctrl", [("tsit5", dict(rtol=1e-5, atol=1e-7)), ("dopri5", dict(rtol=1e-4, atol=1e-6, pcoeff=0.2, icoeff=0.5))])
def test_reverse_sweep_under_adaptive_steps(hip_lib, solver, ctrl, monkeypatch):
    """value-and-gradient under diffrax.PIDController on the reverse sweep: the forward (workgroup) sweep logs the step sizes it
    accepts, the reverse sweep replays them as constants.  MLP d = 5 (second order), Lorenz-96 d = 6 and d = 12 (the workgroup-per-
    trajectory reverse sweep) against the oracle doing the same (itself equal to the forward-sensitivity oracle, tests/test_oracle.py); a log too short for an interval raises MAX_STEPS."""
    rng = np.random.default_rng(51)
    settings = {"solver": solver, "dt0": 0.05, "stepsize_controller": cd.PIDController(**ctrl)}
    for mdl, order in ((mlp_model(rng, 5, 2, (9, 7)), "second"), (lorenz96_model(6, 3), "first"), (lorenz96_model(12, 5), "second")):
        N, T = 4, 8
        t = o.irregular_times(rng, N, T, 0.05)
        t[:, 4:] += 0.15
        y = o.simulate(mdl, t, rng)
        P = params_from(mdl)
        with o.use_solver(solver, adaptive=ctrl):
            ll_ref, g_ref, ex = o.ekf_loglik_grad_adjoint(mdl, t, y, dt0=0.05, full=True, state_order=order)
        hyp = cd.EKFHyperParams(state_order=order, diffeqsolve_settings=settings)
        ll, g = cd.cdnlgssm_loglik_and_grad_all(P, y, t[..., None], hyp)
        np.testing.assert_allclose(ll, ll_ref, rtol=1e-9)
        flat = np.concatenate([np.asarray(a_).reshape(N, -1) for a_ in g.dynamics.drift], axis=-1)
        scale = np.abs(g_ref).max()
        assert np.abs(flat - g_ref).max() < 1e-7 * scale, np.abs(flat - g_ref).max() / scale
        assert np.abs(np.asarray(g.initial.mean.params) - ex["m0"]).max() < 1e-7 * np.abs(ex["m0"]).max()
    from cd_dynamax_amd import models
    monkeypatch.setenv("CDKF_ADJ_DT_CAP", "1")
    mb = models._model_block(P)
    op = models._opts(hyp, 1)
    _, _, status = _ffi.loglik_grad(mb, op, t, y, np.float64)[:3]
    assert (status & 4).any()


def _general_model(rng, drift, d, m):
    """Non-diagonal L, Qc, R, P0, a dense H with bias: every parameter of the model carries a non-trivial gradient."""
    A, B, C = rng.standard_normal((d, d)), rng.standard_normal((m, m)), rng.standard_normal((d, d))
    return o.Model(drift, np.eye(d) + 0.2 * rng.standard_normal((d, d)), A @ A.T / d + 0.3 * np.eye(d),
                   rng.standard_normal((m, d)), 0.1 * rng.standard_normal(m), B @ B.T / m + 0.2 * np.eye(m),
                   rng.standard_normal(d), C @ C.T / d + 0.5 * np.eye(d))


@pytest.mark.parametrize("kind,d,m", [("mlp", 8, 4), ("mlp", 5, 3), ("linear", 4, 2), ("lorenz63", 3, 2), ("lorenz96", 6, 3),
                                      ("lorenz96", 8, 8), ("linear", 1, 1), ("lorenz96", 12, 5), ("linear", 10, 3), ("lorenz96", 20, 20),
                                      ("linear", 9, 12), ("mlp", 12, 6), ("mlp", 20, 9)])
def test_loglik_gradient_all_parameters(hip_lib, kind, d, m):
    """cdnlgssm_loglik_and_grad_all: one forward + one reverse sweep gives d ll / d(every parameter) -- the full pytree
    jax.grad(marginal_log_prob) returns in the reference -- against the oracle's discrete adjoint (FD-pinned).  State or emission
    dimension beyond eight: the workgroup-per-trajectory reverse sweep (ekf_adjoint_wg_kernel; one interval of the grid takes more
    steps than a replay chunk keeps starts for) -- from round 4 on with the MLP drift as well (VERDICT r3 "missing" 3: the network's
    reverse pass on the workgroup's threads, both state orders; ragged hidden sizes 24 / 40)."""
    rng = np.random.default_rng(13)
    if kind == "mlp":
        drift = mlp_model(rng, d, m, (24, 40)).drift
    elif kind == "linear":
        drift = linear_model(rng, d, m).drift
    elif kind == "lorenz63":
        drift = o.lorenz63_model(m).drift
    else:
        drift = lorenz96_model(d, m).drift
    mdl = _general_model(rng, drift, d, m)
    N, T = 6, 12
    t = o.irregular_times(rng, N, T, 0.02)
    t[:, 7:] += 0.12
    y = o.simulate(mdl, t, rng)
    hyp = cd.EKFHyperParams(state_order="first")
    ll_ref, g_ref, ex = o.ekf_loglik_grad_adjoint(mdl, t, y, full=True)
    ll, g = cd.cdnlgssm_loglik_and_grad_all(params_from(mdl), y, t[..., None], hyp)
    np.testing.assert_allclose(ll, ll_ref, rtol=1e-10)
    assert _ffi.lib().cdkf_last_kernel().decode().startswith("ekf_adjoint_wg_kernel<double") == (max(d, m) > 8)

    def close(a, b, name):
        scale = np.abs(b).max() + 1e-300
        assert np.abs(np.asarray(a) - b).max() < 1e-8 * scale, (name, np.abs(np.asarray(a) - b).max() / scale)

    dr = g.dynamics.drift
    flat = np.concatenate([np.asarray(a).reshape(N, -1) for a in dr], axis=-1)
    close(flat, g_ref, "drift")
    close(g.initial.mean.params, ex["m0"], "m0")
    close(g.initial.cov.params, ex["P0"], "P0")
    close(g.dynamics.diffusion_coefficient.params, ex["L"], "L")
    close(g.dynamics.diffusion_cov.params, ex["Qc"], "Qc")
    close(g.emissions.emission_function.weights, ex["H"], "H")
    close(g.emissions.emission_function.bias, ex["bias"], "bias")
    close(g.emissions.emission_cov.params, ex["R"], "R")
    # unbatched call: same numbers without the leading axis
    ll1, g1 = cd.cdnlgssm_loglik_and_grad_all(params_from(mdl), y[2], t[2][:, None], hyp)
    assert np.ndim(ll1) == 0 and g1.emissions.emission_cov.params.shape == (m, m)
    close(g1.emissions.emission_cov.params, ex["R"][2], "R[2]")
    if kind == "mlp":  # default hyper-parameters = state_order 'second': every leaf again
        ll_ref, g_ref, ex = o.ekf_loglik_grad_adjoint(mdl, t, y, full=True, state_order="second")
        ll, g = cd.cdnlgssm_loglik_and_grad_all(params_from(mdl), y, t[..., None])
        np.testing.assert_allclose(ll, ll_ref, rtol=1e-10)
        close(np.concatenate([np.asarray(a).reshape(N, -1) for a in g.dynamics.drift], axis=-1), g_ref, "drift (second)")
        close(g.initial.mean.params, ex["m0"], "m0 (second)")
        close(g.initial.cov.params, ex["P0"], "P0 (second)")
        close(g.dynamics.diffusion_cov.params, ex["Qc"], "Qc (second)")
        close(g.emissions.emission_function.weights, ex["H"], "H (second)")
        close(g.emissions.emission_cov.params, ex["R"], "R (second)")


@pytest.mark.parametrize("kind,d,m,num_iter", [("linear", 4, 2, 2), ("mlp", 5, 3, 2), ("lorenz96", 6, 3, 3), ("lorenz63", 3, 2, 2)])
def test_loglik_gradient_with_iterated_updates(hip_lib, kind, d, m, num_iter):
    """VERDICT r3 "missing" 3 (num_iter > 1 in the reverse sweeps): the reference's iterated update (inference_ekf.py:153-199: every
    iteration from the previous one's posterior, symmetrize once at the end, the log-likelihood term on the first one's inputs) reversed
    iteration by iteration in ekf_adjoint_wave8_kernel (d, m <= 8) -- every leaf against the oracle's adjoint (FD-pinned at num_iter
    2 and 3 in tests/test_oracle.py); the value is cdnlgssm_filter(num_iter=...)'s; beyond eight dimensions the call is refused."""
    rng = np.random.default_rng(40 + d)
    if kind == "mlp":
        drift = mlp_model(rng, d, m, (12, 9)).drift
    elif kind == "linear":
        drift = linear_model(rng, d, m).drift
    elif kind == "lorenz63":
        drift = o.lorenz63_model(m).drift
    else:
        drift = lorenz96_model(d, m).drift
    mdl = _general_model(rng, drift, d, m)
    N, T = 5, 9
    t = o.irregular_times(rng, N, T, 0.03)
    t[:, 5:] += 0.05
    y = o.simulate(mdl, t, rng)
    hyp = cd.EKFHyperParams(state_order="first")
    ll_ref, g_ref, ex = o.ekf_loglik_grad_adjoint(mdl, t, y, full=True, num_iter=num_iter)
    ll, g = cd.cdnlgssm_loglik_and_grad_all(params_from(mdl), y, t[..., None], hyp, num_iter=num_iter)
    assert _ffi.lib().cdkf_last_kernel().decode().startswith("ekf_adjoint_wave8_kernel<double")
    np.testing.assert_allclose(ll, ll_ref, rtol=1e-10)
    post = cd.cdnlgssm_filter(params_from(mdl), y, t[..., None], hyp, num_iter=num_iter, output_fields=[])
    np.testing.assert_allclose(ll, post.marginal_loglik, rtol=1e-10)
    ll1, g1 = cd.cdnlgssm_loglik_and_grad_all(params_from(mdl), y, t[..., None], hyp)
    assert np.abs(ll - ll1).max() > 1e-8 * np.abs(ll).max()              # not the single update's numbers

    def close(a, b, name):
        scale = np.abs(b).max() + 1e-300
        assert np.abs(np.asarray(a) - b).max() < 1e-8 * scale, (name, np.abs(np.asarray(a) - b).max() / scale)

    close(np.concatenate([np.asarray(a).reshape(N, -1) for a in g.dynamics.drift], axis=-1), g_ref, "drift")
    close(g.initial.mean.params, ex["m0"], "m0")
    close(g.initial.cov.params, ex["P0"], "P0")
    close(g.dynamics.diffusion_cov.params, ex["Qc"], "Qc")
    close(g.emissions.emission_function.weights, ex["H"], "H")
    close(g.emissions.emission_function.bias, ex["bias"], "bias")
    close(g.emissions.emission_cov.params, ex["R"], "R")
    big = _general_model(rng, lorenz96_model(20, 4).drift, 20, 4)   # (up to sixteen dimensions the tangent sweep iterates: tests/test_ukf_tangent.py)
    with pytest.raises(NotImplementedError):
        cd.cdnlgssm_loglik_and_grad_all(params_from(big), np.zeros((2, 4, 4)), np.arange(4.0)[None, :, None].repeat(2, 0), hyp, num_iter=2)


@pytest.mark.parametrize("sweep", ["ekf_adjoint_wave2_l96_kernel", "ekf_adjoint_wave_l96_kernel", "ekf_adjoint_wg_kernel"])
def test_lorenz96_d40_value_and_gradient(hip_lib, sweep, monkeypatch):
    """BASELINE config 4's model (Lorenz-96, d = m = 40, H = I) can be trained: value and gradient of the EKF log-likelihood w.r.t. the
    forcing and every other parameter -- forward sweep on the wavefront kernel (ekf_filter_wave_l96_kernel), reverse sweep on one
    or two wavefronts per trajectory (ekf_adjoint_wave_l96_kernel / ekf_adjoint_wave2_l96_kernel, round 4; the latter the default) or on
    the workgroup kernel -- against the
    oracle's discrete adjoint (value_and_grad of marginal_log_prob, ssm_temissions.py:550-568); then with half of the components
    observed (d = 40, m = 20), the drift block alone, and in fp32."""
    monkeypatch.setenv("CDKF_WAVE40_ADJ", "0" if sweep == "ekf_adjoint_wg_kernel" else "1")
    monkeypatch.setenv("CDKF_WAVE40_ADJ_WAVES", "1" if sweep == "ekf_adjoint_wave_l96_kernel" else "2")
    rng = np.random.default_rng(440)
    for m in (40, 20):
        mdl = lorenz96_model(40, m)
        N, T = 3, 6
        t = o.irregular_times(rng, N, T, 0.012 * T)
        t[:, 4:] += 0.035
        y = o.simulate(mdl, t, rng)
        P = params_from(mdl)
        ll_ref, g_ref, ex = o.ekf_loglik_grad_adjoint(mdl, t, y, full=True, state_order="second")
        ll, g = cd.cdnlgssm_loglik_and_grad_all(P, y, t[..., None])
        assert _ffi.lib().cdkf_last_kernel().decode().startswith(sweep + "<double")
        np.testing.assert_allclose(ll, ll_ref, rtol=1e-10)

        def close(a, b, name, tol=1e-8):
            scale = np.abs(b).max() + 1e-300
            assert np.abs(np.asarray(a) - b).max() < tol * scale, (name, m, np.abs(np.asarray(a) - b).max() / scale)

        close(np.asarray(g.dynamics.drift.forcing if hasattr(g.dynamics.drift, "forcing") else g.dynamics.drift[0]).reshape(N, -1), g_ref, "forcing")
        close(g.initial.mean.params, ex["m0"], "m0")
        close(g.initial.cov.params, ex["P0"], "P0")
        close(g.dynamics.diffusion_cov.params, ex["Qc"], "Qc")
        close(g.emissions.emission_function.weights, ex["H"], "H")
        close(g.emissions.emission_cov.params, ex["R"], "R")
        ll2, gd = cd.cdnlgssm_loglik_and_grad(P, y, t[..., None])
        np.testing.assert_allclose(ll2, ll_ref, rtol=1e-10)
        close(np.asarray(gd[0]).reshape(N, -1), g_ref, "forcing (drift block)")
        ll32, g32 = cd.cdnlgssm_loglik_and_grad(P, y.astype(np.float32), t[..., None])
        assert _ffi.lib().cdkf_last_kernel().decode().startswith(sweep + "<float")
        close(np.asarray(g32[0]).reshape(N, -1), g_ref, "forcing (fp32)", 2e-3)


@pytest.mark.parametrize("drift", ["lorenz96", "linear"])
def test_reverse_sweep_with_a_scattered_selection_of_observed_components(hip_lib, drift, monkeypatch):
    """The update's adjoint takes the products with H as copies when the emission picks state components -- any subset, in any order
    (detected in the kernel): Lorenz-96 d = 16 and a linear drift d = 11 observed through rows 5, 2, 11 (or 9), 0, 7 of the identity, dense R
    and P0, every leaf against the oracle; a bias or a doubled row sends the same model down the dense products, with the same answer."""
    rng = np.random.default_rng(661)
    d = 16 if drift == "lorenz96" else 11
    rows = [5, 2, 11 if d > 11 else 9, 0, 7]
    m = len(rows)
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    Rm = rng.standard_normal((m, m)) / np.sqrt(m)
    base = lorenz96_model(d, m) if drift == "lorenz96" else linear_model(rng, d, m)
    for variant in ("selection", "bias", "dense"):
        H = np.eye(d)[rows]
        bias = np.zeros(m)
        if variant == "bias":
            bias[2] = 0.3
        if variant == "dense":
            H[1, 3] = 0.5
        RR = 0.5 * np.eye(m) + 0.1 * Rm @ Rm.T
        RR = 0.5 * (RR + RR.T)   # (symmetric to the last bit: the wavefront sweeps' gate compares R with its transpose)
        mdl = o.Model(base.drift, np.eye(d), 0.4 * np.eye(d) + 0.1 * A @ A.T, H, bias, RR, base.m0, 0.6 * np.eye(d) + 0.2 * A.T @ A)
        N, T = 3, 6
        t = o.irregular_times(rng, N, T, 0.02 * T)
        y = o.simulate(mdl, t, rng)
        ll_ref, g_ref, ex = o.ekf_loglik_grad_adjoint(mdl, t, y, full=True, state_order="first")
        ll, g = cd.cdnlgssm_loglik_and_grad_all(params_from(mdl), y, t[..., None], cd.EKFHyperParams(state_order="first"))
        # (round 4: Lorenz-96 through a selection of components takes the wavefront-per-trajectory reverse sweep)
        want_kernel = "ekf_adjoint_wave2_l96_kernel<double" if (drift == "lorenz96" and variant == "selection") else "ekf_adjoint_wg_kernel<double"
        assert _ffi.lib().cdkf_last_kernel().decode().startswith(want_kernel)
        np.testing.assert_allclose(ll, ll_ref, rtol=1e-10)
        flat = np.concatenate([np.asarray(a_).reshape(N, -1) for a_ in g.dynamics.drift], axis=-1)
        for got, want in ((flat, g_ref), (g.initial.mean.params, ex["m0"]), (g.initial.cov.params, ex["P0"]), (g.dynamics.diffusion_cov.params, ex["Qc"]),
                          (g.emissions.emission_function.weights, ex["H"]), (g.emissions.emission_function.bias, ex["bias"]),
                          (g.emissions.emission_cov.params, ex["R"])):
            assert np.abs(np.asarray(got) - want).max() < 1e-8 * (np.abs(want).max() + 1e-300), variant


def test_lorenz63_gradient_all_parameters_on_the_lane_grid(hip_lib, tmp_path):
    """Small Lorenz-63 batches with H = I: the reverse sweep on the sixteen-lane grid (grad_lpe_l63_kernel<..., true>) also returns
    the model block -- m0, P0, L, Qc, H, bias, R -- of jax.grad(marginal_log_prob) (ssm_temissions.py:550-568 differentiates every
    trainable leaf).  Against the oracle's discrete adjoint (FD-pinned) with dense L, Qc, R, P0: one and several Runge-Kutta steps
    per interval, N not a multiple of four, T = 1; and against the wavefront-per-trajectory reverse sweep (CDKF_NO_LPE_GRAD=1 in a
    child process)."""
    import os, subprocess, sys
    rng = np.random.default_rng(2718)
    base = o.lorenz63_model(3)
    A, B, C = rng.standard_normal((3, 3)), rng.standard_normal((3, 3)), rng.standard_normal((3, 3))
    Rm = B @ B.T / 3 + 0.4 * np.eye(3)
    Rm = 0.5 * (Rm + Rm.T)  # bitwise symmetric: the in-grid update's condition
    mdl = o.Model(base.drift, np.eye(3) + 0.2 * rng.standard_normal((3, 3)), A @ A.T / 3 + 0.3 * np.eye(3), np.eye(3), np.zeros(3),
                  Rm, np.array([1.0, -1.5, 18.0]), C @ C.T / 3 + 0.5 * np.eye(3))
    P = params_from(mdl)
    hyp = cd.EKFHyperParams(state_order="first")

    def leaves(g, N):
        return {"drift": np.concatenate([np.asarray(a).reshape(N, -1) for a in g.dynamics.drift], axis=-1),
                "m0": g.initial.mean.params, "P0": g.initial.cov.params, "L": g.dynamics.diffusion_coefficient.params,
                "Qc": g.dynamics.diffusion_cov.params, "H": g.emissions.emission_function.weights,
                "bias": g.emissions.emission_function.bias, "R": g.emissions.emission_cov.params}

    # (total time spans: about one, seven and eighty steps of dt0 = 0.01 per interval -- the last beyond the 64 step starts kept in LDS)
    for N, T, span in ((6, 14, 0.1), (5, 9, 0.6), (3, 1, 0.01), (2, 4, 2.4)):
        t = o.irregular_times(rng, N, T, span)
        y = o.simulate(mdl, t, rng)
        ll_ref, g_ref, ex = o.ekf_loglik_grad_adjoint(mdl, t, y, full=True)
        ll, g = cd.cdnlgssm_loglik_and_grad_all(P, y, t[..., None], hyp)
        assert hip_lib.cdkf_last_kernel().startswith(b"grad_lpe_l63_kernel<double, 3, true, true>"), hip_lib.cdkf_last_kernel()
        np.testing.assert_allclose(ll, ll_ref, rtol=1e-11)
        got = leaves(g, N)
        ex = dict(ex, drift=g_ref)
        for name in ("drift", "m0", "P0", "L", "Qc", "H", "bias", "R"):
            scale = np.abs(ex[name]).max() + 1e-300
            assert np.abs(np.asarray(got[name]) - ex[name]).max() < 1e-9 * scale, (name, N, T)
    # more trajectories than one wavefront per SIMD: still this kernel for the model block (the alternative is 40x slower)
    N, T = 4400, 4
    t = o.irregular_times(rng, N, T, 0.02)
    y = o.simulate(mdl, t, rng)
    ll, g = cd.cdnlgssm_loglik_and_grad_all(P, y, t[..., None], hyp)
    assert hip_lib.cdkf_last_kernel().startswith(b"grad_lpe_l63_kernel<double, 3, true, true>"), hip_lib.cdkf_last_kernel()
    sub = np.array([0, 1234, 4097, 4399])
    ll_ref, g_ref, ex = o.ekf_loglik_grad_adjoint(mdl, t[sub], y[sub], full=True)
    got = leaves(g, N)
    np.testing.assert_allclose(ll[sub], ll_ref, rtol=1e-11)
    for name in ("m0", "P0", "L", "Qc", "H", "bias", "R"):
        assert np.abs(np.asarray(got[name])[sub] - ex[name]).max() < 1e-9 * (np.abs(ex[name]).max() + 1e-300), name
    assert np.abs(got["drift"][sub] - g_ref).max() < 1e-9 * np.abs(g_ref).max()
    # fp32 and the other kernel on the last batch but one
    t = o.irregular_times(rng, 7, 20, 0.5)
    y = o.simulate(mdl, t, rng)
    ll_ref, g_ref, ex = o.ekf_loglik_grad_adjoint(mdl, t, y, full=True)
    ll32, g32 = cd.cdnlgssm_loglik_and_grad_all(P, y.astype(np.float32), t[..., None].astype(np.float32), hyp)
    assert hip_lib.cdkf_last_kernel().startswith(b"grad_lpe_l63_kernel<float, 3, true, true>")
    got32 = leaves(g32, 7)
    for name in ("m0", "P0", "Qc", "H", "bias", "R"):
        assert np.abs(np.asarray(got32[name]) - ex[name]).max() < 2e-2 * np.abs(ex[name]).max(), name
    ll, g = cd.cdnlgssm_loglik_and_grad_all(P, y, t[..., None], hyp)
    got = leaves(g, 7)
    np.savez(tmp_path / "in.npz", t=t, y=y, L=mdl.L, Qc=mdl.Qc, R=mdl.R, m0=mdl.m0, P0=mdl.P0)
    code = ("import sys, numpy as np; sys.path[:0] = [%r, %r, %r]\n"
            "import cd_dynamax_amd as cd, cdkf_oracle as o\nfrom cd_dynamax_amd import _ffi\nfrom helpers import params_from\n"
            "d = np.load(%r); mdl = o.Model(o.lorenz63_model(3).drift, d['L'], d['Qc'], np.eye(3), np.zeros(3), d['R'], d['m0'], d['P0'])\n"
            "ll, g = cd.cdnlgssm_loglik_and_grad_all(params_from(mdl), d['y'], d['t'][..., None], cd.EKFHyperParams(state_order='first'))\n"
            "assert _ffi.lib().cdkf_last_kernel().startswith(b'ekf_adjoint_wave8_kernel'), _ffi.lib().cdkf_last_kernel()\n"
            "np.savez(%r, ll=ll, H=g.emissions.emission_function.weights, R=g.emissions.emission_cov.params, P0=g.initial.cov.params,\n"
            "         L=g.dynamics.diffusion_coefficient.params, rho=g.dynamics.drift.rho)\n"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(o.__file__)),
               os.path.dirname(os.path.abspath(__file__)), str(tmp_path / "in.npz"), str(tmp_path / "out.npz")))
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CDKF_NO_LPE_GRAD="1"), check=True, timeout=600)
    other = np.load(tmp_path / "out.npz")
    assert relerr(ll, other["ll"]) < 1e-12
    for name in ("H", "R", "P0", "L"):
        assert np.abs(np.asarray(got[name]) - other[name]).max() < 1e-9 * np.abs(other[name]).max(), name
    assert np.abs(np.asarray(g.dynamics.drift.rho) - other["rho"]).max() < 1e-9 * np.abs(other["rho"]).max()


@pytest.mark.parametrize("m", [1, 2])
def test_lorenz63_gradient_on_the_lane_grid_partial_observations(hip_lib, m):
    """The same reverse sweep with H = I[:m], m < 3 (SURVEY section 8d's H = [1, 0, 0] case): the reverse update then runs per lane
    (lpe_update_adj_gen).  Drift block against the oracle's forward sensitivities, every leaf against its discrete adjoint."""
    rng = np.random.default_rng(100 + m)
    base = o.lorenz63_model(m)
    A, B, C = rng.standard_normal((3, 3)), rng.standard_normal((m, m)), rng.standard_normal((3, 3))
    Rm = B @ B.T / m + 0.5 * np.eye(m)
    Rm = 0.5 * (Rm + Rm.T)
    mdl = o.Model(base.drift, np.eye(3) + 0.2 * rng.standard_normal((3, 3)), A @ A.T / 3 + 0.3 * np.eye(3), np.eye(3)[:m], np.zeros(m),
                  Rm, np.array([1.0, -1.5, 18.0]), C @ C.T / 3 + 0.5 * np.eye(3))
    P = params_from(mdl)
    hyp = cd.EKFHyperParams(state_order="first")
    name = b"grad_lpe_l63_kernel<double, %d, false, " % m
    for N, T, span in ((7, 25, 0.2), (5, 8, 0.8), (2, 1, 0.1)):
        t = o.irregular_times(rng, N, T, span)
        y = o.simulate(mdl, t, rng)
        ll_ref, g_ref = o.ekf_loglik_grad(mdl, t, y)
        ll, g = cd.cdnlgssm_loglik_and_grad(P, y, t[..., None])
        assert hip_lib.cdkf_last_kernel().startswith(name + b"false>"), hip_lib.cdkf_last_kernel()
        np.testing.assert_allclose(ll, ll_ref, rtol=1e-11)
        gd = np.stack([g.sigma, g.rho, g.beta], -1)
        assert np.abs(gd - g_ref).max() <= 1e-9 * np.abs(g_ref).max()  # (T = 1: no predict, the drift block is exactly zero)
        ll_ref, g_ref, ex = o.ekf_loglik_grad_adjoint(mdl, t, y, full=True)
        ll, g = cd.cdnlgssm_loglik_and_grad_all(P, y, t[..., None], hyp)
        assert hip_lib.cdkf_last_kernel().startswith(name + b"true>"), hip_lib.cdkf_last_kernel()
        got = {"drift": np.concatenate([np.asarray(a).reshape(N, -1) for a in g.dynamics.drift], axis=-1),
               "m0": g.initial.mean.params, "P0": g.initial.cov.params, "L": g.dynamics.diffusion_coefficient.params,
               "Qc": g.dynamics.diffusion_cov.params, "H": g.emissions.emission_function.weights,
               "bias": g.emissions.emission_function.bias, "R": g.emissions.emission_cov.params}
        ex = dict(ex, drift=g_ref)
        for key in got:
            assert np.abs(np.asarray(got[key]) - ex[key]).max() <= 1e-9 * np.abs(ex[key]).max(), (key, N, T)


@pytest.mark.parametrize("solver", ["tsit5", "heun", "euler"])
def test_workgroup_kernels_other_runge_kutta_methods(hip_lib, solver):
    """diffeqsolve_settings={'solver': ...} beyond the register-resident shapes: Lorenz-96 d = 12 (workgroup kernels) and
    d = 6 / an MLP drift at d = 5 (state_dim <= 8 takes the workgroup kernels for non-default methods), EKF / UKF / smoother."""
    rng = np.random.default_rng(17)
    settings = {"solver": solver}
    for mdl in (lorenz96_model(12, 5), lorenz96_model(6, 6), mlp_model(rng, 5, 2, (9, 7))):
        N, T = 3, 8
        t = o.irregular_times(rng, N, T, 0.03)
        y = o.simulate(mdl, t, rng)
        P = params_from(mdl)
        with o.use_solver(solver):
            ref = o.ekf_filter(mdl, t, y, state_order="first")
            refu = o.ukf_filter(mdl, t, y)
            refs = o.ekf_smoother(mdl, t, y, state_order="first")
        post = cd.cdnlgssm_filter(P, y, t[..., None], cd.EKFHyperParams(state_order="first", diffeqsolve_settings=settings))
        for k in FILTER_KEYS:
            assert relerr(getattr(post, k), ref[k]) < 1e-10, k
        postu = cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams(diffeqsolve_settings=settings))
        assert relerr(postu.filtered_covariances, refu["filtered_covariances"]) < 1e-9
        sm = cd.cdnlgssm_smoother(P, y, t[..., None], cd.EKFHyperParams(state_order="first", diffeqsolve_settings=settings))
        assert relerr(sm.smoothed_covariances, refs["smoothed_covariances"]) < 1e-9


@pytest.mark.parametrize("solver,ctrl", [("dopri5", dict(rtol=1e-6, atol=1e-8)), ("tsit5", dict(rtol=1e-5, atol=1e-7, pcoeff=0.3, icoeff=0.4)),
                                          ("bosh3", dict(rtol=1e-4, atol=1e-6))])
def test_workgroup_kernels_adaptive_steps(hip_lib, solver, ctrl):
    """diffeqsolve_settings={'stepsize_controller': PIDController(...)} beyond the register-resident shapes: the workgroup kernels
    form the embedded error estimate's RMS over the mean and the full covariance across the workgroup, so every thread takes the
    same accept / reject decision.  Lorenz-96 d = 12 and d = 40 (the wavefront kernels hand adaptive solves over), an MLP at d = 5:
    EKF, UKF and smoother against the oracle's restatement of the controller; dt0 far above what the tolerance allows."""
    rng = np.random.default_rng(31)
    settings = {"solver": solver, "dt0": 0.05, "stepsize_controller": cd.PIDController(**ctrl)}
    for mdl, (N, T) in ((lorenz96_model(12, 5), (3, 8)), (mlp_model(rng, 5, 2, (9, 7)), (3, 8)), (lorenz96_model(40, 40), (2, 4))):
        t = o.irregular_times(rng, N, T, 0.04)
        t[:, T // 2:] += 0.15
        y = o.simulate(mdl, t, rng)
        P = params_from(mdl)
        with o.use_solver(solver, adaptive=ctrl):
            ref = o.ekf_filter(mdl, t, y, dt0=0.05, state_order="first")
            refs = o.ekf_smoother(mdl, t, y, dt0=0.05, state_order="first")
            refu = o.ukf_filter(mdl, t, y, dt0=0.05) if mdl.d < 40 else None
        post = cd.cdnlgssm_filter(P, y, t[..., None], cd.EKFHyperParams(state_order="first", diffeqsolve_settings=settings))
        assert _ffi.lib().cdkf_last_kernel().decode().startswith("ekf_filter_wg_kernel")
        for k in FILTER_KEYS:
            assert relerr(getattr(post, k), ref[k]) < 1e-9, k
        np.testing.assert_allclose(post.marginal_loglik, ref["marginal_loglik"], rtol=1e-9)
        sm = cd.cdnlgssm_smoother(P, y, t[..., None], cd.EKFHyperParams(state_order="first", diffeqsolve_settings=settings))
        assert relerr(sm.smoothed_covariances, refs["smoothed_covariances"]) < 1e-8
        if refu is not None:
            postu = cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams(diffeqsolve_settings=settings))
            assert relerr(postu.filtered_covariances, refu["filtered_covariances"]) < 1e-8
    # a step budget that cannot be met raises the MAX_STEPS status
    from cd_dynamax_amd import models
    mb = models._model_block(P)
    op = models._opts(cd.EKFHyperParams(diffeqsolve_settings=dict(settings, max_steps=2)), 1)
    _, _, status = _ffi.run_host("ekf_filter", mb, op, t, y, [False] * 4, np.float64)
    assert (status & 4).any()


@pytest.mark.parametrize("d,m,h", [(9, 2, (3, 50)), (10, 8, (17, 33)), (6, 9, (20, 64)), (9, 9, (33, 64))])
def test_mlp_drift_on_the_workgroup_kernels_second_layer_wider_than_the_first(hip_lib, d, m, h):
    """MLP drift beyond the wavefront kernel's shapes with hidden sizes h1 < h2: the tangent image T [h2 x d] was given d * h1 reals
    of LDS, and the tail of the arrays behind it overwrote the copy of the weights (1-3 % errors in every output; every earlier
    test had h1 >= h2; found by scripts/gpu_fuzz_filters.py).  EKF both orders, smoother and UKF against the oracle."""
    rng = np.random.default_rng(d * 10 + m)
    mdl = mlp_model(rng, d, min(m, d), h)
    if m > d:
        mdl = o.Model(mdl.drift, mdl.L, mdl.Qc, rng.standard_normal((m, d)) / np.sqrt(d), np.zeros(m), 0.5 * np.eye(m), mdl.m0, mdl.P0)
    N, T = 3, 7
    t = o.irregular_times(rng, N, T, 0.012 * T)
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    for order in ("first", "second"):
        ref = o.ekf_smoother(mdl, t, y, state_order=order)
        post = cd.cdnlgssm_smoother(P, y, t[..., None], cd.EKFHyperParams(state_order=order))
        assert _ffi.lib().cdkf_last_kernel().decode().startswith("ekf_smoother_wg_kernel<double")
        for k in ("filtered_means", "filtered_covariances", "smoothed_means", "smoothed_covariances"):
            assert relerr(getattr(post, k), ref[k]) < 1e-10, (order, k)
    refu = o.ukf_filter(mdl, t, y)
    pu = cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams())
    assert relerr(pu.filtered_means, refu["filtered_means"]) < 1e-10


@pytest.mark.parametrize("d,m,h", [(5, 2, (9, 7)), (8, 4, (64, 64)), (2, 1, (1, 1))])
def test_mlp_second_order_in_float32_on_the_wavefront_kernel(hip_lib, d, m, h):
    """state_order='second' in fp32 on ekf_filter_wave8_kernel<float>: the mean's 0.5 P grad(div f) is summed over the lane grid with
    v_permlane16_swap / v_permlane32_swap -- through the compiler builtins the fp32 sums came out as twice one half (r0 + r0) and the
    sweep was off by 1e-2 from the fp64 oracle, where the fp32 oracle is within 2e-7 (found by scripts/gpu_fuzz_filters.py; the
    swaps are inline assembly since).  Both orders within 5e-6."""
    rng = np.random.default_rng(d + 70)
    mdl = mlp_model(rng, d, m, h)
    N, T = 3, 8
    t = o.irregular_times(rng, N, T, 0.012 * T)
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    for order in ("first", "second"):
        ref = o.ekf_filter(mdl, t, y, state_order=order)
        p32 = cd.cdnlgssm_filter(P, y.astype(np.float32), t[..., None].astype(np.float32), cd.EKFHyperParams(state_order=order))
        # (fp32 runs the two-wavefront mapping by default, cdkf_wave8s_kernels.h: the same lane-grid sums)
        assert _ffi.lib().cdkf_last_kernel().decode().startswith(("ekf_filter_wave8_kernel<float>", "ekf_filter_wave8s_kernel<float"))
        assert relerr(p32.filtered_means, ref["filtered_means"]) < 5e-6, order
        assert relerr(p32.predicted_covariances, ref["predicted_covariances"]) < 5e-6, order
        np.testing.assert_allclose(p32.marginal_loglik, ref["marginal_loglik"], rtol=2e-5)


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

W8_SPLIT_WORKER = r'''
import os, sys
sys.path[:0] = [os.environ["CDKF_ROOT"], os.path.join(os.environ["CDKF_ROOT"], "oracle"), os.path.join(os.environ["CDKF_ROOT"], "tests")]
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi
from helpers import FILTER_KEYS, mlp_model, params_from, relerr
want = "wave8s" if os.environ["CDKF_W8_SPLIT"] == "2" else "wave8_kernel"
worst = {}
for d, m, h in ((8, 4, (64, 64)), (5, 2, (24, 40)), (8, 8, (64, 17))):
    rng = np.random.default_rng(100 + d + m)
    mdl = mlp_model(rng, d, m, h)
    N, T = 6, 40
    t = o.irregular_times(rng, N, T, 0.006 * T)
    t[:, 30:] += 0.07          # one interval of eight Dormand-Prince steps
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    for order in ("first", "second"):
        ref = o.ekf_filter(mdl, t, y, state_order=order)
        for dtype, tol in ((np.float64, 1e-9), (np.float32, 1e-5)):
            post = cd.cdnlgssm_filter(P, y.astype(dtype), t[..., None], cd.EKFHyperParams(state_order=order))
            assert want in _ffi.lib().cdkf_last_kernel().decode(), _ffi.lib().cdkf_last_kernel().decode()
            errs = [relerr(getattr(post, k), ref[k]) for k in FILTER_KEYS]
            errs.append(float(np.max(np.abs(np.asarray(post.marginal_loglik, np.float64) - ref["marginal_loglik"]) / np.abs(ref["marginal_loglik"]))))
            worst[dtype.__name__] = max(worst.get(dtype.__name__, 0.0), max(errs))
            assert max(errs) < tol, (d, m, h, order, dtype.__name__, errs)
        # the same sweep as the forward pass of the reverse sweep (it leaves the stage checkpoints the adjoint starts from)
        ll_ref, g_ref = o.ekf_loglik_grad_adjoint(mdl, t, y, state_order=order)
        ll, g = cd.cdnlgssm_loglik_and_grad(P, y, t[..., None], cd.EKFHyperParams(state_order=order))
        flat = np.concatenate([np.asarray(a).reshape(N, -1) for a in g], axis=-1)
        assert np.abs(flat - g_ref).max() < 1e-8 * np.abs(g_ref).max(), (d, m, h, order)
        np.testing.assert_allclose(ll, ll_ref, rtol=1e-10)
        ll32, g32 = cd.cdnlgssm_loglik_and_grad(P, y.astype(np.float32), t[..., None].astype(np.float32), cd.EKFHyperParams(state_order=order))
        flat32 = np.concatenate([np.asarray(a).reshape(N, -1) for a in g32], axis=-1)
        assert np.abs(flat32 - g_ref).max() < 2e-2 * np.abs(g_ref).max(), (d, m, h, order)
sys.stdout.write("SPLIT_OK %s %s\n" % (os.environ["CDKF_W8_SPLIT"], worst))
'''


@pytest.mark.parametrize("split", ["0", "2"])
def test_mlp_sweep_on_one_and_on_two_wavefronts_per_trajectory(hip_lib, tmp_path, split):
    """The MLP-drift sweep has two mappings -- one wavefront per trajectory (cdkf_wave8_kernels.h) and one trajectory over two
    wavefronts (cdkf_wave8s_kernels.h; the default in fp32) -- and the library picks by precision.  Both are held to the oracle in BOTH
    precisions here (CDKF_W8_SPLIT forces one; the library reads it once, hence the child process): full and ragged hidden sizes, m = d,
    both state orders, an eight-step interval, and as the forward pass of the reverse sweep (value and every weight gradient).
    Reference: inference_ekf.py:46-148, 202-326; ssm_temissions.py:550-568."""
    import subprocess, sys
    script = tmp_path / "w8_split_worker.py"
    script.write_text(W8_SPLIT_WORKER)
    env = dict(os.environ, CDKF_ROOT=ROOT, CDKF_W8_SPLIT=split)
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and f"SPLIT_OK {split}" in p.stdout, p.stdout + p.stderr


@pytest.mark.parametrize("d,rows", [(12, [3, 0, 7]), (20, list(range(20))), (28, [1, 5, 9, 13, 17, 21, 25, 27, 2, 6]), (36, list(range(0, 36, 2)))])
def test_wavefront_reverse_sweep_of_lorenz96_at_every_instantiated_width(hip_lib, d, rows, monkeypatch):
    """ekf_adjoint_wave_l96_kernel (round 4: Lorenz-96 through a selection of components, one wavefront per trajectory) at state
    dimensions whose 16-wide tile grids differ (one, two, three tiles; last tile of 4, 12, 16 columns), with dense L Qc L^T, R, P0, a
    scattered selection, an interval of 13 Runge-Kutta steps (two replay chunks of eight starts) and one of zero length: every leaf
    against the oracle's discrete adjoint, and against the workgroup reverse sweep on the same inputs."""
    monkeypatch.setenv("CDKF_WAVE40_ADJ", "1")
    rng = np.random.default_rng(900 + d)
    m = len(rows)
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    Rm = rng.standard_normal((m, m)) / np.sqrt(m)
    RR = 0.5 * np.eye(m) + 0.1 * Rm @ Rm.T
    RR = 0.5 * (RR + RR.T)
    base = lorenz96_model(d, m)
    mdl = o.Model(base.drift, np.eye(d), 0.4 * np.eye(d) + 0.1 * A @ A.T, np.eye(d)[rows], np.zeros(m), RR, base.m0 + 0.1 * rng.standard_normal(d),
                  0.6 * np.eye(d) + 0.2 * A.T @ A)
    N, T = 3, 7
    t = o.irregular_times(rng, N, T, 0.006 * T)
    t[:, 3:] += 0.125          # thirteen steps of 0.01
    t[:, 5] = t[:, 4]          # an interval of zero length
    t[:, 6] = t[:, 5] + 0.004
    y = o.simulate(mdl, t, rng)
    hyp = cd.EKFHyperParams(state_order="first")
    ll_ref, g_ref, ex = o.ekf_loglik_grad_adjoint(mdl, t, y, full=True, state_order="first")
    got = {}
    for which in ("2", "1", "0"):   # two wavefronts per trajectory (the default), one, the workgroup kernel
        monkeypatch.setenv("CDKF_WAVE40_ADJ", "0" if which == "0" else "1")
        monkeypatch.setenv("CDKF_WAVE40_ADJ_WAVES", which if which != "0" else "2")
        ll, g = cd.cdnlgssm_loglik_and_grad_all(params_from(mdl), y, t[..., None], hyp)
        kern = _ffi.lib().cdkf_last_kernel().decode()
        want = {"2": "ekf_adjoint_wave2_l96_kernel<double, %d>" % d, "1": "ekf_adjoint_wave_l96_kernel<double, %d>" % d, "0": "ekf_adjoint_wg_kernel<double"}[which]
        assert kern.startswith(want), kern
        np.testing.assert_allclose(ll, ll_ref, rtol=1e-10)
        flat = np.concatenate([np.asarray(a_).reshape(N, -1) for a_ in g.dynamics.drift], axis=-1)
        leaves = [flat, g.initial.mean.params, g.initial.cov.params, g.dynamics.diffusion_cov.params, g.emissions.emission_function.weights,
                  g.emissions.emission_function.bias, g.emissions.emission_cov.params]
        for a_, b_, name in zip(leaves, (g_ref, ex["m0"], ex["P0"], ex["Qc"], ex["H"], ex["bias"], ex["R"]), ("forcing", "m0", "P0", "Qc", "H", "bias", "R")):
            scale = np.abs(b_).max() + 1e-300
            assert np.abs(np.asarray(a_) - b_).max() < 1e-8 * scale, (which, name, np.abs(np.asarray(a_) - b_).max() / scale)
        got[which] = [np.asarray(a_) for a_ in leaves]
    for w_ in ("2", "1"):
        for a_, b_ in zip(got[w_], got["0"]):
            assert np.abs(a_ - b_).max() <= 1e-9 * (np.abs(b_).max() + 1e-300)
    # fp32: the forcing's gradient at single-precision accuracy
    monkeypatch.setenv("CDKF_WAVE40_ADJ", "1")
    for w_ in ("1", "2"):
        monkeypatch.setenv("CDKF_WAVE40_ADJ_WAVES", w_)
        ll32, g32 = cd.cdnlgssm_loglik_and_grad(params_from(mdl), y.astype(np.float32), t[..., None], hyp)
        assert _ffi.lib().cdkf_last_kernel().decode().startswith("ekf_adjoint_wave%s_l96_kernel<float, %d>" % ("2" if w_ == "2" else "", d))
        assert np.abs(np.asarray(g32[0]).reshape(N, -1) - g_ref).max() < 5e-3 * np.abs(g_ref).max()


def test_lorenz96_filter_with_one_factorisation_for_a_diagonal_R(hip_lib, monkeypatch):
    """Round 4: with a diagonal R (entries >= 1e-2) the wavefront Lorenz-96 filter factors S + 1e-9 I only and takes the log-likelihood's
    log det S and v^T S^-1 v by first-order corrections in 1e-9 (ekf_filter_wave_l96_kernel<R, D, true>): the same log-likelihood as
    the two factorisations in lockstep to 1e-12, the same moments bitwise (the gain never used the other factor), the oracle's numbers --
    full and partial observation; a dense or a tiny R keeps the two systems."""
    rng = np.random.default_rng(31)
    d = 40
    for rows, rdiag in ((list(range(d)), None), ([3, 17, 30, 8, 39, 21], np.array([0.5, 2.0, 0.02, 1.0, 0.3, 4.0]))):
        m = len(rows)
        base = lorenz96_model(d, m)
        Rm = np.eye(m) if rdiag is None else np.diag(rdiag)
        mdl = o.Model(base.drift, base.L, base.Qc, np.eye(d)[rows], np.zeros(m), Rm, base.m0, base.P0)
        N, T = 3, 12
        t = o.irregular_times(rng, N, T, 0.01 * T)
        y = o.simulate(mdl, t, rng)
        P = params_from(mdl)
        ref = o.ekf_filter(mdl, t, y)
        out = {}
        for two in (False, True):
            if two:
                monkeypatch.setenv("CDKF_W40_TWO_FACTORS", "1")
            else:
                monkeypatch.delenv("CDKF_W40_TWO_FACTORS", raising=False)
            post = cd.cdnlgssm_filter(P, y, t[..., None], cd.EKFHyperParams())
            kern = _ffi.lib().cdkf_last_kernel().decode()
            assert kern == ("ekf_filter_wave_l96_kernel<double, 40>" if two else "ekf_filter_wave_l96_kernel<double, 40, true>"), kern
            out[two] = post
        assert relerr(out[False].marginal_loglik, out[True].marginal_loglik) < 1e-12
        assert relerr(out[False].marginal_loglik, ref["marginal_loglik"]) < 1e-11
        for k in ("filtered_means", "filtered_covariances", "predicted_means", "predicted_covariances"):
            assert np.array_equal(np.asarray(getattr(out[False], k)), np.asarray(getattr(out[True], k))), k
            assert relerr(getattr(out[False], k), ref[k]) < 1e-9, k
    monkeypatch.delenv("CDKF_W40_TWO_FACTORS", raising=False)
    for Rm in (np.diag(np.r_[1e-3, np.ones(d - 1)]), np.eye(d) + 0.01 * (np.eye(d, k=1) + np.eye(d, k=-1))):   # a tiny entry; off-diagonal entries
        base = lorenz96_model(d, d)
        mdl = o.Model(base.drift, base.L, base.Qc, np.eye(d), np.zeros(d), Rm, base.m0, base.P0)
        t = o.irregular_times(rng, 2, 4, 0.04)
        y = o.simulate(mdl, t, rng)
        post = cd.cdnlgssm_filter(params_from(mdl), y, t[..., None], cd.EKFHyperParams())
        assert _ffi.lib().cdkf_last_kernel().decode() == "ekf_filter_wave_l96_kernel<double, 40>"
        assert relerr(post.marginal_loglik, o.ekf_filter(mdl, t, y)["marginal_loglik"]) < 1e-10
