"""Randomised parity soak through the C ABI: shapes, layouts, precisions, orders, shared / per-trajectory time grids, edge sizes
(T = 1, N = 1, N just above a wavefront), every kernel family (register, wavefront, workgroup)."""
import ctypes as C

import numpy as np
import pytest

import cd_dynamax_amd as cd
import cdkf_oracle as o
from cd_dynamax_amd import _ffi, models
from helpers import linear_model, lorenz96_model, mlp_model, params_from, relerr

pytestmark = pytest.mark.gpu


def _run(algo, mdl, opts, layout, t, y, dtype, want=(True,) * 4):
    """Call the host entry point in an explicit layout and return arrays in the reference shapes."""
    N, T, m = y.shape
    d = mdl.d
    blk = models._model_block(params_from(mdl))
    opts.layout = layout
    tt = np.asarray(t, dtype)
    if layout == _ffi.LAYOUT_NT:
        y_l = np.ascontiguousarray(y, dtype)
        t_l = np.ascontiguousarray(tt)
        shp = [(N, T, d), (N, T, d, d)] * 2
        back = lambda a: a
    elif layout == _ffi.LAYOUT_TN:
        y_l = np.ascontiguousarray(np.asarray(y, dtype).transpose(1, 0, 2))
        t_l = np.ascontiguousarray(tt if opts.t_shared else tt.T)
        shp = [(T, N, d), (T, N, d, d)] * 2
        back = lambda a: np.swapaxes(a, 0, 1)
    else:
        y_l = np.ascontiguousarray(np.asarray(y, dtype).transpose(1, 2, 0))
        t_l = np.ascontiguousarray(tt if opts.t_shared else tt.T)
        shp = [(T, d, N), (T, d, d, N)] * 2
        back = lambda a: np.moveaxis(a, -1, 0)
    ll = np.empty(N, dtype)
    st = np.zeros(N, np.int32)
    outs = [np.full(s, np.nan, dtype) if w else None for s, w in zip(shp, want)]
    vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    fn = getattr(_ffi.lib(), f"cdkf_{algo}_{'f32' if dtype == np.float32 else 'f64'}")
    _ffi.check(fn(C.byref(blk.c), C.byref(opts), N, T, vp(t_l), vp(y_l), vp(ll), *[vp(a) for a in outs], vp(st)))
    return ll, [None if a is None else back(a) for a in outs], st


CASES = [
    # (model factory, N, T)
    (lambda r: o.lorenz63_model(3), 1, 1), (lambda r: o.lorenz63_model(1), 65, 2), (lambda r: o.lorenz63_model(2), 130, 7),
    (lambda r: linear_model(r, 2, 6), 3, 9), (lambda r: linear_model(r, 4, 2), 70, 5), (lambda r: linear_model(r, 1, 1), 64, 3),
    (lambda r: linear_model(r, 5, 3), 5, 6), (lambda r: lorenz96_model(6, 6), 4, 5), (lambda r: lorenz96_model(12, 5), 3, 4),
    (lambda r: mlp_model(r, 3, 2, (9, 4)), 6, 5), (lambda r: mlp_model(r, 10, 3, (8, 8)), 2, 4), (lambda r: linear_model(r, 17, 20), 2, 3),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_soak_filters_and_smoother(hip_lib, case):
    rng = np.random.default_rng(1000 + case)
    factory, N, T = CASES[case]
    mdl = factory(rng)
    shared = bool(case % 3 == 1)
    t = o.irregular_times(rng, 1 if shared else N, T, 0.05)
    tt = np.broadcast_to(t, (N, T)) if shared else t
    y = o.simulate(mdl, tt, rng)
    order = ["second", "first", "zeroth"][case % 3]
    num_iter = 1 + (case % 2)
    refs = {"ekf": o.ekf_filter(mdl, tt, y, state_order=order, num_iter=num_iter), "ukf": o.ukf_filter(mdl, tt, y),
            "eks": o.ekf_smoother(mdl, tt, y, state_order=order)}
    for layout in (_ffi.LAYOUT_NT, _ffi.LAYOUT_TN, _ffi.LAYOUT_TCN):
        for dtype, tol in ((np.float64, 2e-9), (np.float32, 3e-3)):
            opts = _ffi.default_opts()
            opts.t_shared = 1 if shared else 0
            opts.state_order = _ffi.ORDER[order]
            opts.num_iter = num_iter
            t_in = t[0] if shared else t
            ll, outs, st = _run("ekf_filter", mdl, opts, layout, t_in, y, dtype)
            assert not st.any(), (layout, dtype, st)
            for a, k in zip(outs, ("filtered_means", "filtered_covariances", "predicted_means", "predicted_covariances")):
                assert relerr(a, refs["ekf"][k]) < tol, (layout, dtype, k)
            assert relerr(ll, refs["ekf"]["marginal_loglik"]) < tol
            if dtype == np.float64:      # a subset of outputs: the others must stay untouched (NULL pointers)
                ll2, outs2, _ = _run("ekf_filter", mdl, opts, layout, t_in, y, dtype, want=(False, True, False, False))
                assert relerr(outs2[1], refs["ekf"]["filtered_covariances"]) < tol and relerr(ll2, ll) < 1e-13
                opts.num_iter = 1
                llu, outsu, _ = _run("ukf_filter", mdl, opts, layout, t_in, y, dtype)
                assert relerr(outsu[0], refs["ukf"]["filtered_means"]) < 1e-8, layout
                lls, outss, _ = _run("ekf_smoother", mdl, opts, layout, t_in, y, dtype)
                assert relerr(outss[2], refs["eks"]["smoothed_means"]) < 1e-8, layout
                assert relerr(outss[3], refs["eks"]["smoothed_covariances"]) < 1e-8, layout


@pytest.mark.parametrize("kind", ["lorenz63", "linear44", "mlp", "lorenz96_13", "linear_11_4"])
def test_soak_gradient_layouts(hip_lib, kind):
    """cdkf_ekf_loglik_grad[_all]_* honour opts.layout for t and y (forward-sensitivity and reverse-sweep kernels), shared and
    per-trajectory grids; N = 1 and N just above a wavefront."""
    rng = np.random.default_rng(7)
    mdl = {"lorenz63": lambda: o.lorenz63_model(3), "linear44": lambda: linear_model(rng, 4, 4),
           "mlp": lambda: mlp_model(rng, 5, 2, (6, 7)), "lorenz96_13": lambda: lorenz96_model(13, 6),
           "linear_11_4": lambda: linear_model(rng, 11, 4)}[kind]()
    for N, T, shared in ((1, 6, False), (66, 5, True)):
        t = o.irregular_times(rng, 1 if shared else N, T, 0.04)
        tt = np.broadcast_to(t, (N, T)) if shared else t
        y = o.simulate(mdl, tt, rng)
        ll_ref, g_ref = o.ekf_loglik_grad_adjoint(mdl, tt, y)
        blk = models._model_block(params_from(mdl))
        for layout in (_ffi.LAYOUT_NT, _ffi.LAYOUT_TN, _ffi.LAYOUT_TCN):
            opts = _ffi.default_opts()
            opts.state_order = _ffi.ORDER["first"]
            opts.t_shared = 1 if shared else 0
            opts.layout = layout
            y_l = {_ffi.LAYOUT_NT: y, _ffi.LAYOUT_TN: y.transpose(1, 0, 2), _ffi.LAYOUT_TCN: y.transpose(1, 2, 0)}[layout]
            t_l = t[0] if shared else (t if layout == _ffi.LAYOUT_NT else t.T)
            y_l, t_l = np.ascontiguousarray(y_l), np.ascontiguousarray(t_l)
            ll, g, st = np.empty(N), np.empty((N, blk.theta.size)), np.zeros(N, np.int32)
            gm = np.empty((N, _ffi.model_grad_size(mdl.d, mdl.m)))
            vp = lambda a: a.ctypes.data_as(C.c_void_p)
            _ffi.check(_ffi.lib().cdkf_ekf_loglik_grad_f64(C.byref(blk.c), C.byref(opts), N, T, vp(t_l), vp(y_l), vp(ll), vp(g), vp(st)))
            assert relerr(g, g_ref) < 1e-8 and relerr(ll, ll_ref) < 1e-10, (kind, layout, N)
            _ffi.check(_ffi.lib().cdkf_ekf_loglik_grad_all_f64(C.byref(blk.c), C.byref(opts), N, T, vp(t_l), vp(y_l), vp(ll), vp(g),
                                                               vp(gm), vp(st)))
            assert relerr(g, g_ref) < 1e-8 and np.isfinite(gm).all(), (kind, layout, N)


def test_soak_gradient_on_the_lane_grid_against_the_other_kernels(hip_lib):
    """Forty random Lorenz-63 problems (emission dimension 1-3, dense L / Qc / R / P0, 1-44 trajectories, 1-39 observations, one to a few
    hundred Runge-Kutta steps per interval): the drift block and the model block from the forward + reverse sweep on the sixteen-lane grid
    against the forward-sensitivity kernel / the wavefront-per-trajectory reverse sweep (scripts/gpu_lpe_grad_soak.py runs the cases in
    two processes, the second with CDKF_NO_LPE_GRAD=1)."""
    import os, re, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "scripts", "gpu_lpe_grad_soak.py")], capture_output=True, text=True,
                         timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "MISMATCH" not in res.stdout, res.stdout[-2000:]
    worst = [float(v) for v in re.findall(r": ([0-9.e+-]+)[,}]", res.stdout.splitlines()[-1])]
    assert len(worst) == 6 and max(worst) < 1e-9, res.stdout[-500:]


def test_soak_reverse_sweep_beyond_eight_dimensions(hip_lib):
    """Sixteen random problems for the workgroup-per-trajectory reverse sweep (ekf_adjoint_wg_kernel): Lorenz-96 and linear drifts,
    state dimension 9 .. 41, emission dimension 1 .. 41 (m > d included), dense non-diagonal L / Qc / R / P0 and a dense H with bias or a
    selection of components, 1 .. 5 trajectories, 1 .. 7 observations, intervals from zero length (repeated observation times) to a few
    dozen Runge-Kutta steps (several replay chunks), other fixed-step tableaus -- every leaf against the oracle's discrete adjoint."""
    rng = np.random.default_rng(4242)
    worst = 0.0
    for case in range(16):
        lin = case % 3 == 1
        d = int(rng.integers(9, 20)) if lin else int(rng.integers(9, 42))
        m = int(rng.integers(1, d + 1)) if case % 4 else int(min(41, d + rng.integers(1, 4)))
        drift = linear_model(rng, d, min(m, d)).drift if lin else o.Lorenz96Drift(8.0 + rng.standard_normal())
        A, B, Cc = rng.standard_normal((d, d)), rng.standard_normal((m, m)), rng.standard_normal((d, d))
        if case % 5 == 2 and m <= d:
            H, bias = np.eye(d)[rng.permutation(d)[:m]], np.zeros(m)
        else:
            H, bias = rng.standard_normal((m, d)) / np.sqrt(d), 0.1 * rng.standard_normal(m)
        mdl = o.Model(drift, np.eye(d) + 0.1 * rng.standard_normal((d, d)), A @ A.T / d + 0.3 * np.eye(d), H, bias,
                      B @ B.T / m + 0.3 * np.eye(m), (0.0 if lin else 8.0) + rng.standard_normal(d), Cc @ Cc.T / d + 0.5 * np.eye(d))
        N, T = int(rng.integers(1, 6)), int(rng.integers(1, 8))
        t = o.irregular_times(rng, N, T, 0.02 * T)
        solver = [None, None, "heun", "euler", "tsit5"][case % 5]
        if T > 3:
            t[:, 3:] += rng.uniform(0.05, 0.3) if solver is None else 0.02   # one long interval: 5 .. 30 steps (Dormand-Prince only)
            t[0, 2] = t[0, 1]                          # and a repeated observation time
        y = o.simulate(mdl, t, rng)
        hyp = cd.EKFHyperParams(state_order="second" if case % 2 else "first", diffeqsolve_settings={"solver": solver} if solver else {})
        if solver:
            with o.use_solver(solver):
                ll_ref, g_ref, ex = o.ekf_loglik_grad_adjoint(mdl, t, y, full=True)
        else:
            ll_ref, g_ref, ex = o.ekf_loglik_grad_adjoint(mdl, t, y, full=True)
        ll, g = cd.cdnlgssm_loglik_and_grad_all(params_from(mdl), y, t[..., None], hyp)
        assert _ffi.lib().cdkf_last_kernel().decode().startswith("ekf_adjoint_wg_kernel<double"), case
        np.testing.assert_allclose(ll, ll_ref, rtol=1e-9, err_msg=str(case))
        flat = np.concatenate([np.asarray(a_).reshape(N, -1) for a_ in g.dynamics.drift], axis=-1)
        pairs = [(flat, g_ref), (g.initial.mean.params, ex["m0"]), (g.initial.cov.params, ex["P0"]), (g.dynamics.diffusion_coefficient.params, ex["L"]),
                 (g.dynamics.diffusion_cov.params, ex["Qc"]), (g.emissions.emission_function.weights, ex["H"]),
                 (g.emissions.emission_function.bias, ex["bias"]), (g.emissions.emission_cov.params, ex["R"])]
        for a_, b_ in pairs:
            err = np.abs(np.asarray(a_) - b_).max() / (np.abs(b_).max() + 1e-300)
            worst = max(worst, err)
            assert err < 2e-8, (case, d, m, lin, solver, err)
    assert worst < 2e-8


@pytest.mark.parametrize("waves", [1, 2])
def test_soak_wavefront_reverse_sweep_of_lorenz96(hip_lib, waves, monkeypatch):
    """Fourteen random problems for ekf_adjoint_wave_l96_kernel / ekf_adjoint_wave2_l96_kernel (round 4: one or two wavefronts per
    trajectory): every instantiated state dimension 12 .. 40, a random
    selection of 1 .. d observed components in random order, dense symmetric L Qc L^T / R / P0, a random forcing, 1 .. 5 trajectories,
    1 .. 7 observations, intervals from zero length to thirty Runge-Kutta steps (up to four replay chunks of eight starts), both state
    orders -- every leaf against the oracle's discrete adjoint."""
    monkeypatch.setenv("CDKF_WAVE40_ADJ_WAVES", str(waves))
    rng = np.random.default_rng(777)
    worst = 0.0
    for case in range(14):
        d = int(rng.choice([12, 16, 20, 24, 28, 32, 36, 40]))
        m = d if case % 4 == 0 else int(rng.integers(1, d + 1))
        A, B, Cc = rng.standard_normal((d, d)), rng.standard_normal((m, m)), rng.standard_normal((d, d))
        RR = B @ B.T / m + 0.3 * np.eye(m)
        mdl = o.Model(o.Lorenz96Drift(8.0 + rng.standard_normal()), np.eye(d) + 0.1 * rng.standard_normal((d, d)), A @ A.T / d + 0.3 * np.eye(d),
                      np.eye(d)[rng.permutation(d)[:m]], np.zeros(m), 0.5 * (RR + RR.T), 8.0 + rng.standard_normal(d), Cc @ Cc.T / d + 0.5 * np.eye(d))
        N, T = int(rng.integers(1, 6)), int(rng.integers(1, 8))
        t = o.irregular_times(rng, N, T, 0.02 * T)
        if T > 3:
            t[:, 3:] += rng.uniform(0.05, 0.3)   # one long interval: 5 .. 30 steps
            t[0, 2] = t[0, 1]                    # and a repeated observation time
        y = o.simulate(mdl, t, rng)
        order = "second" if case % 2 else "first"
        ll_ref, g_ref, ex = o.ekf_loglik_grad_adjoint(mdl, t, y, full=True, state_order=order)
        ll, g = cd.cdnlgssm_loglik_and_grad_all(params_from(mdl), y, t[..., None], cd.EKFHyperParams(state_order=order))
        assert _ffi.lib().cdkf_last_kernel().decode().startswith("ekf_adjoint_wave%s_l96_kernel<double, %d>" % ("2" if waves == 2 else "", d)), case
        np.testing.assert_allclose(ll, ll_ref, rtol=1e-9, err_msg=str(case))
        flat = np.concatenate([np.asarray(a_).reshape(N, -1) for a_ in g.dynamics.drift], axis=-1)
        pairs = [(flat, g_ref), (g.initial.mean.params, ex["m0"]), (g.initial.cov.params, ex["P0"]), (g.dynamics.diffusion_coefficient.params, ex["L"]),
                 (g.dynamics.diffusion_cov.params, ex["Qc"]), (g.emissions.emission_function.weights, ex["H"]),
                 (g.emissions.emission_function.bias, ex["bias"]), (g.emissions.emission_cov.params, ex["R"])]
        for a_, b_ in pairs:
            err = np.abs(np.asarray(a_) - b_).max() / (np.abs(b_).max() + 1e-300)
            worst = max(worst, err)
            assert err < 2e-8, (case, d, m, order, err)
    assert worst < 2e-8


@pytest.mark.parametrize("d,m,dtype", [(43, 43, np.float64), (50, 12, np.float32), (62, 62, np.float32), (9, 1, np.float64)])
def test_reverse_sweep_at_the_edges_of_its_lds_plan(hip_lib, d, m, dtype):
    """ekf_adjoint_wg_kernel at the largest shapes its nine-matrix LDS plan admits (q = 43 in fp64, 62 in fp32; beyond d = 42 a thread owns
    up to sixteen covariance entries: the NE = 16 instantiation) and at the smallest (d = 9, m = 1), T = 1 and T = 3, N = 1 and 3:
    log-likelihood and d/dF against the oracle; one state dimension further the library refuses."""
    rng = np.random.default_rng(d * 100 + m)
    mdl = lorenz96_model(d, m)
    P = params_from(mdl)
    tol_ll, tol_g = (1e-10, 1e-8) if dtype == np.float64 else (2e-5, 5e-3)
    for N, T in ((1, 1), (3, 3)):
        t = o.irregular_times(rng, N, T, 0.012 * T)
        y = o.simulate(mdl, t, rng)
        ll_ref, g_ref = o.ekf_loglik_grad_adjoint(mdl, t, y)
        ll, g = cd.cdnlgssm_loglik_and_grad(P, y.astype(dtype), t[..., None].astype(dtype))
        name = _ffi.lib().cdkf_last_kernel().decode()
        assert name.startswith("ekf_adjoint_wg_kernel<%s, %d>" % ("double" if dtype == np.float64 else "float", 8 if d <= 42 else 16)), name
        np.testing.assert_allclose(ll, ll_ref, rtol=tol_ll)
        assert np.abs(np.asarray(g.forcing).reshape(N, 1) - g_ref).max() < tol_g * max(1.0, np.abs(g_ref).max()), (N, T)
    if (d, dtype) in ((43, np.float64), (62, np.float32)):  # one state dimension further the nine matrices no longer fit: refused, by the host gate or by the launch
        big = lorenz96_model(d + 1, d + 1)
        with pytest.raises(NotImplementedError):
            cd.cdnlgssm_loglik_and_grad(params_from(big), np.zeros((1, 2, d + 1), dtype), np.arange(2.0, dtype=dtype)[None, :, None])
