"""Drifts and emissions that depend on the inputs u and on time t (VERDICT r4 "missing" 2): the reference evaluates f(m, u, t),
jacfwd(f)(m, u, t), h(m, u, t) with u = inputs[t0_idx] held over the interval and t the solver's stage time (inference_ekf.py:95,
101-114, 277-286; inference_ukf.py:142, 189; reverse-time in the smoother, diffrax_utils.py:13-25; cdnlgssm_utils.py:13-61: a
LearnableFunction is any callable of (x, u, t)).  Here: the oracle's restatement pinned by an independent integrator (CPU), the
run-time compiled register kernels on the host under ASan (CPU, tests/hostsim), and the HIP path against the oracle (GPU)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import cdkf_oracle as o  # noqa: E402
import cd_dynamax_amd as cd  # noqa: E402
from helpers import relerr  # noqa: E402

FILTER_KEYS = ("marginal_loglik", "filtered_means", "filtered_covariances", "predicted_means", "predicted_covariances")

# forced, controlled Lorenz-63: rho(t) = rho + 4 sin t; u_0 pushes x, u_1 modulates the damping of z
L63_UT = ("const auto rho = theta[1] + R(4) * sin(t);"
          "fx[0] = theta[0] * (x[1] - x[0]) + u[0]; fx[1] = x[0] * (rho - x[2]) - x[1];"
          "fx[2] = x[0] * x[1] - theta[2] * x[2] + R(0.5) * u[1] * x[2];")


def l63_ut_f(x, th, u, t):
    rho = th[1] + 4 * np.sin(t)
    return np.stack([th[0] * (x[..., 1] - x[..., 0]) + u[..., 0], x[..., 0] * (rho - x[..., 2]) - x[..., 1],
                     x[..., 0] * x[..., 1] - th[2] * x[..., 2] + 0.5 * u[..., 1] * x[..., 2]], -1)


def l63_ut_jac(x, th, u, t):
    rho = th[1] + 4 * np.sin(t)
    J = np.zeros(x.shape + (3,))
    J[..., 0, 0], J[..., 0, 1] = -th[0], th[0]
    J[..., 1, 0], J[..., 1, 1], J[..., 1, 2] = rho - x[..., 2], -1, -x[..., 0]
    J[..., 2, 0], J[..., 2, 1], J[..., 2, 2] = x[..., 1], x[..., 0], -th[2] + 0.5 * u[..., 1]
    return J


def l63_ut_vjp(x, lam, G, th, u, t):
    """gradient of lam . f + <G, F> w.r.t. (x, theta) for one row (what the oracle's reverse sweep needs of a callable drift)"""
    F = l63_ut_jac(x[None], th, u, t)[0]
    xb = F.T @ lam + np.array([-G[1, 2] + G[2, 1], G[2, 0], -G[1, 0]])
    return xb, np.array([lam[0] * (x[1] - x[0]) - G[0, 0] + G[0, 1], lam[1] * x[0] + G[1, 0], -lam[2] * x[2] - G[2, 2]])


def l63_ut_model(theta, m_obs=2):
    base = o.lorenz63_model(m_obs)
    drift = o.CallableDrift(theta, l63_ut_f, l63_ut_jac, lambda x, th, u, t: np.zeros_like(x), ut=True, vjp=l63_ut_vjp,
                            gvjp=lambda x, uu, th, u, t: (np.zeros(3), np.zeros(3)))
    return o.Model(drift, base.L, base.Qc, base.H, base.bias, base.R, base.m0, base.P0)


def l63_ut_params(mdl, theta, divgrad=""):
    return cd.ParamsCDNLGSSM(
        initial=cd.ParamsLGSSMInitial(cd.LearnableVector(mdl.m0), cd.LearnableMatrix(mdl.P0)),
        dynamics=cd.ParamsCDNLGSSMDynamics(cd.LearnableCustomDrift(theta, L63_UT, None, divgrad), cd.LearnableMatrix(mdl.L),
                                           cd.LearnableMatrix(mdl.Qc), 2.0),
        emissions=cd.ParamsCDNLGSSMEmissions(cd.LearnableLinear(mdl.H, mdl.bias), cd.LearnableMatrix(mdl.R)))


def problem(N=3, T=12, seed=0):
    rng = np.random.default_rng(seed)
    theta = np.array([10.0, 28.0, 8.0 / 3.0])
    mdl = l63_ut_model(theta)
    t = o.irregular_times(rng, N, T, 0.3) + 1.0
    u = rng.standard_normal((N, T, 2))
    y = 3 * rng.standard_normal((N, T, 2))
    return theta, mdl, t, y, u


def test_oracle_stage_times_and_inputs_against_an_independent_integrator():
    """The oracle's predict and smoother steps with f(x, u, t) against scipy's solve_ivp (rtol 1e-12) of the same moment equations
    written with explicit time: forward over [t0, t1] (stage times t_prev + c_i h) and backwards (the reference integrates
    -rhs(t1 - s) over s in [0, t1 - t0]: diffrax_utils.py:13-25), dt0 = 1e-3 so that the fixed-step error is below the bar."""
    from scipy.integrate import solve_ivp
    theta, mdl, t, y, u = problem(N=2, T=3)
    LQL = mdl.L @ mdl.Qc @ mdl.L.T
    m0, P0 = np.array([[1.0, -2.0, 20.0], [3.0, 1.0, 25.0]]), np.stack([np.eye(3) * 0.5, np.eye(3) * 0.2])
    t0, t1 = np.array([1.0, 2.5]), np.array([1.3, 2.7])
    o._CTX["u"] = u[:, 0]
    m1, P1 = o.ekf_predict(mdl, m0, P0, t0, t1, "first", dt0=1e-3)
    for n in range(2):
        def rhs(tt, z, n=n):
            x, P = z[:3], z[3:].reshape(3, 3)
            un = u[n:n + 1, 0]
            F = l63_ut_jac(x[None], theta, un, np.array([tt]))[0]
            return np.concatenate([l63_ut_f(x[None], theta, un, np.array([tt]))[0], (F @ P + P @ F.T + LQL).ravel()])
        sol = solve_ivp(rhs, (t0[n], t1[n]), np.concatenate([m0[n], P0[n].ravel()]), rtol=1e-12, atol=1e-14, method="DOP853")
        assert np.abs(sol.y[:3, -1] - m1[n]).max() < 1e-9 * np.abs(m1[n]).max()
        assert np.abs(sol.y[3:, -1].reshape(3, 3) - P1[n]).max() < 1e-9 * np.abs(P1[n]).max()
    # smoother step k = 0 of a two-observation sequence, against the backward ODE integrated in physical time from t1 down to t0
    tt = np.stack([t0, t1], 1)
    yy = y[:, :2]
    flt = o.ekf_filter(mdl, tt, yy, "first", dt0=1e-3, inputs=u[:, :2])
    sm = o.ekf_smoother(mdl, tt, yy, "first", dt0=1e-3, inputs=u[:, :2], filtered=flt)
    for n in range(2):
        mf, Pf = flt["filtered_means"][n, 0], flt["filtered_covariances"][n, 0]
        aux = np.linalg.solve(0.5 * (Pf + Pf.T) + 1e-9 * np.eye(3), LQL).T

        def rhs(tt_, z, n=n, mf=mf, aux=aux):   # d/dt in physical time (inference_ekf.py:433-438), solved from t1 down to t0
            ms, Ps = z[:3], z[3:].reshape(3, 3)
            un = u[n:n + 1, 0]
            G = l63_ut_jac(mf[None], theta, un, np.array([tt_]))[0] + aux
            return np.concatenate([l63_ut_f(mf[None], theta, un, np.array([tt_]))[0] + G @ (ms - mf), (G @ Ps + Ps @ G.T - LQL).ravel()])
        z1 = np.concatenate([flt["filtered_means"][n, 1], flt["filtered_covariances"][n, 1].ravel()])
        sol = solve_ivp(rhs, (t1[n], t0[n]), z1, rtol=1e-12, atol=1e-14, method="DOP853")
        assert np.abs(sol.y[:3, -1] - sm["smoothed_means"][n, 0]).max() < 1e-8 * np.abs(sm["smoothed_means"][n, 0]).max()
        assert np.abs(sol.y[3:, -1].reshape(3, 3) - sm["smoothed_covariances"][n, 0]).max() < 1e-8 * np.abs(sm["smoothed_covariances"][n, 0]).max()


def test_oracle_ignores_inputs_for_registry_drifts_and_zero_inputs_change_nothing():
    """The reference's own Learnable* drifts ignore u and t (cdnlgssm_utils.py:50-83): so does the oracle; and a callable drift
    that reads them gives other numbers with other inputs."""
    rng = np.random.default_rng(3)
    mdl = o.lorenz63_model(3)
    t = o.irregular_times(rng, 2, 8, 0.1)
    y = o.simulate(mdl, t, rng)
    a, b = o.ekf_filter(mdl, t, y), o.ekf_filter(mdl, t, y, inputs=rng.standard_normal((2, 8, 4)))
    for k in FILTER_KEYS:
        assert np.array_equal(a[k], b[k])
    theta, mdl2, t2, y2, u2 = problem()
    z = o.ekf_filter(mdl2, t2, y2, "first", inputs=np.zeros((3, 12, 2)))
    w = o.ekf_filter(mdl2, t2, y2, "first", inputs=u2)
    assert np.all(np.isfinite(z["marginal_loglik"])) and np.abs(z["marginal_loglik"] - w["marginal_loglik"]).max() > 1e-3   # (inputs matter)


def _grad_fd(theta, t, y, u):
    """Richardson-extrapolated central differences of the oracle's EKF log-likelihood w.r.t. theta."""
    def ll_of(thv):
        return o.ekf_filter(l63_ut_model(thv), t, y, "first", inputs=u)["marginal_loglik"]
    fd = lambda h: np.stack([(ll_of(theta + h * np.eye(3)[p]) - ll_of(theta - h * np.eye(3)[p])) / (2 * h) for p in range(3)], -1)
    return (4 * fd(5e-5) - fd(1e-4)) / 3


def test_oracle_reverse_sweep_with_inputs_and_time_matches_finite_differences():
    """ekf_loglik_grad_adjoint(inputs=...) -- the reverse sweep with the stage times and the interval's inputs in the context -- against
    Richardson finite differences of ekf_filter's log-likelihood (drift block), fixed steps and an adaptive solve."""
    theta, mdl, t, y, u = problem(N=2, T=8)
    ll, g = o.ekf_loglik_grad_adjoint(mdl, t, y, state_order="first", inputs=u)
    np.testing.assert_allclose(ll, o.ekf_filter(mdl, t, y, "first", inputs=u)["marginal_loglik"], rtol=1e-12)
    g_fd = _grad_fd(theta, t, y, u)
    assert np.abs(g - g_fd).max() < 1e-8 * np.abs(g_fd).max()
    ll2, g2 = o.ekf_loglik_grad_adjoint(mdl, t, y, state_order="second", inputs=u)   # (grad(div f) = 0 for this drift)
    assert np.abs(g2 - g).max() < 1e-12 * np.abs(g).max()


def test_register_kernels_with_inputs_and_time_on_the_host():
    """The run-time compiled register kernels (EKF, UKF, smoother's two passes, forward-sensitivity gradient) with the forced,
    controlled Lorenz-63 source, compiled for the host and run under ASan + UBSan on the launcher's own argument blocks: 1e-12 of the
    oracle (the gradient: 1e-8 of Richardson finite differences of the oracle's log-likelihood).  (The EKF leg under the sanitizers, the
    others as plain host builds: same templates, a third of the compile time.)"""
    import hostsim_util as hs
    import shutil
    if hs.clang() is None or shutil.which("hipcc") is None:
        pytest.skip("needs clang++ and the HIP library")
    from cd_dynamax_amd import _ffi, models
    theta, mdl, t, y, u = problem()
    N, T, d, m = 3, 12, 3, 2
    mb = models._model_block(l63_ut_params(mdl, theta))
    TN = lambda a, shape: np.swapaxes(a.reshape((T, N) + shape), 0, 1)

    def run(algo, hyp, outs, layout=_ffi.LAYOUT_TN, which=0, preload=None, san="plain"):
        opts = models._opts(hyp)
        opts.layout, opts.layout_in, opts.t_shared = layout, _ffi.LAYOUT_NT, 0
        models._attach_inputs(mb, opts, u, y, np.float64)
        ddir, srcs = hs.dump_custom_source(mb.c.drift_kind, 8, m, algo + 256 * 2, 1)
        return hs.reg_run(os.path.join(ddir, srcs[which]), mb, opts, t, y, algo, np.float64, san, outs, inputs=u, preload=preload)
    full = (N, N * T * d, N * T * d * d, N * T * d, N * T * d * d, N, 0, 0)
    for algo, hyp, ref in ((0, cd.EKFHyperParams(state_order="first"), o.ekf_filter(mdl, t, y, "first", inputs=u)),
                           (1, cd.UKFHyperParams(), o.ukf_filter(mdl, t, y, inputs=u))):
        ll, fm, fP, pm, pP, *_ = run(algo, hyp, full, san="asan" if algo == 0 else "plain")   # (one sanitizer build: they take 3 x as long)
        got = dict(marginal_loglik=ll, filtered_means=TN(fm, (d,)), filtered_covariances=TN(fP, (d, d)), predicted_means=TN(pm, (d,)),
                   predicted_covariances=TN(pP, (d, d)))
        for k in FILTER_KEYS:
            assert relerr(got[k], ref[k]) < 1e-12, (algo, k)
    hyp = cd.EKFHyperParams(state_order="first")
    ll, g, *_ = run(3, hyp, (N, N * 3, 0, 0, 0, N, 0, 0), layout=_ffi.LAYOUT_TCN)
    g_fd = _grad_fd(theta, t, y, u)
    assert np.abs(g.reshape(N, 3) - g_fd).max() < 1e-8 * np.abs(g_fd).max()
    ref = o.ekf_smoother(mdl, t, y, "first", inputs=u)
    _, fm, fP, *_ = run(2, hyp, (N, N * T * d, N * T * d * d, 0, 0, N, 0, 0), which=0)
    *_, sm, sP = run(2, hyp, (N, N * T * d, N * T * d * d, 0, 0, N, N * T * d, N * T * d * d), which=1, preload=(fm, fP))
    assert relerr(TN(sm, (d,)), ref["smoothed_means"]) < 1e-12 and relerr(TN(sP, (d, d)), ref["smoothed_covariances"]) < 1e-12


# ---- the HIP path ------------------------------------------------------------------------------------------------------------------
PEND_UT = "fx[0] = x[1]; fx[1] = -theta[0] * sin(x[0]) - theta[1] * x[1] + u[0];"      # torque-controlled pendulum
H_UT = "hx[0] = eta[0] * sin(x[0]) * (R(1) + R(0.3) * cos(t)) + eta[1] * u[0];"          # time- and input-dependent emission


def _pend_model(theta, eta, ut_emission):
    f = lambda x, th, u, t: np.stack([x[..., 1], -th[0] * np.sin(x[..., 0]) - th[1] * x[..., 1] + u[..., 0]], -1)

    def jac(x, th, u, t):
        J = np.zeros(x.shape + (2,))
        J[..., 0, 1] = 1
        J[..., 1, 0], J[..., 1, 1] = -th[0] * np.cos(x[..., 0]), -th[1]
        return J
    g = lambda x, th, u, t: np.stack([th[0] * np.sin(x[..., 0]) * 0, np.zeros_like(x[..., 0])], -1)   # d/dx_k sum_i dF_ii = 0 (F_11 = -th1)
    h = lambda x, eta, u, t: (eta[0] * np.sin(x[..., 0]) * (1 + 0.3 * np.cos(t)) + eta[1] * u[..., 0])[..., None]

    def hj(x, eta, u, t):
        H = np.zeros(x.shape[:-1] + (1, 2))
        H[..., 0, 0] = eta[0] * np.cos(x[..., 0]) * (1 + 0.3 * np.cos(t))
        return H
    return o.Model(o.CallableDrift(theta, f, jac, g, ut=True), np.eye(2), np.array([[0.05, 0.01], [0.01, 0.1]]), eta[None, :], np.zeros(1),
                   0.1 * np.eye(1), np.array([0.7, 0.0]), 0.3 * np.eye(2), emission=(h, hj) if ut_emission else None, emission_ut=ut_emission)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-9), (np.float32, 1e-4)])
def test_forced_lorenz63_with_inputs_through_the_python_surface(hip_lib, dtype, tol):
    """rho(t) = rho + 4 sin t and two inputs, d = 3 register path: EKF (first and second order), UKF, smoother against the oracle;
    the drift-block gradient against Richardson finite differences of the oracle's log-likelihood (fp64: 1e-8)."""
    theta, mdl, t, y, u = problem(N=37, T=25, seed=4)
    P = l63_ut_params(mdl, theta)
    yy, tt, uu = y.astype(dtype), t[..., None].astype(dtype), u.astype(dtype)
    for order in ("first", "second"):
        ref = o.ekf_filter(mdl, t, y, order, inputs=u)
        post = cd.cdnlgssm_filter(P, yy, tt, cd.EKFHyperParams(state_order=order), inputs=uu)
        for k in FILTER_KEYS:
            assert relerr(getattr(post, k), ref[k]) < tol, (order, k)
    refu = o.ukf_filter(mdl, t, y, inputs=u)
    postu = cd.cdnlgssm_filter(P, yy, tt, cd.UKFHyperParams(), inputs=uu)
    for k in FILTER_KEYS:
        assert relerr(getattr(postu, k), refu[k]) < tol * 10, k
    refs = o.ekf_smoother(mdl, t, y, "first", inputs=u)
    sm = cd.cdnlgssm_smoother(P, yy, tt, cd.EKFHyperParams(state_order="first"), inputs=uu)
    assert relerr(sm.smoothed_means, refs["smoothed_means"]) < tol and relerr(sm.smoothed_covariances, refs["smoothed_covariances"]) < tol * 10
    # inputs shared by the batch ([T, d_u]) and the same inputs given per trajectory agree bitwise
    shared = cd.cdnlgssm_filter(P, yy, tt, cd.EKFHyperParams(state_order="first"), inputs=uu[0])
    tiled = cd.cdnlgssm_filter(P, yy, tt, cd.EKFHyperParams(state_order="first"), inputs=np.broadcast_to(uu[0], uu.shape))
    assert np.array_equal(shared.filtered_means, tiled.filtered_means)
    if dtype == np.float64:
        ll, g = cd.cdnlgssm_loglik_and_grad(P, y[:5], t[:5, :, None], cd.EKFHyperParams(state_order="first"), inputs=u[:5])
        g_fd = _grad_fd(theta, t[:5], y[:5], u[:5])
        assert np.abs(np.asarray(g.theta) - g_fd).max() < 1e-8 * np.abs(g_fd).max()
        np.testing.assert_allclose(ll, o.ekf_filter(mdl, t[:5], y[:5], "first", inputs=u[:5])["marginal_loglik"], rtol=1e-11)


@pytest.mark.gpu
def test_controlled_pendulum_and_time_dependent_emission(hip_lib):
    """u enters the drift (a torque) and the emission (a feed-through), t enters the emission: EKF with 1 and 3 re-linearisations, UKF,
    smoother; emission Jacobian by dual numbers (u, t constant under differentiation)."""
    rng = np.random.default_rng(83)
    theta, eta = np.array([2.0, 0.3]), np.array([1.5, 0.2])
    mdl = _pend_model(theta, eta, True)
    N, T = 23, 30
    t = o.irregular_times(rng, N, T, 0.4) + 0.5
    u = np.sin(3 * t)[..., None] + 0.3 * rng.standard_normal((N, T, 1))
    y = 0.8 * rng.standard_normal((N, T, 1))
    P = cd.ParamsCDNLGSSM(
        initial=cd.ParamsLGSSMInitial(cd.LearnableVector(mdl.m0), cd.LearnableMatrix(mdl.P0)),
        dynamics=cd.ParamsCDNLGSSMDynamics(cd.LearnableCustomDrift(theta, PEND_UT, None, ""), cd.LearnableMatrix(mdl.L), cd.LearnableMatrix(mdl.Qc), 2.0),
        emissions=cd.ParamsCDNLGSSMEmissions(cd.LearnableCustomEmission(eta, H_UT), cd.LearnableMatrix(mdl.R)))
    for num_iter in (1, 3):
        ref = o.ekf_filter(mdl, t, y, "second", num_iter=num_iter, inputs=u)
        post = cd.cdnlgssm_filter(P, y, t[..., None], cd.EKFHyperParams(), inputs=u, num_iter=num_iter)
        for k in FILTER_KEYS:
            assert relerr(getattr(post, k), ref[k]) < 1e-9, (num_iter, k)
    refu = o.ukf_filter(mdl, t, y, inputs=u)
    postu = cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams(), inputs=u)
    for k in FILTER_KEYS:
        assert relerr(getattr(postu, k), refu[k]) < 1e-9, k
    refs = o.ekf_smoother(mdl, t, y, "first", inputs=u)
    sm = cd.cdnlgssm_smoother(P, y, t[..., None], cd.EKFHyperParams(state_order="first"), inputs=u)
    assert relerr(sm.smoothed_means, refs["smoothed_means"]) < 1e-9


# ---- beyond six dimensions: the workgroup kernels and the reverse sweep ------------------------------------------------------------
D8, M8 = 8, 5


def _l96_ut(d=D8):
    """Lorenz-96 with a time-dependent forcing and two inputs: f_i = (x_{i+1} - x_{i-2}) x_{i-1} - x_i + th_0 + 2 sin(t + i) + th_1 u_{i mod 2}"""
    src = "".join(f"fx[{i}] = (x[{(i + 1) % d}] - x[{(i - 2) % d}]) * x[{(i - 1) % d}] - x[{i}] + theta[0] + R(2) * sin(t + R({i})) + theta[1] * u[{i % 2}];"
                  for i in range(d))
    idx = np.arange(d)

    def f(x, th, u, t):
        return (x[..., (idx + 1) % d] - x[..., (idx - 2) % d]) * x[..., (idx - 1) % d] - x + th[0] + 2 * np.sin(t[..., None] + idx) + th[1] * u[..., idx % 2]

    def jac(x, th, u, t):
        J = np.zeros(x.shape + (d,))
        for i in range(d):
            J[..., i, (i + 1) % d] += x[..., (i - 1) % d]
            J[..., i, (i - 2) % d] -= x[..., (i - 1) % d]
            J[..., i, (i - 1) % d] += x[..., (i + 1) % d] - x[..., (i - 2) % d]
            J[..., i, i] -= 1
        return J

    def vjp(x, lam, G, th, u, t):   # gradient of lam . f + <G, F> w.r.t. (x, theta), one row
        F = jac(x[None], th, u, t)[0]
        xb = F.T @ lam
        for i in range(d):
            xb[(i - 1) % d] += G[i, (i + 1) % d] - G[i, (i - 2) % d]
            xb[(i + 1) % d] += G[i, (i - 1) % d]
            xb[(i - 2) % d] -= G[i, (i - 1) % d]
        return xb, np.array([lam.sum(), lam @ u[0, idx % 2]])
    return src, f, jac, vjp


def l96_ut_problem(N=2, T=6, span=0.1, seed=1, d=D8, m=M8):
    rng = np.random.default_rng(seed)
    theta = np.array([8.0, 0.7])
    src, f, jac, vjp = _l96_ut(d)
    H = rng.standard_normal((m, d)) / np.sqrt(d)
    drift = o.CallableDrift(theta, f, jac, lambda x, th, u, t: np.zeros_like(x), ut=True, vjp=vjp, gvjp=lambda x, uu, th, u, t: (np.zeros(d), np.zeros(2)))
    mdl = o.Model(drift, np.eye(d), 0.3 * np.eye(d), H, 0.1 * rng.standard_normal(m), 0.5 * np.eye(m), 8 + rng.standard_normal(d), 0.5 * np.eye(d))
    t = o.irregular_times(rng, N, T, span) + 0.5
    u = rng.standard_normal((N, T, 2))
    y = 2 * rng.standard_normal((N, T, m)) + 5
    P = cd.ParamsCDNLGSSM(
        initial=cd.ParamsLGSSMInitial(cd.LearnableVector(mdl.m0), cd.LearnableMatrix(mdl.P0)),
        dynamics=cd.ParamsCDNLGSSMDynamics(cd.LearnableCustomDrift(theta, src, None, ""), cd.LearnableMatrix(mdl.L), cd.LearnableMatrix(mdl.Qc), 2.0),
        emissions=cd.ParamsCDNLGSSMEmissions(cd.LearnableLinear(mdl.H, mdl.bias), cd.LearnableMatrix(mdl.R)))
    return theta, mdl, P, t, y, u


def test_workgroup_kernels_and_reverse_sweep_with_inputs_and_time_on_the_host():
    """d = 8 on the run-time compiled workgroup kernels, host build (the filter and the reverse sweep under ASan + UBSan): EKF, UKF, the
    smoother's backward sweep and the reverse sweep of the gradient (every leaf), each reading the interval's inputs row and the stage
    times (t1 - s backwards)."""
    import hostsim_util as hs
    import shutil
    if hs.clang() is None or shutil.which("hipcc") is None:
        pytest.skip("needs clang++ and the HIP library")
    from cd_dynamax_amd import models
    theta, mdl, P, t, y, u = l96_ut_problem()
    mb = models._model_block(P)
    unit = lambda srcs, tail: [s for s in srcs if s.endswith(tail)][0]

    def opts_for(hyp):
        opts = models._opts(hyp)
        models._attach_inputs(mb, opts, u, y, np.float64)
        return opts
    hyp = cd.EKFHyperParams(state_order="first")
    ref = o.ekf_filter(mdl, t, y, "first", inputs=u)
    ddir, srcs = hs.dump_custom_source(mb.c.drift_kind, 8, M8, 0 + 256 * 2, 1)
    fwd = hs.wg_run(os.path.join(ddir, unit(srcs, "_0.hip")), mb, opts_for(hyp), t, y, np.float64, "asan", kind=-1, inputs=u)
    for k in FILTER_KEYS:
        assert relerr(fwd[k], ref[k]) < 1e-12, k
    ll_a, g_a, ex = o.ekf_loglik_grad_adjoint(mdl, t, y, full=True, state_order="first", inputs=u)
    ddir, srcs = hs.dump_custom_source(mb.c.drift_kind, 8, M8, 3 + 256 * 2, 1)
    g, gm, st = hs.awg_run(os.path.join(ddir, unit(srcs, "_2.hip")), mb, opts_for(hyp), t, y, np.float64, "asan", fwd, inputs=u)
    assert np.abs(g - g_a).max() < 1e-9 * np.abs(g_a).max() and not st.any()
    assert np.abs(gm[:, :D8] - ex["m0"]).max() < 1e-9 * np.abs(ex["m0"]).max()
    if os.environ.get("CDKF_HOSTSIM_FULL") != "1":   # (the unscented filter and the backward sweep: the GPU legs below hold them; a minute of host builds)
        return
    refu = o.ukf_filter(mdl, t, y, inputs=u)
    ddir, srcs = hs.dump_custom_source(mb.c.drift_kind, 8, M8, 1 + 256 * 2, 1)
    outu = hs.wg_run(os.path.join(ddir, unit(srcs, "_0.hip")), mb, opts_for(cd.UKFHyperParams()), t, y, np.float64, "plain", ukf=True, kind=-1, inputs=u)
    for k in FILTER_KEYS:
        assert relerr(outu[k], refu[k]) < 1e-11, k
    refs = o.ekf_smoother(mdl, t, y, "first", inputs=u, filtered=ref)
    ddir, srcs = hs.dump_custom_source(mb.c.drift_kind, 8, M8, 2 + 256 * 2, 1)
    outs = hs.wg_run(os.path.join(ddir, unit(srcs, "_1.hip")), mb, opts_for(hyp), t, y, np.float64, "plain", smoother=True, kind=-1, inputs=u,
                     filtered=(ref["filtered_means"], ref["filtered_covariances"]))
    assert relerr(outs["smoothed_means"], refs["smoothed_means"]) < 1e-12 and relerr(outs["smoothed_covariances"], refs["smoothed_covariances"]) < 1e-12


@pytest.mark.gpu
def test_workgroup_path_with_inputs_and_time(hip_lib):
    """d = 12, m = 7 (the run-time compiled workgroup kernels; forward sweeps and the reverse sweep): EKF / UKF / smoother against the
    oracle at 1e-9, every gradient leaf at 1e-8; a long-gap grid (several Runge-Kutta steps and replay chunks per interval) as well."""
    for span, T in ((0.1, 10), (1.2, 8)):
        theta, mdl, P, t, y, u = l96_ut_problem(N=5, T=T, span=span, seed=7, d=12, m=7)
        ref = o.ekf_filter(mdl, t, y, "first", inputs=u)
        post = cd.cdnlgssm_filter(P, y, t[..., None], cd.EKFHyperParams(state_order="first"), inputs=u)
        for k in FILTER_KEYS:
            assert relerr(getattr(post, k), ref[k]) < 1e-9, (span, k)
        refu = o.ukf_filter(mdl, t, y, inputs=u)
        postu = cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams(), inputs=u)
        for k in FILTER_KEYS:
            assert relerr(getattr(postu, k), refu[k]) < 1e-9, (span, k)
        refs = o.ekf_smoother(mdl, t, y, "first", inputs=u, filtered=ref)
        sm = cd.cdnlgssm_smoother(P, y, t[..., None], cd.EKFHyperParams(state_order="first"), inputs=u)
        assert relerr(sm.smoothed_means, refs["smoothed_means"]) < 1e-9 and relerr(sm.smoothed_covariances, refs["smoothed_covariances"]) < 1e-9
        ll_a, g_a, ex = o.ekf_loglik_grad_adjoint(mdl, t, y, full=True, state_order="first", inputs=u)
        ll, g = cd.cdnlgssm_loglik_and_grad_all(P, y, t[..., None], cd.EKFHyperParams(state_order="first"), inputs=u)
        np.testing.assert_allclose(ll, ll_a, rtol=1e-10)
        assert np.abs(np.asarray(g.dynamics.drift.theta) - g_a).max() < 1e-8 * np.abs(g_a).max(), span
        assert np.abs(np.asarray(g.initial.mean.params) - ex["m0"]).max() < 1e-8 * np.abs(ex["m0"]).max()
        assert np.abs(np.asarray(g.emissions.emission_function.weights) - ex["H"]).max() < 1e-8 * np.abs(ex["H"]).max()


@pytest.mark.gpu
def test_small_model_reverse_sweep_with_inputs_and_time(hip_lib):
    """The forced, controlled Lorenz-63 through cdnlgssm_loglik_and_grad_all (d = 3 goes to the reverse sweep of the workgroup kernels
    for a drift given as source): every leaf against the oracle's reverse sweep."""
    theta, mdl, t, y, u = problem(N=6, T=15, seed=9)
    P = l63_ut_params(mdl, theta)
    ll_a, g_a, ex = o.ekf_loglik_grad_adjoint(mdl, t, y, full=True, state_order="first", inputs=u)
    ll, g = cd.cdnlgssm_loglik_and_grad_all(P, y, t[..., None], cd.EKFHyperParams(state_order="first"), inputs=u)
    np.testing.assert_allclose(ll, ll_a, rtol=1e-10)
    assert np.abs(np.asarray(g.dynamics.drift.theta) - g_a).max() < 1e-8 * np.abs(g_a).max()
    assert np.abs(np.asarray(g.emissions.emission_cov.params) - ex["R"]).max() < 1e-8 * np.abs(ex["R"]).max()
