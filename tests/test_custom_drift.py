"""User-supplied drifts compiled at run time (cdkf_custom_drift_register; reference: any callable as
ParamsCDNLGSSMDynamics.drift, cdnlgssm_utils.py:38-61).  CPU: registration, hipRTC compilation of every variant (no GPU
needed), diagnostics for broken snippets.  GPU: parity with the oracle running the same drift as NumPy callables."""
import os
import numpy as np
import pytest

import cd_dynamax_amd as cd
import cdkf_oracle as o
from cd_dynamax_amd import _ffi
from helpers import FILTER_KEYS, random_quadratic_drift, relerr

PEND_F = "fx[0] = x[1]; fx[1] = -theta[0] * sin(x[0]) - theta[1] * x[1];"
PEND_J = "F[0][1] = R(1); F[1][0] = -theta[0] * cos(x[0]); F[1][1] = -theta[1];"
# Van der Pol: f = [x1, mu (1 - x0^2) x1 - x0];  div f = mu (1 - x0^2);  grad(div f) = [-2 mu x0, 0]
VDP_F = "fx[0] = x[1]; fx[1] = theta[0] * (R(1) - x[0] * x[0]) * x[1] - x[0];"
VDP_J = "F[0][1] = R(1); F[1][0] = -R(2) * theta[0] * x[0] * x[1] - R(1); F[1][1] = theta[0] * (R(1) - x[0] * x[0]);"
VDP_G = "g[0] = -R(2) * theta[0] * x[0];"


def pendulum_oracle(theta):
    f = lambda x, th: np.stack([x[..., 1], -th[0] * np.sin(x[..., 0]) - th[1] * x[..., 1]], -1)

    def jac(x, th):
        F = np.zeros(x.shape + (2,), x.dtype)
        F[..., 0, 1] = 1
        F[..., 1, 0] = -th[0] * np.cos(x[..., 0])
        F[..., 1, 1] = -th[1]
        return F
    return o.CallableDrift(theta, f, jac, None)


def vdp_oracle(theta):
    f = lambda x, th: np.stack([x[..., 1], th[0] * (1 - x[..., 0] ** 2) * x[..., 1] - x[..., 0]], -1)

    def jac(x, th):
        F = np.zeros(x.shape + (2,), x.dtype)
        F[..., 0, 1] = 1
        F[..., 1, 0] = -2 * th[0] * x[..., 0] * x[..., 1] - 1
        F[..., 1, 1] = th[0] * (1 - x[..., 0] ** 2)
        return F
    g = lambda x, th: np.stack([-2 * th[0] * x[..., 0], np.zeros_like(x[..., 0])], -1)
    return o.CallableDrift(theta, f, jac, g)


def make_model(drift, m):
    H = np.eye(2)[:m] if m <= 2 else np.vstack([np.eye(2), [[1.0, -1.0]]])
    return o.Model(drift, np.eye(2), np.array([[0.05, 0.01], [0.01, 0.1]]), H, np.zeros(m) + 0.05, 0.2 * np.eye(m),
                   np.array([1.0, 0.0]), 0.5 * np.eye(2))


def params_for(mdl, drift):
    return cd.ParamsCDNLGSSM(
        initial=cd.ParamsLGSSMInitial(cd.LearnableVector(mdl.m0), cd.LearnableMatrix(mdl.P0)),
        dynamics=cd.ParamsCDNLGSSMDynamics(drift, cd.LearnableMatrix(mdl.L), cd.LearnableMatrix(mdl.Qc), 2.0),
        emissions=cd.ParamsCDNLGSSMEmissions(cd.LearnableLinear(mdl.H, mdl.bias), cd.LearnableMatrix(mdl.R)))


def test_register_and_compile_all_variants():
    k1 = _ffi.register_custom_drift(2, 2, PEND_F, PEND_J, None)
    assert k1 >= _ffi.DRIFT_CUSTOM_BASE and _ffi.register_custom_drift(2, 2, PEND_F, PEND_J, None) == k1   # same sources, same kind
    k2 = _ffi.register_custom_drift(2, 1, VDP_F, VDP_J, VDP_G)
    assert k2 != k1
    L = _ffi.lib()
    for kind in (k1, k2):
        for nbytes in (8, 4):
            for algo in (0, 1, 2):
                assert L.cdkf_custom_drift_compile(kind, nbytes, 1, algo, 1, 0) == 0, L.cdkf_last_error().decode()
    assert L.cdkf_custom_drift_compile(k2, 8, 3, 0, 0, 0) == 0                # zeroth order, emission_dim 3
    with pytest.raises(_ffi.CdkfError, match="dual numbers"):
        _ffi.register_custom_drift(9, 1, PEND_F, PEND_J, None)             # state_dim > 6: no hand-written Jacobian (see below)
    with pytest.raises(_ffi.CdkfError):
        _ffi.register_custom_drift(65, 1, PEND_F, None, None)
    assert L.cdkf_custom_drift_compile(12345, 8, 1, 0, 1, 0) != 0


def test_broken_snippet_reports_the_compiler_diagnostic():
    kind = _ffi.register_custom_drift(2, 1, "fx[0] = x[1]; fx[1] = -theta[0] * sine(x[0]);", "F[0][1] = R(1);", None)
    L = _ffi.lib()
    assert L.cdkf_custom_drift_compile(kind, 8, 1, 0, 1, 0) != 0
    msg = L.cdkf_last_error().decode()
    assert "drift_f:1" in msg and "sine" in msg


@pytest.mark.gpu
@pytest.mark.parametrize("m", [1, 2, 3])
def test_custom_drift_filters_and_smoother(hip_lib, m):
    """Damped pendulum as a run-time compiled drift: EKF (first / zeroth order), UKF, EKF smoother, fp32, forecast --
    against the oracle evaluating the same drift through NumPy callables; irregular per-trajectory times, N = 70."""
    rng = np.random.default_rng(60 + m)
    theta = np.array([2.0, 0.3])
    mdl = make_model(pendulum_oracle(theta), m)
    N, T = 70, 30
    t = o.irregular_times(rng, N, T, 0.08)
    y = o.simulate(mdl, t, rng)
    P = params_for(mdl, cd.LearnableCustomDrift(theta, PEND_F, PEND_J))
    for order in ("first", "zeroth"):
        ref = o.ekf_filter(mdl, t, y, state_order=order)
        post = cd.cdnlgssm_filter(P, y, t[..., None], cd.EKFHyperParams(state_order=order))
        for k in FILTER_KEYS:
            assert relerr(getattr(post, k), ref[k]) < 1e-11, (order, k)
        np.testing.assert_allclose(post.marginal_loglik, ref["marginal_loglik"], rtol=1e-11)
    with pytest.raises(_ffi.CdkfError, match="state_order 'second'"):
        cd.cdnlgssm_filter(P, y, t[..., None], cd.EKFHyperParams())              # default order needs grad(div f)
    ref = o.ukf_filter(mdl, t, y)
    post = cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams())
    for k in FILTER_KEYS:
        assert relerr(getattr(post, k), ref[k]) < 1e-10, k
    ref = o.ekf_smoother(mdl, t, y, state_order="first")
    sm = cd.cdnlgssm_smoother(P, y, t[..., None], cd.EKFHyperParams(state_order="first"))
    assert relerr(sm.smoothed_means, ref["smoothed_means"]) < 1e-10
    assert relerr(sm.smoothed_covariances, ref["smoothed_covariances"]) < 1e-10
    ref = o.ekf_filter(mdl, t, y, state_order="first")
    p32 = cd.cdnlgssm_filter(P, y.astype(np.float32), t[..., None].astype(np.float32), cd.EKFHyperParams(state_order="first"))
    assert p32.filtered_means.dtype == np.float32 and relerr(p32.filtered_means, ref["filtered_means"]) < 2e-4
    fc = cd.cdnlgssm_forecast(P, (mdl.m0, mdl.P0), np.array([[0.0]]), np.linspace(0.1, 1.0, 7)[:, None],
                              cd.EKFHyperParams(state_order="first"))
    rm, rP = o.forecast(mdl, mdl.m0[None], mdl.P0[None], np.array([0.0]), np.linspace(0.1, 1.0, 7)[None], "ekf", state_order="first")
    assert relerr(fc.forecasted_state_means, rm[0]) < 1e-11 and relerr(fc.forecasted_state_covariances, rP[0]) < 1e-11


@pytest.mark.gpu
def test_custom_drift_second_order_with_divgrad(hip_lib):
    """Van der Pol with its grad(div f) supplied: the reference's default state_order='second' (0.5 P grad(div f) added to
    the mean ODE, inference_ekf.py:108-116) and the parameter value changing between calls without recompilation."""
    rng = np.random.default_rng(71)
    for mu in (1.5, 0.7):
        theta = np.array([mu])
        mdl = make_model(vdp_oracle(theta), 1)
        N, T = 9, 40
        t = o.irregular_times(rng, N, T, 0.05)
        y = o.simulate(mdl, t, rng)
        P = params_for(mdl, cd.LearnableCustomDrift(theta, VDP_F, VDP_J, VDP_G))
        ref = o.ekf_filter(mdl, t, y, state_order="second")
        post = cd.cdnlgssm_filter(P, y, t[..., None])
        for k in FILTER_KEYS:
            assert relerr(getattr(post, k), ref[k]) < 1e-11, k
        ref1 = o.ekf_filter(mdl, t, y, state_order="first")
        assert relerr(ref1["filtered_means"], ref["filtered_means"]) > 1e-6       # the second-order term is not a no-op here
    # the gradient at the default order (a non-zero grad(div f): no forward-sensitivity kernel) comes from the tangent sweep of the literal
    # recursion (round 5, cdkf_ukf_tangent_kernels.h: ekf_tangent_body) -- against central differences of the filter's own log-likelihood
    ll, g = cd.cdnlgssm_loglik_and_grad(P, y, t[..., None])
    assert _ffi.lib().cdkf_last_kernel().decode().startswith("ekf_tangent_kernel<double>")
    np.testing.assert_allclose(ll, post.marginal_loglik, rtol=1e-10)
    e = 1e-6
    fd = (cd.cdnlgssm_filter(params_for(mdl, cd.LearnableCustomDrift(theta + e, VDP_F, VDP_J, VDP_G)), y, t[..., None]).marginal_loglik
          - cd.cdnlgssm_filter(params_for(mdl, cd.LearnableCustomDrift(theta - e, VDP_F, VDP_J, VDP_G)), y, t[..., None]).marginal_loglik) / (2 * e)
    assert np.abs(np.asarray(g.theta)[:, 0] - fd).max() < 1e-6 * max(1.0, np.abs(fd).max())
    # another Runge-Kutta method through the same run-time compiled kernels
    hyp = cd.EKFHyperParams(diffeqsolve_settings={"solver": "tsit5"})
    with o.use_solver("tsit5"):
        ref = o.ekf_filter(mdl, t, y, state_order="second")
        refs = o.ekf_smoother(mdl, t, y, state_order="second")
    post = cd.cdnlgssm_filter(P, y, t[..., None], hyp)
    assert relerr(post.filtered_covariances, ref["filtered_covariances"]) < 1e-11
    sm = cd.cdnlgssm_smoother(P, y, t[..., None], hyp)
    assert relerr(sm.smoothed_means, refs["smoothed_means"]) < 1e-10
    # ... and adaptive step-size control
    ctrl = dict(rtol=1e-6, atol=1e-8)
    hyp = cd.EKFHyperParams(diffeqsolve_settings={"solver": "dopri5", "dt0": 0.02, "stepsize_controller": cd.PIDController(**ctrl)})
    with o.use_solver("dopri5", adaptive=ctrl):
        ref = o.ekf_filter(mdl, t, y, state_order="second", dt0=0.02)
    post = cd.cdnlgssm_filter(P, y, t[..., None], hyp)
    assert relerr(post.filtered_covariances, ref["filtered_covariances"]) < 1e-10


# ---- user-supplied emission functions ---------------------------------------------------------------------------------------
SIN_H = "hx[0] = eta[0] * sin(x[0]) + eta[1];"
SIN_J = "H[0][0] = eta[0] * cos(x[0]);"
QUAD_H = "hx[0] = eta[0] * x[0] * x[0] + x[2]; hx[1] = eta[1] * x[1] + eta[2] * tanh(x[2]);"
QUAD_J = "H[0][0] = R(2) * eta[0] * x[0]; H[0][2] = R(1); H[1][1] = eta[1]; H[1][2] = eta[2] * (R(1) - tanh(x[2]) * tanh(x[2]));"


def sin_emission():
    h = lambda x, eta: (eta[0] * np.sin(x[..., 0]) + eta[1])[..., None]

    def jac(x, eta):
        H = np.zeros(x.shape[:-1] + (1, 2), x.dtype)
        H[..., 0, 0] = eta[0] * np.cos(x[..., 0])
        return H
    return h, jac


def quad_emission():
    h = lambda x, eta: np.stack([eta[0] * x[..., 0] ** 2 + x[..., 2], eta[1] * x[..., 1] + eta[2] * np.tanh(x[..., 2])], -1)

    def jac(x, eta):
        H = np.zeros(x.shape[:-1] + (2, 3), x.dtype)
        H[..., 0, 0] = 2 * eta[0] * x[..., 0]
        H[..., 0, 2] = 1
        H[..., 1, 1] = eta[1]
        H[..., 1, 2] = eta[2] * (1 - np.tanh(x[..., 2]) ** 2)
        return H
    return h, jac


def test_custom_emission_compiles_without_gpu():
    kd = _ffi.register_custom_drift(2, 2, PEND_F, PEND_J, None)
    ke = _ffi.register_custom_emission(2, 1, SIN_H, SIN_J)
    assert ke >= 1000 and _ffi.register_custom_emission(2, 1, SIN_H, SIN_J) == ke
    L = _ffi.lib()
    for algo in (0, 1, 2):
        assert L.cdkf_custom_drift_compile(kd, 8, 1, algo, 1, ke) == 0, L.cdkf_last_error().decode()
    assert L.cdkf_custom_drift_compile(kd, 8, 2, 0, 1, ke) != 0          # registered for emission_dim 1
    bad = _ffi.register_custom_emission(2, 1, "hx[0] = sinus(x[0]);", SIN_J)
    assert L.cdkf_custom_drift_compile(kd, 4, 1, 0, 1, bad) != 0 and "emission_h:1" in L.cdkf_last_error().decode()


@pytest.mark.gpu
def test_custom_emission_pendulum_sine(hip_lib):
    """The classic non-linear-observation pendulum (angle observed through its sine): EKF with 1 and 3 re-linearisations
    (iterated EKF, inference_ekf.py:183-199), UKF with the sigma points pushed through h (inference_ukf.py:162-203),
    EKF smoother -- against the oracle evaluating the same h and its Jacobian as NumPy callables."""
    rng = np.random.default_rng(81)
    theta, eta = np.array([2.0, 0.3]), np.array([1.5, 0.2])
    mdl = o.Model(pendulum_oracle(theta), np.eye(2), np.array([[0.05, 0.01], [0.01, 0.1]]), eta[None, :], np.zeros(1),
                  0.1 * np.eye(1), np.array([0.7, 0.0]), 0.3 * np.eye(2), emission=sin_emission())
    # eta travels as [H.ravel() | bias]: H = [[1.5, 0.2]], bias = [0]  ->  eta = (1.5, 0.2, 0)
    N, T = 40, 30
    t = o.irregular_times(rng, N, T, 0.08)
    y = o.simulate(mdl, t, rng) if False else (mdl.h(np.cumsum(rng.standard_normal((N, T, 2)) * 0.05, axis=1) + mdl.m0)
                                               + 0.3 * rng.standard_normal((N, T, 1)))
    P = cd.ParamsCDNLGSSM(
        initial=cd.ParamsLGSSMInitial(cd.LearnableVector(mdl.m0), cd.LearnableMatrix(mdl.P0)),
        dynamics=cd.ParamsCDNLGSSMDynamics(cd.LearnableCustomDrift(theta, PEND_F, PEND_J), cd.LearnableMatrix(mdl.L),
                                           cd.LearnableMatrix(mdl.Qc), 2.0),
        emissions=cd.ParamsCDNLGSSMEmissions(cd.LearnableCustomEmission(eta, SIN_H, SIN_J), cd.LearnableMatrix(mdl.R)))
    for num_iter in (1, 3):
        ref = o.ekf_filter(mdl, t, y, state_order="first", num_iter=num_iter)
        post = cd.cdnlgssm_filter(P, y, t[..., None], cd.EKFHyperParams(state_order="first"), num_iter=num_iter)
        for k in FILTER_KEYS:
            assert relerr(getattr(post, k), ref[k]) < 1e-11, (num_iter, k)
        np.testing.assert_allclose(post.marginal_loglik, ref["marginal_loglik"], rtol=1e-11)
        # the emission's Jacobian (and the drift's) derived from h_src / f_src by dual numbers -- jacfwd in the reference
        Pa = P._replace(dynamics=P.dynamics._replace(drift=cd.LearnableCustomDrift(theta, PEND_F)),
                        emissions=P.emissions._replace(emission_function=cd.LearnableCustomEmission(eta, SIN_H)))
        posta = cd.cdnlgssm_filter(Pa, y, t[..., None], cd.EKFHyperParams(state_order="first"), num_iter=num_iter)
        for k in FILTER_KEYS:
            assert relerr(getattr(posta, k), getattr(post, k)) < 1e-13, (num_iter, k)
    refu = o.ukf_filter(mdl, t, y)
    postu = cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams())
    for k in FILTER_KEYS:
        assert relerr(getattr(postu, k), refu[k]) < 1e-10, k
    np.testing.assert_allclose(postu.marginal_loglik, refu["marginal_loglik"], rtol=1e-10)
    refs = o.ekf_smoother(mdl, t, y, state_order="first")
    sm = cd.cdnlgssm_smoother(P, y, t[..., None], cd.EKFHyperParams(state_order="first"))
    assert relerr(sm.smoothed_means, refs["smoothed_means"]) < 1e-10
    p32 = cd.cdnlgssm_filter(P, y.astype(np.float32), t[..., None].astype(np.float32), cd.EKFHyperParams(state_order="first"))
    assert relerr(p32.filtered_means, o.ekf_filter(mdl, t, y, state_order="first")["filtered_means"]) < 5e-4
    # emission moments of the filtered marginals under the source emission (inference_ekf.py:768-855): h(m), H P H^T + R
    ym, yc = cd.cdnlgssm_emissions(P, t[0][:, None], post.filtered_means[0], post.filtered_covariances[0])
    Hm = mdl.Hjac(ref["filtered_means"][0])
    assert relerr(ym, mdl.h(ref["filtered_means"][0])) < 1e-10
    assert relerr(yc, Hm @ ref["filtered_covariances"][0] @ np.swapaxes(Hm, -1, -2) + mdl.R) < 1e-10


@pytest.mark.gpu
def test_custom_emission_with_builtin_drift(hip_lib):
    """A registry drift (Lorenz-63) under a user-defined two-component emission: the host turns the drift into source so
    that the whole model goes through the run-time compiled kernel; default state_order='second'."""
    rng = np.random.default_rng(82)
    eta = np.array([0.05, 1.2, 3.0])
    base = o.lorenz63_model(2)
    Hblk = np.concatenate([eta, np.zeros(3)]).reshape(2, 3)          # eta padded into the [m, d] block, bias = 0
    mdl = o.Model(base.drift, base.L, base.Qc, Hblk, np.zeros(2), base.R, base.m0, base.P0, emission=quad_emission())
    N, T = 20, 25
    t = o.irregular_times(rng, N, T, 0.03)
    y = mdl.h(base.m0 + np.cumsum(rng.standard_normal((N, T, 3)) * 0.3, axis=1)) + rng.standard_normal((N, T, 2))
    P = params_for(base, cd.LearnableLorenz63(10.0, 28.0, 8 / 3))
    P = P._replace(emissions=cd.ParamsCDNLGSSMEmissions(cd.LearnableCustomEmission(eta, QUAD_H, QUAD_J), cd.LearnableMatrix(base.R)))
    ref = o.ekf_filter(mdl, t, y)
    post = cd.cdnlgssm_filter(P, y, t[..., None])
    for k in FILTER_KEYS:
        assert relerr(getattr(post, k), ref[k]) < 1e-10, k
    refu = o.ukf_filter(mdl, t, y)
    postu = cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams())
    assert relerr(postu.filtered_covariances, refu["filtered_covariances"]) < 1e-9
    # the extended filter's gradient under a source emission: the tangent sweep (round 5) against central differences of the filter
    ll, g = cd.cdnlgssm_loglik_and_grad_all(P, y, t[..., None])
    assert _ffi.lib().cdkf_last_kernel().decode().startswith("ekf_tangent_kernel<double>")
    np.testing.assert_allclose(ll, post.marginal_loglik, rtol=1e-10)
    with_eta = lambda ev: P._replace(emissions=cd.ParamsCDNLGSSMEmissions(cd.LearnableCustomEmission(ev, QUAD_H, QUAD_J), cd.LearnableMatrix(base.R)))
    for pidx in range(3):
        e = 1e-6 * np.eye(3)[pidx]
        fd = (cd.cdnlgssm_filter(with_eta(eta + e), y, t[..., None]).marginal_loglik - cd.cdnlgssm_filter(with_eta(eta - e), y, t[..., None]).marginal_loglik) / 2e-6
        got = np.asarray(g.emissions.emission_function.eta)[:, pidx]
        assert np.abs(got - fd).max() < 2e-6 * max(1.0, np.abs(fd).max()), (pidx, got, fd)


# ---- derivatives by dual numbers (round 3): the Jacobian, grad(div f) and the parameter gradient from f_src alone -----------------------
# a drift with every provided function in it: damped, driven, saturating
NL_F = ("auto s = sin(x[0]); auto e = exp(-theta[2] * x[1] * x[1]);"
        "fx[0] = x[1] + theta[1] * tanh(x[0] * x[1]);"
        "fx[1] = -theta[0] * s * e - theta[1] * x[1] + R(0.3) * cos(R(2) * x[0]) / (R(1) + x[0] * x[0]) + sqrt(R(1) + x[1] * x[1]) * pow(theta[2], 2);")
L63_F = "fx[0] = theta[0] * (x[1] - x[0]); fx[1] = x[0] * (theta[1] - x[2]) - x[1]; fx[2] = x[0] * x[1] - theta[2] * x[2];"
L63_J = ("F[0][0] = -theta[0]; F[0][1] = theta[0]; F[1][0] = theta[1] - x[2]; F[1][1] = -R(1); F[1][2] = -x[0];"
         "F[2][0] = x[1]; F[2][1] = x[0]; F[2][2] = -theta[2];")


def test_dual_number_variants_compile_without_a_gpu():
    """jac_src None (Jacobian by dual numbers), divgrad_src "auto" (nested), the gradient sweep (algo 3): every variant goes through
    hipRTC for gfx950 on the CPU box, both precisions; a snippet that pins a temporary to R fails in the dual-number variant only,
    with the compiler's diagnostic."""
    L = _ffi.lib()
    k = _ffi.register_custom_drift(2, 3, NL_F, None, "auto")
    for nbytes in (8, 4):
        for algo in (0, 1, 2):
            assert L.cdkf_custom_drift_compile(k, nbytes, 1, algo, 2, 0) == 0, L.cdkf_last_error().decode()
        assert L.cdkf_custom_drift_compile(k, nbytes, 2, 3, 1, 0) == 0, L.cdkf_last_error().decode()
    k3 = _ffi.register_custom_drift(3, 3, L63_F, L63_J, "")
    assert L.cdkf_custom_drift_compile(k3, 8, 3, 3, 2, 0) == 0, L.cdkf_last_error().decode()
    pinned = _ffi.register_custom_drift(2, 1, "R s = sin(x[0]); fx[0] = x[1]; fx[1] = -theta[0] * s;", "F[0][1] = R(1); F[1][0] = -theta[0] * cos(x[0]);", None)
    assert L.cdkf_custom_drift_compile(pinned, 8, 1, 0, 1, 0) == 0                 # plain filter: R temporaries are fine
    assert L.cdkf_custom_drift_compile(pinned, 8, 1, 3, 1, 0) != 0                 # gradient: f_src is compiled with T = a dual number
    assert "drift_f:1" in L.cdkf_last_error().decode()


@pytest.mark.gpu
def test_custom_drift_derivatives_by_dual_numbers(hip_lib):
    """(1) Lorenz-63 written as a snippet: with the Jacobian derived by dual numbers the filter reproduces the built-in drift's
    numbers (second order too: grad(div f) "auto" = 0), and the log-likelihood gradient w.r.t. (sigma, rho, beta) equals the built-in
    forward-sensitivity kernel's and the oracle's.  (2) A drift with tanh / exp / sqrt / pow / cos in it: derived Jacobian and
    grad(div f) against finite differences of the oracle running the same drift, gradient against central differences of the
    HIP log-likelihood itself."""
    rng = np.random.default_rng(90)
    ref_mdl = o.lorenz63_model(2)
    N, T = 7, 25
    t = o.irregular_times(rng, N, T, 0.3)
    y = o.simulate(ref_mdl, t, rng)
    th = np.array([10.0, 28.0, 8.0 / 3.0])
    from helpers import params_from
    P_builtin = params_from(ref_mdl)
    flt_ref = cd.cdnlgssm_filter(P_builtin, y, t[..., None])
    ll_ref, g_ref = o.ekf_loglik_grad(ref_mdl, t, y)
    for jac, dg in ((L63_J, ""), (None, "auto")):
        P = params_for(ref_mdl, cd.LearnableCustomDrift(th, L63_F, jac, dg))
        flt = cd.cdnlgssm_filter(P, y, t[..., None])
        for k in FILTER_KEYS:
            assert relerr(getattr(flt, k), getattr(flt_ref, k)) < 1e-12, (k, jac is None)
        if dg == "":      # (the gradient sweep carries no second-order term: grad(div f) must be registered as identically zero)
            ll, g = cd.cdnlgssm_loglik_and_grad(P, y, t[..., None])
            np.testing.assert_allclose(ll, ll_ref, rtol=1e-11)
            assert np.abs(np.asarray(g.theta) - g_ref).max() < 1e-9 * np.abs(g_ref).max()
    P1 = params_for(ref_mdl, cd.LearnableCustomDrift(th, L63_F, None, None))
    ll1, g1 = cd.cdnlgssm_loglik_and_grad(P1, y, t[..., None], cd.EKFHyperParams(state_order="first"))   # Jacobian AND gradient from f_src alone
    assert np.abs(np.asarray(g1.theta) - g_ref).max() < 1e-9 * np.abs(g_ref).max()
    # the reference's default state_order 'second' with grad(div f) "auto" (= 0 for this drift): the reverse sweep with third derivatives
    ll2, g2 = cd.cdnlgssm_loglik_and_grad(params_for(ref_mdl, cd.LearnableCustomDrift(th, L63_F, None, "auto")), y, t[..., None])
    assert "ekf_adjoint_wg_kernel" in _ffi.lib().cdkf_last_kernel().decode()
    assert np.abs(np.asarray(g2.theta) - g_ref).max() < 1e-9 * np.abs(g_ref).max()

    # (2) a drift that is not in any registry
    theta = np.array([1.7, 0.25, 0.4])

    def f_np(x, thv):
        s, e = np.sin(x[..., 0]), np.exp(-thv[2] * x[..., 1] ** 2)
        return np.stack([x[..., 1] + thv[1] * np.tanh(x[..., 0] * x[..., 1]),
                         -thv[0] * s * e - thv[1] * x[..., 1] + 0.3 * np.cos(2 * x[..., 0]) / (1 + x[..., 0] ** 2)
                         + np.sqrt(1 + x[..., 1] ** 2) * thv[2] ** 2], -1)

    def jac_np(x, thv, h=1e-6):
        cols = [(f_np(x + h * np.eye(2)[j], thv) - f_np(x - h * np.eye(2)[j], thv)) / (2 * h) for j in range(2)]
        return np.stack(cols, -1)

    def g_np(x, thv, h=1e-4):
        div = lambda z: np.trace(jac_np(z, thv, 1e-5), axis1=-2, axis2=-1)
        return np.stack([(div(x + h * np.eye(2)[j]) - div(x - h * np.eye(2)[j])) / (2 * h) for j in range(2)], -1)

    mdl = make_model(o.CallableDrift(theta, f_np, jac_np, g_np), 1)
    N, T = 6, 30
    t = o.irregular_times(rng, N, T, 0.4)
    y = o.simulate(mdl, t, rng)
    P = params_for(mdl, cd.LearnableCustomDrift(theta, NL_F, None, "auto"))
    for order, tol in (("first", 1e-7), ("second", 2e-5)):   # (the oracle's own derivatives here are finite differences)
        ref = o.ekf_filter(mdl, t, y, state_order=order)
        post = cd.cdnlgssm_filter(P, y, t[..., None], cd.EKFHyperParams(state_order=order))
        assert relerr(post.filtered_means, ref["filtered_means"]) < tol, order
        assert relerr(post.marginal_loglik, ref["marginal_loglik"]) < tol, order
    hyp = cd.EKFHyperParams(state_order="first")
    Pg = params_for(mdl, cd.LearnableCustomDrift(theta, NL_F, None, None))
    ll, g = cd.cdnlgssm_loglik_and_grad(Pg, y, t[..., None], hyp)
    g = np.asarray(g.theta)
    for p in range(3):
        h = 1e-5 * max(1.0, abs(theta[p]))
        lp = cd.cdnlgssm_filter(params_for(mdl, cd.LearnableCustomDrift(theta + h * np.eye(3)[p], NL_F, None, None)), y, t[..., None], hyp).marginal_loglik
        lm = cd.cdnlgssm_filter(params_for(mdl, cd.LearnableCustomDrift(theta - h * np.eye(3)[p], NL_F, None, None)), y, t[..., None], hyp).marginal_loglik
        fd = (lp - lm) / (2 * h)
        assert np.abs(g[:, p] - fd).max() < 1e-6 * max(1.0, np.abs(fd).max()), (p, g[:, p], fd)
    g32 = np.asarray(cd.cdnlgssm_loglik_and_grad(Pg, y.astype(np.float32), t[..., None].astype(np.float32), hyp)[1].theta)
    assert np.abs(g32 - g).max() < 5e-3 * np.abs(g).max()


@pytest.mark.gpu
def test_fit_sgd_on_a_custom_drift(hip_lib):
    """fit_sgd with a run-time compiled drift: the trainable leaf is the drift's theta, value and gradient come from the dual-number
    sweep; Adam lowers the loss and moves the pendulum's stiffness and damping towards the truth."""
    from cd_dynamax_amd import fit
    from cd_dynamax_amd.params import ParameterProperties as PP
    rng = np.random.default_rng(91)
    truth = np.array([2.0, 0.3])
    mdl = make_model(pendulum_oracle(truth), 2)
    N, T = 24, 60
    t = o.irregular_times(rng, N, T, 3.0)
    y = o.simulate(mdl, t, rng)
    model = cd.ContDiscreteNonlinearGaussianSSM(2, 2)
    start = params_for(mdl, cd.LearnableCustomDrift(np.array([1.2, 0.8]), PEND_F, None, None))
    frozen = PP(trainable=False)
    props = cd.ParamsCDNLGSSM(
        initial=cd.ParamsLGSSMInitial(cd.LearnableVector(frozen), cd.LearnableMatrix(frozen)),
        dynamics=cd.ParamsCDNLGSSMDynamics(cd.LearnableCustomDrift(PP(), None, None, None), cd.LearnableMatrix(frozen), cd.LearnableMatrix(frozen), 1.0),
        emissions=cd.ParamsCDNLGSSMEmissions(cd.LearnableLinear(frozen, frozen), cd.LearnableMatrix(frozen)))
    hyp = cd.EKFHyperParams(state_order="first")
    new, losses = model.fit_sgd(start, props, y, t[..., None], hyp, optimizer=fit.Adam(0.05), batch_size=N, num_epochs=80)
    got = np.asarray(new.dynamics.drift.theta)
    assert np.all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert np.all(np.abs(got - truth) < 0.5 * np.abs(np.array([1.2, 0.8]) - truth)), got


# ---- beyond six dimensions: the same source compiled into the workgroup-per-trajectory kernels -------------------------------------
def cubic_l96_src(d):
    """Lorenz-96 with a cubic damping term: f_i = (x_{i+1} - x_{i-2}) x_{i-1} - x_i - theta_1 x_i^3 + theta_0; grad(div f)_k = -6 theta_1 x_k."""
    return (f"for (int i = 0; i < {d}; ++i) {{ const int ip1 = (i + 1) % {d}, im1 = (i + {d - 1}) % {d}, im2 = (i + {d - 2}) % {d}; "
            f"fx[i] = (x[ip1] - x[im2]) * x[im1] - R(1) * x[i] - theta[1] * x[i] * x[i] * x[i] + theta[0]; }}")


def cubic_l96_oracle(theta, d):
    f = lambda x, th: (np.roll(x, -1, -1) - np.roll(x, 2, -1)) * np.roll(x, 1, -1) - x - th[1] * x ** 3 + th[0]

    def jac(x, th):
        F = np.zeros(x.shape + (d,), x.dtype)
        for i in range(d):
            F[..., i, (i + 1) % d] += x[..., (i - 1) % d]
            F[..., i, (i - 2) % d] += -x[..., (i - 1) % d]
            F[..., i, (i - 1) % d] += x[..., (i + 1) % d] - x[..., (i - 2) % d]
            F[..., i, i] += -1 - 3 * th[1] * x[..., i] ** 2
        return F
    g = lambda x, th: -6 * th[1] * x

    def vjp(x, lam, G, th):   # gradient of lam . f + <G, F> w.r.t. (x, theta), by hand
        xb = jac(x[None], th)[0].T @ lam
        for i in range(d):
            ip1, im1, im2 = (i + 1) % d, (i - 1) % d, (i - 2) % d
            xb[im1] += G[i, ip1] - G[i, im2]
            xb[ip1] += G[i, im1]
            xb[im2] -= G[i, im1]
        xb = xb - 6 * th[1] * np.diag(G) * x
        return xb, np.array([lam.sum(), -(lam * x ** 3).sum() - 3 * (np.diag(G) * x ** 2).sum()])
    gvjp = lambda x, u, th: (-6 * th[1] * u, np.array([0.0, -6 * (u * x).sum()]))   # gradient of u . g, g = -6 theta_1 x
    return o.CallableDrift(theta, f, jac, g, vjp=vjp, gvjp=gvjp)


def wide_model(rng, d, m, theta, selection=False):
    H = np.eye(d)[:: max(1, d // m)][:m] if selection else rng.standard_normal((m, d)) / np.sqrt(d)
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    Rm = rng.standard_normal((m, m)) / np.sqrt(m)
    return o.Model(cubic_l96_oracle(theta, d), np.eye(d), 0.05 * np.eye(d) + 0.02 * A @ A.T, H, 0.05 * rng.standard_normal(m),
                   0.3 * np.eye(m) + 0.05 * Rm @ Rm.T, theta[0] + 0.3 * rng.standard_normal(d), 0.2 * np.eye(d))


def test_wide_custom_drift_compiles_without_a_gpu():
    """state_dim 12: the workgroup kernels go through hipRTC with the drift's source for gfx950 on the CPU box (filter with grad(div f)
    by nested dual numbers, smoother, reverse sweep); a hand-written Jacobian is refused at registration."""
    L = _ffi.lib()
    k = _ffi.register_custom_drift(12, 2, cubic_l96_src(12), None, "auto")
    assert L.cdkf_custom_drift_compile(k, 8, 5, 2, 2, 0) == 0, L.cdkf_last_error().decode()       # filter + smoother, fp64
    assert L.cdkf_custom_drift_compile(k, 4, 12, 1, 1, 0) == 0, L.cdkf_last_error().decode()      # unscented filter, fp32
    assert L.cdkf_custom_drift_compile(k, 8, 5, 3, 1, 0) == 0, L.cdkf_last_error().decode()       # the reverse sweep with the drift compiled in
    assert L.cdkf_custom_drift_compile(k, 8, 50, 3, 1, 0) != 0                                    # ... has an LDS plan of its own (q <= 43 in fp64)
    bad = _ffi.register_custom_drift(12, 1, "for (int i = 0; i < 12; ++i) fx[i] = -theta[0] * sine(x[i]);", None, None)
    assert L.cdkf_custom_drift_compile(bad, 8, 5, 0, 1, 0) != 0 and "drift_f:1" in L.cdkf_last_error().decode()


@pytest.mark.gpu
@pytest.mark.parametrize("d,m,selection", [(12, 5, False), (9, 9, True), (7, 3, False), (3, 8, False)])
def test_wide_custom_drift_filters_and_smoother(hip_lib, d, m, selection):
    """A drift given as source beyond the register-resident kernels' six dimensions (state or emission): Jacobian by dual numbers, a
    thread per direction; grad(div f) by nested dual numbers, a thread per (i, k) pair; the unscented filter's sigma points a thread
    per pair -- EKF first / second / zeroth order, UKF, smoother, fp32, forecast, another Runge-Kutta method and the step-size
    controller, against the oracle evaluating the same drift through NumPy callables."""
    rng = np.random.default_rng(800 + 10 * d + m)
    theta = np.array([4.0, 0.05])
    mdl = wide_model(rng, d, m, theta, selection)
    N, T = 5, 14
    t = o.irregular_times(rng, N, T, 0.04)
    y = o.simulate(mdl, t, rng)
    P = params_for(mdl, cd.LearnableCustomDrift(theta, cubic_l96_src(d), None, "auto"))
    for order in ("second", "first", "zeroth"):
        ref = o.ekf_filter(mdl, t, y, state_order=order)
        post = cd.cdnlgssm_filter(P, y, t[..., None], cd.EKFHyperParams(state_order=order))
        for k in FILTER_KEYS:
            assert relerr(getattr(post, k), ref[k]) < 1e-10, (order, k)
        np.testing.assert_allclose(post.marginal_loglik, ref["marginal_loglik"], rtol=1e-10)
        if order == "second":
            assert "custom drift" in _ffi.lib().cdkf_last_kernel().decode()
            ref1 = o.ekf_filter(mdl, t, y, state_order="first")
            assert relerr(ref1["filtered_means"], ref["filtered_means"]) > 1e-7     # the second-order term is not a no-op here
    ref = o.ukf_filter(mdl, t, y)
    post = cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams())
    for k in FILTER_KEYS:
        assert relerr(getattr(post, k), ref[k]) < 1e-9, k
    ref = o.ekf_smoother(mdl, t, y, state_order="second")
    sm = cd.cdnlgssm_smoother(P, y, t[..., None])
    assert relerr(sm.smoothed_means, ref["smoothed_means"]) < 1e-9
    assert relerr(sm.smoothed_covariances, ref["smoothed_covariances"]) < 1e-9
    ref = o.ekf_filter(mdl, t, y, state_order="second")
    p32 = cd.cdnlgssm_filter(P, y.astype(np.float32), t[..., None].astype(np.float32))
    assert p32.filtered_means.dtype == np.float32 and relerr(p32.filtered_means, ref["filtered_means"]) < 5e-4
    tf = np.linspace(0.05, 0.4, 5)
    fc = cd.cdnlgssm_forecast(P, (mdl.m0, mdl.P0), np.array([[0.0]]), tf[:, None], cd.EKFHyperParams(state_order="first"))
    rm, rP = o.forecast(mdl, mdl.m0[None], mdl.P0[None], np.array([0.0]), tf[None], "ekf", state_order="first")
    assert relerr(fc.forecasted_state_means, rm[0]) < 1e-10 and relerr(fc.forecasted_state_covariances, rP[0]) < 1e-10
    hyp = cd.EKFHyperParams(diffeqsolve_settings={"solver": "tsit5"})
    with o.use_solver("tsit5"):
        ref = o.ekf_filter(mdl, t, y, state_order="second")
    post = cd.cdnlgssm_filter(P, y, t[..., None], hyp)
    assert relerr(post.filtered_covariances, ref["filtered_covariances"]) < 1e-10
    ctrl = dict(rtol=1e-6, atol=1e-8)
    hyp = cd.EKFHyperParams(diffeqsolve_settings={"solver": "dopri5", "dt0": 0.02, "stepsize_controller": cd.PIDController(**ctrl)})
    with o.use_solver("dopri5", adaptive=ctrl):
        ref = o.ekf_filter(mdl, t, y, state_order="second", dt0=0.02)
    post = cd.cdnlgssm_filter(P, y, t[..., None], hyp)
    assert relerr(post.filtered_covariances, ref["filtered_covariances"]) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("d,m,selection", [(12, 5, False), (9, 9, True), (7, 10, False), (2, 1, False)])
def test_wide_custom_drift_loglik_gradient(hip_lib, d, m, selection):
    """d ll / d theta and every other leaf for a drift given as source, on the shape-generic reverse sweep compiled with it at run time:
    what the reverse of F(m, theta) Ps + (F Ps)^T needs -- second derivatives of f contracted with G = 2 Lam Ps -- comes from nested dual
    numbers, one evaluation per (Jacobian column, state component or parameter).  Against the oracle's reverse sweep with the drift's
    vector-Jacobian products written out by hand, and against central differences of the HIP log-likelihood."""
    rng = np.random.default_rng(900 + 10 * d + m)
    theta = np.array([4.0, 0.05])
    if d < 4:   # (the cubic Lorenz-96 needs four components: a damped oscillator with the same two parameters' roles)
        src = "fx[0] = x[1] + theta[0]; fx[1] = -x[0] - theta[1] * x[1] * x[1] * x[1];"
        f = lambda x, th: np.stack([x[..., 1] + th[0], -x[..., 0] - th[1] * x[..., 1] ** 3], -1)

        def jac(x, th):
            F = np.zeros(x.shape + (2,), x.dtype)
            F[..., 0, 1] = 1
            F[..., 1, 0] = -1
            F[..., 1, 1] = -3 * th[1] * x[..., 1] ** 2
            return F

        def vjp(x, lam, G, th):
            xb = jac(x[None], th)[0].T @ lam
            xb[1] += -6 * th[1] * x[1] * G[1, 1]
            return xb, np.array([lam[0], -lam[1] * x[1] ** 3 - 3 * G[1, 1] * x[1] ** 2])
        drift = o.CallableDrift(theta, f, jac, None, vjp=vjp)
        mdl = wide_model(rng, 4, m, theta, selection)
        mdl = o.Model(drift, np.eye(2), mdl.Qc[:2, :2], mdl.H[:, :2], mdl.bias, mdl.R, np.array([0.5, -0.2]), 0.3 * np.eye(2))
    else:
        src = cubic_l96_src(d)
        mdl = wide_model(rng, d, m, theta, selection)
    N, T = 4, 9
    t = o.irregular_times(rng, N, T, 0.04)
    y = o.simulate(mdl, t, rng)
    P = params_for(mdl, cd.LearnableCustomDrift(theta, src, None, None))
    hyp = cd.EKFHyperParams(state_order="first")
    ll_ref, g_ref, full = o.ekf_loglik_grad_adjoint(mdl, t, y, full=True, state_order="first")
    ll, grads = cd.cdnlgssm_loglik_and_grad_all(P, y, t[..., None], hyp)
    assert "ekf_adjoint_wg_kernel" in _ffi.lib().cdkf_last_kernel().decode() and "custom drift" in _ffi.lib().cdkf_last_kernel().decode()
    np.testing.assert_allclose(ll, ll_ref, rtol=1e-10)
    scale = np.abs(g_ref).max()
    assert np.abs(np.asarray(grads.dynamics.drift.theta) - g_ref).max() < 1e-8 * scale
    for got, want in ((grads.initial.mean.params, full["m0"]), (grads.initial.cov.params, full["P0"]),
                      (grads.dynamics.diffusion_coefficient.params, full["L"]), (grads.dynamics.diffusion_cov.params, full["Qc"]),
                      (grads.emissions.emission_function.weights, full["H"]), (grads.emissions.emission_function.bias, full["bias"]),
                      (grads.emissions.emission_cov.params, full["R"])):
        assert np.abs(np.asarray(got) - want).max() < 1e-8 * max(scale, np.abs(want).max())
    if d > 6 or m > 6:   # the drift block alone takes the same sweep up there (below: the forward-sensitivity kernel)
        ll2, g2 = cd.cdnlgssm_loglik_and_grad(P, y, t[..., None], hyp)
        assert np.abs(np.asarray(g2.theta) - g_ref).max() < 1e-8 * scale
    for p in range(2):   # central differences of the HIP log-likelihood itself
        h = 1e-5 * max(1.0, abs(theta[p]))
        lls = []
        for sgn in (1, -1):
            th = theta.copy()
            th[p] += sgn * h
            lls.append(cd.cdnlgssm_filter(params_for(mdl, cd.LearnableCustomDrift(th, src, None, None)), y, t[..., None], hyp).marginal_loglik)
        fd = (lls[0] - lls[1]) / (2 * h)
        assert np.abs(np.asarray(grads.dynamics.drift.theta)[:, p] - fd).max() < 2e-5 * max(1.0, np.abs(fd).max())
    if d >= 4:   # the reference's default state_order = 'second' with a non-zero grad(div f) = -6 theta_1 x: third derivatives of f, triply nested duals
        ll_ref, g_ref, full = o.ekf_loglik_grad_adjoint(mdl, t, y, full=True, state_order="second")
        ll, grads = cd.cdnlgssm_loglik_and_grad_all(params_for(mdl, cd.LearnableCustomDrift(theta, src, None, "auto")), y, t[..., None])
        np.testing.assert_allclose(ll, ll_ref, rtol=1e-10)
        scale = np.abs(g_ref).max()
        assert np.abs(np.asarray(grads.dynamics.drift.theta) - g_ref).max() < 1e-8 * scale
        for got, want in ((grads.initial.mean.params, full["m0"]), (grads.initial.cov.params, full["P0"]), (grads.dynamics.diffusion_cov.params, full["Qc"]),
                          (grads.emissions.emission_function.weights, full["H"]), (grads.emissions.emission_cov.params, full["R"])):
            assert np.abs(np.asarray(got) - want).max() < 1e-8 * max(scale, np.abs(want).max())
        ll1, _, _ = o.ekf_loglik_grad_adjoint(mdl, t, y, full=True, state_order="first")
        assert np.abs(ll1 - ll_ref).max() > 1e-9 * np.abs(ll_ref).max()     # (the second-order term is not a no-op here)
    with pytest.raises(NotImplementedError):   # without grad(div f) the default order is refused, as for the filter
        cd.cdnlgssm_loglik_and_grad_all(params_for(mdl, cd.LearnableCustomDrift(theta, src, None, None)), y, t[..., None])


@pytest.mark.gpu
def test_scalar_custom_drift_gradient_of_every_leaf(hip_lib):
    """The smallest shape, d = m = 1 (f = theta_0 - theta_1 x^3): a slot of the reverse sweep's LDS plan holds ONE real there, so the
    partial sums of the second-derivative contraction meet in a vector instead -- scripts/gpu_fuzz_custom.py found the shape refused."""
    rng = np.random.default_rng(611)
    theta = np.array([0.4, 0.3])
    f = lambda x, th: th[0] - th[1] * x ** 3
    jac = lambda x, th: (-3 * th[1] * x ** 2)[..., None]
    g = lambda x, th: -6 * th[1] * x
    vjp = lambda x, lam, G, th: (-3 * th[1] * x ** 2 * lam - 6 * th[1] * x * G[0], np.array([lam[0], -lam[0] * x[0] ** 3 - 3 * G[0, 0] * x[0] ** 2]))
    gvjp = lambda x, u, th: (-6 * th[1] * u, np.array([0.0, -6 * u[0] * x[0]]))
    mdl = o.Model(o.CallableDrift(theta, f, jac, g, vjp=vjp, gvjp=gvjp), np.eye(1), 0.2 * np.eye(1), np.array([[0.8]]), np.array([0.1]), 0.3 * np.eye(1),
                  np.array([0.5]), 0.4 * np.eye(1))
    N, T = 5, 9
    t = o.irregular_times(rng, N, T, 0.3)
    y = o.simulate(mdl, t, rng)
    for order, gsrc in (("first", None), ("second", "auto")):
        P = params_for(mdl, cd.LearnableCustomDrift(theta, "fx[0] = theta[0] - theta[1] * x[0] * x[0] * x[0];", None, gsrc))
        ll_ref, g_ref, full = o.ekf_loglik_grad_adjoint(mdl, t, y, full=True, state_order=order)
        ll, grads = cd.cdnlgssm_loglik_and_grad_all(P, y, t[..., None], cd.EKFHyperParams(state_order=order))
        np.testing.assert_allclose(ll, ll_ref, rtol=1e-10)
        scale = np.abs(g_ref).max()
        assert np.abs(np.asarray(grads.dynamics.drift.theta) - g_ref).max() < 1e-8 * scale, order
        for got, want in ((grads.initial.mean.params, full["m0"]), (grads.initial.cov.params, full["P0"]), (grads.dynamics.diffusion_cov.params, full["Qc"]),
                          (grads.emissions.emission_function.weights, full["H"]), (grads.emissions.emission_function.bias, full["bias"]),
                          (grads.emissions.emission_cov.params, full["R"])):
            assert np.abs(np.asarray(got) - want).max() < 1e-8 * max(scale, np.abs(want).max()), order


@pytest.mark.gpu
def test_custom_drift_gradient_with_a_long_snippet(hip_lib):
    """The case scripts/gpu_fuzz_custom.py (seed 12, case 0) caught: a 24-statement quadratic drift at d = 24, m = 22, intervals of
    several Runge-Kutta steps -- the run-time compiled reverse sweep returned d ll / d theta = 0.05 where it is 0.22: only the last
    reversed step's share, the rest lost from a register two lanes kept the sum in across the steps (the sum
    goes straight into the result array since).  Filter and smoother of the same drift along the way."""
    rng = np.random.default_rng(12)
    rng.random()                                   # (the fuzzer's draws, in its order)
    d = int(rng.integers(7, 25))
    m = int(rng.integers(1, min(d + 4, 24)))
    assert (d, m) == (24, 22)
    src, make = random_quadratic_drift(rng, d)
    theta = np.array([0.5 + 0.5 * rng.random(), 0.2 * rng.standard_normal()])
    spd = lambda n, sc: (lambda A: A @ A.T / n * sc + 0.3 * np.eye(n))(rng.standard_normal((n, n)))
    if rng.random() < 0.4 and m <= d:
        H, bias = np.eye(d)[rng.permutation(d)[:m]], np.zeros(m)
    else:
        H, bias = rng.standard_normal((m, d)) / np.sqrt(d), 0.1 * rng.standard_normal(m)
    mdl = o.Model(make(theta), np.eye(d) + 0.1 * rng.standard_normal((d, d)), spd(d, 0.3), H, bias, spd(m, 0.5), 0.5 * rng.standard_normal(d), spd(d, 0.3))
    N, T = int(rng.integers(1, 5)), int(rng.integers(2, 9))
    assert (N, T) == (2, 8)
    t = o.irregular_times(rng, N, T, 0.03 * T)
    y = o.simulate(mdl, t, rng)
    P = params_for(mdl, cd.LearnableCustomDrift(theta, src, None, "auto"))
    ref = o.ekf_smoother(mdl, t, y)
    sm = cd.cdnlgssm_smoother(P, y, t[..., None])
    assert relerr(sm.filtered_covariances, ref["filtered_covariances"]) < 1e-10 and relerr(sm.smoothed_means, ref["smoothed_means"]) < 1e-9
    ll_ref, g_ref, full = o.ekf_loglik_grad_adjoint(mdl, t, y, full=True, state_order="first")
    Pn = params_for(mdl, cd.LearnableCustomDrift(theta, src, None, None))
    ll, grads = cd.cdnlgssm_loglik_and_grad_all(Pn, y, t[..., None], cd.EKFHyperParams(state_order="first"))
    np.testing.assert_allclose(ll, ll_ref, rtol=1e-10)
    scale = np.abs(g_ref).max()
    assert np.abs(np.asarray(grads.dynamics.drift.theta) - g_ref).max() < 1e-8 * scale
    assert np.abs(np.asarray(grads.emissions.emission_function.weights) - full["H"]).max() < 1e-8 * np.abs(full["H"]).max()
    assert np.abs(np.asarray(grads.initial.cov.params) - full["P0"]).max() < 1e-8 * np.abs(full["P0"]).max()


@pytest.mark.gpu
def test_lorenz96_as_source_matches_the_built_in_drift_at_forty_dimensions(hip_lib):
    """Lorenz-96 at d = 40 written as a snippet (theta_1 = 0 removes the cubic term): the run-time compiled workgroup kernels reproduce
    the built-in drift's filter and smoother (the wavefront kernels of config 4) and the oracle."""
    rng = np.random.default_rng(840)
    d, m = 40, 20
    theta = np.array([8.0, 0.0])
    mdl = wide_model(rng, d, m, theta, True)
    builtin = o.Model(o.Lorenz96Drift(8.0), mdl.L, mdl.Qc, mdl.H, mdl.bias, mdl.R, mdl.m0, mdl.P0)
    N, T = 3, 10
    t = o.irregular_times(rng, N, T, 0.03)
    y = o.simulate(builtin, t, rng)
    Pc = params_for(mdl, cd.LearnableCustomDrift(theta, cubic_l96_src(d), None, ""))
    Pb = params_for(builtin, cd.LearnableLorenz96(8.0))
    ref = o.ekf_smoother(builtin, t, y)
    for P in (Pc, Pb):
        sm = cd.cdnlgssm_smoother(P, y, t[..., None])
        assert relerr(sm.filtered_covariances, ref["filtered_covariances"]) < 1e-10
        assert relerr(sm.smoothed_means, ref["smoothed_means"]) < 1e-9
        np.testing.assert_allclose(sm.marginal_loglik, ref["marginal_loglik"], rtol=1e-10)
    assert "custom drift" not in _ffi.lib().cdkf_last_kernel().decode()


RTC_CACHE_WORKER = r'''
import os, sys, time
sys.path[:0] = [os.environ["CDKF_ROOT"]]
from cd_dynamax_amd import _ffi
L = _ffi.lib()
k = _ffi.register_custom_drift(2, 2, "fx[0] = x[1];\nfx[1] = -theta[0] * sin(x[0]) - theta[1] * x[1] * R(1.5);", None, None)
t0 = time.perf_counter()
assert L.cdkf_custom_drift_compile(k, 8, 1, 0, 1, 0) == 0, L.cdkf_last_error().decode()   # lane-per-trajectory filter, fp64
k7 = _ffi.register_custom_drift(7, 1, "for (int i = 0; i < 7; ++i) fx[i] = -theta[0] * x[i] * x[(i + 1) % 7];", None, None)
assert L.cdkf_custom_drift_compile(k7, 8, 3, 0, 1, 0) == 0, L.cdkf_last_error().decode()  # workgroup filter (lowered name kept)
sys.stdout.write("SECONDS %.3f\n" % (time.perf_counter() - t0))
'''


def test_rtc_code_objects_are_cached_on_disk(tmp_path):
    """hipRTC code objects persist (key: generated source + options + target + hipRTC version + the content of every
    kernel header).  Cold process: entries appear; warm process: the same calls return without compiling; a damaged entry is recompiled
    over; CDKF_RTC_CACHE=0 writes nothing.  No GPU needed (cdkf_custom_drift_compile names the target itself)."""
    import subprocess, sys
    script = tmp_path / "rtc_cache_worker.py"
    script.write_text(RTC_CACHE_WORKER)
    cache = tmp_path / "cache"
    cache.mkdir()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(**extra):
        env = dict(os.environ, CDKF_ROOT=root, CDKF_RTC_CACHE_DIR=str(cache), CDKF_RTC_CACHE="1", CDKF_RTC_POLICY="o1")   # (-O1: what is under test is the cache, not the optimiser)
        env.update(extra)
        p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout + p.stderr
        return float(p.stdout.split("SECONDS")[1])

    cold = run()
    assert "MANIFEST" in os.listdir(cache) and len(open(cache / "MANIFEST").read().splitlines()) == 2   # (round 5: what each object is)
    entries = sorted(e for e in os.listdir(cache) if e != "MANIFEST")
    assert len(entries) == 2 and all(e.endswith(".co") for e in entries), entries
    blobs = [open(cache / e, "rb").read() for e in entries]
    warm = run()
    assert warm < 0.5 * cold and warm < 2.0, (cold, warm)
    assert [open(cache / e, "rb").read() for e in entries] == blobs            # untouched by the warm run
    with open(cache / entries[0], "r+b") as f:                                # a damaged entry: bad magic
        f.write(b"XXXX")
    run()
    assert open(cache / entries[0], "rb").read() == blobs[0]                  # recompiled over, bit-identical code object
    for e in entries + ["MANIFEST"]:
        os.remove(cache / e)
    run(CDKF_RTC_CACHE="0")
    assert os.listdir(cache) == []
    os.chmod(cache, 0o777)                                                    # a directory others may write to is not a cache
    run()
    assert os.listdir(cache) == []


@pytest.mark.gpu
@pytest.mark.parametrize("d,m", [(6, 1), (4, 3), (3, 2)])
def test_forward_sensitivities_of_a_source_drift_with_powers(hip_lib, d, m):
    """What scripts/gpu_fuzz_custom.py (seed 40404, case 7) caught in round 4: `pow(x[j], 2)` in a drift's source and the drift-only
    gradient on the register-resident forward-sensitivity kernel (d, m <= 6) -- d ll / d theta was wrong by a factor (the all-leaf
    reverse sweep of the same source, and the same source with `x[j] * x[j]` or `pow(x[j], 2.0)`, were right).  cdkf_dual.h now forms
    a^c for a small whole c by multiplications (every order of derivative the nested dual numbers ask for, no call into the maths
    library from the run-time compiled kernel); here the three spellings against the oracle and each other."""
    import re
    src = None
    for seed in range(400):   # a drift whose source squares a component through pow(): the helper's style 1 with a j == k term
        rng = np.random.default_rng(5000 + seed)
        s_, make = random_quadratic_drift(rng, d)
        if "pow(" in s_:
            src = s_
            break
    assert src is not None
    theta = np.array([0.7, -0.15])
    A = rng.standard_normal((d, d))
    B = rng.standard_normal((m, m))
    mdl = o.Model(make(theta), np.eye(d) + 0.1 * rng.standard_normal((d, d)), A @ A.T / d * 0.3 + 0.3 * np.eye(d),
                  rng.standard_normal((m, d)) / np.sqrt(d), 0.1 * rng.standard_normal(m), B @ B.T / m * 0.5 + 0.3 * np.eye(m),
                  0.5 * rng.standard_normal(d), 0.3 * np.eye(d))
    N, T = 3, 6
    t = o.irregular_times(rng, N, T, 0.03 * T)
    y = o.simulate(mdl, t, rng)
    hyp = cd.EKFHyperParams(state_order="first")
    ll_ref, g_ref = o.ekf_loglik_grad_adjoint(mdl, t, y, state_order="first")
    variants = {"pow(x, 2)": src, "pow(x, 2.0)": re.sub(r"pow\((x\[\d+\]), 2\)", r"pow(\1, 2.0)", src),
                "x * x": re.sub(r"pow\((x\[\d+\]), 2\)", r"(\1 * \1)", src)}
    for name, s_ in variants.items():
        P = params_for(mdl, cd.LearnableCustomDrift(theta, s_, None, None))
        ll, g1 = cd.cdnlgssm_loglik_and_grad(P, y, t[..., None], hyp)
        np.testing.assert_allclose(ll, ll_ref, rtol=1e-11, err_msg=name)
        assert np.abs(np.asarray(g1.theta) - g_ref).max() < 1e-9 * np.abs(g_ref).max(), name
        ll2, g2 = cd.cdnlgssm_loglik_and_grad_all(P, y, t[..., None], hyp)
        assert np.abs(np.asarray(g2.dynamics.drift.theta) - g_ref).max() < 1e-9 * np.abs(g_ref).max(), name
