"""User-supplied drifts compiled at run time (cdkf_custom_drift_register; reference: any callable as
ParamsCDNLGSSMDynamics.drift, cdnlgssm_utils.py:38-61).  CPU: registration, hipRTC compilation of every variant (no GPU
needed), diagnostics for broken snippets.  GPU: parity with the oracle running the same drift as NumPy callables."""
import numpy as np
import pytest

import cd_dynamax_amd as cd
import cdkf_oracle as o
from cd_dynamax_amd import _ffi
from helpers import FILTER_KEYS, relerr

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
                assert L.cdkf_custom_drift_compile(kind, nbytes, 1, algo, 1) == 0, L.cdkf_last_error().decode()
    assert L.cdkf_custom_drift_compile(k2, 8, 3, 0, 0) == 0                # zeroth order, emission_dim 3
    with pytest.raises(_ffi.CdkfError):
        _ffi.register_custom_drift(9, 1, PEND_F, PEND_J, None)             # state_dim > 6
    assert L.cdkf_custom_drift_compile(12345, 8, 1, 0, 1) != 0


def test_broken_snippet_reports_the_compiler_diagnostic():
    kind = _ffi.register_custom_drift(2, 1, "fx[0] = x[1]; fx[1] = -theta[0] * sine(x[0]);", "F[0][1] = R(1);", None)
    L = _ffi.lib()
    assert L.cdkf_custom_drift_compile(kind, 8, 1, 0, 1) != 0
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
    with pytest.raises(NotImplementedError, match="no gradient kernel"):
        cd.cdnlgssm_loglik_and_grad(P, y, t[..., None])
    # another Runge-Kutta method through the same run-time compiled kernels
    hyp = cd.EKFHyperParams(diffeqsolve_settings={"solver": "tsit5"})
    with o.use_solver("tsit5"):
        ref = o.ekf_filter(mdl, t, y, state_order="second")
        refs = o.ekf_smoother(mdl, t, y, state_order="second")
    post = cd.cdnlgssm_filter(P, y, t[..., None], hyp)
    assert relerr(post.filtered_covariances, ref["filtered_covariances"]) < 1e-11
    sm = cd.cdnlgssm_smoother(P, y, t[..., None], hyp)
    assert relerr(sm.smoothed_means, refs["smoothed_means"]) < 1e-10
