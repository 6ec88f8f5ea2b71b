"""fit_sgd (reference: ssm_temissions.py:492-600, utils/optimize_utils.py:48-140): host logic on CPU, the value-and-gradient
path on the GPU against the oracle."""
import numpy as np
import pytest

import cd_dynamax_amd as cd
import cdkf_oracle as o
from cd_dynamax_amd import fit
from cd_dynamax_amd.params import ParameterProperties as PP
from helpers import params_from


def _l63_problem(m=3):
    """The set-up of the reference's SGD timer (test_scripts/timers/timer_sgd.py:38-80): Lorenz-63 drift trainable,
    everything else frozen."""
    model = cd.ContDiscreteNonlinearGaussianSSM(3, m)
    frozen = PP(trainable=False)
    params, props = model.initialize(
        key=0,
        initial_mean={"params": np.zeros(3), "props": frozen},
        initial_cov={"params": 100 * np.eye(3), "props": frozen},
        dynamics_drift={"params": cd.LearnableLorenz63(10.0, 28.0, 8 / 3), "props": cd.LearnableLorenz63(PP(), PP(), PP())},
        dynamics_diffusion_coefficient={"params": cd.LearnableMatrix(np.eye(3)), "props": cd.LearnableMatrix(frozen)},
        dynamics_diffusion_cov={"params": cd.LearnableMatrix(np.eye(3)), "props": cd.LearnableMatrix(frozen)},
        emission_function={"params": cd.LearnableLinear(np.eye(m, 3), np.zeros(m)), "props": cd.LearnableLinear(frozen, frozen)},
        emission_cov={"params": cd.LearnableMatrix(np.eye(m)), "props": cd.LearnableMatrix(frozen)},
    )
    return model, params, props


def test_adam_matches_optax_formula():
    """optax.adam: mu = b1 mu + (1-b1) g; nu = b2 nu + (1-b2) g^2; update = -lr mu_hat / (sqrt(nu_hat) + eps)."""
    opt = fit.Adam(0.1)
    th = np.array([1.0, -2.0])
    st = opt.init(th)
    g1 = np.array([0.5, -4.0])
    u1, st = opt.update(g1, st)
    np.testing.assert_allclose(u1, -0.1 * g1 / (np.abs(g1) + 1e-8), rtol=1e-12)  # first step: -lr * sign(g)
    g2 = np.array([0.25, 1.0])
    u2, st = opt.update(g2, st)
    mu = 0.9 * 0.1 * g1 + 0.1 * g2
    nu = 0.999 * 0.001 * g1 ** 2 + 0.001 * g2 ** 2
    np.testing.assert_allclose(u2, -0.1 * (mu / (1 - 0.81)) / (np.sqrt(nu / (1 - 0.999 ** 2)) + 1e-8), rtol=1e-12)
    u, _ = fit.SGD(0.5).update(g1, None)
    np.testing.assert_allclose(u, -0.5 * g1)


def test_trainable_leaves_flatten_and_refusals():
    from cd_dynamax_amd.bijectors import RealToPSDBijector
    model, params, props = _l63_problem()
    tr = fit._Trainable(params, props)
    assert [p for p, _, _, _ in tr.items] == ["dynamics.drift.sigma", "dynamics.drift.rho", "dynamics.drift.beta"]
    assert tr.drift_only and tr.size == 3
    u = tr.to_unconstrained(params)
    np.testing.assert_allclose(u, [10.0, 28.0, 8 / 3])
    assert tr.from_unconstrained(params, u + 1.0).dynamics.drift == cd.LearnableLorenz63(11.0, 29.0, 8 / 3 + 1.0)
    # a frozen drift leaf drops out; a trainable covariance with the PSD bijector joins in unconstrained form
    p2 = props._replace(dynamics=props.dynamics._replace(drift=cd.LearnableLorenz63(PP(), PP(False), PP())),
                        emissions=props.emissions._replace(emission_cov=cd.LearnableMatrix(PP(constrainer=RealToPSDBijector()))))
    tr2 = fit._Trainable(params, p2)
    assert not tr2.drift_only and tr2.size == 2 + 6
    u2 = tr2.to_unconstrained(params)
    back = tr2.from_unconstrained(params, u2)
    np.testing.assert_allclose(back.emissions.emission_cov.params, params.emissions.emission_cov.params, atol=1e-14)
    assert back.dynamics.drift.rho == params.dynamics.drift.rho
    p4 = props._replace(dynamics=props.dynamics._replace(drift=cd.LearnableLorenz63(PP(constrainer=object()), PP(), PP())))
    with pytest.raises(NotImplementedError, match="constrainer"):
        fit._Trainable(params, p4)
    frozen = PP(False)
    with pytest.raises(ValueError, match="no trainable"):
        fit._Trainable(params, props._replace(dynamics=props.dynamics._replace(drift=cd.LearnableLorenz63(frozen, frozen, frozen))))
    th = fit._drift_theta(params.dynamics.drift)
    assert fit._drift_from_theta(params.dynamics.drift, th) == params.dynamics.drift


def test_psd_bijector_restates_tfp_chain():
    """dynamax RealToPSDBijector = CholeskyOuterProduct o TransformDiagonal(Exp) o FillTriangular (bijectors.py:21-35).
    fill_triangular against the example in TFP's documentation; inverse and vector-Jacobian product against the forward."""
    from cd_dynamax_amd.bijectors import RealToPSDBijector, fill_triangular, fill_triangular_inverse
    np.testing.assert_array_equal(fill_triangular(np.arange(1.0, 7.0)), [[4, 0, 0], [6, 5, 0], [3, 2, 1]])
    np.testing.assert_array_equal(fill_triangular_inverse(fill_triangular(np.arange(1.0, 11.0))), np.arange(1.0, 11.0))
    b = RealToPSDBijector()
    rng = np.random.default_rng(0)
    x = rng.standard_normal(10)
    P = b.forward(x)
    assert np.allclose(P, P.T) and np.linalg.eigvalsh(P).min() > 0
    np.testing.assert_allclose(b.inverse(P), x, atol=1e-12)
    L = fill_triangular(x)
    L[np.arange(4), np.arange(4)] = np.exp(np.diag(L))
    np.testing.assert_allclose(P, L @ L.T, rtol=1e-14)
    G = rng.standard_normal((4, 4))
    fd = np.array([((b.forward(x + 1e-6 * e) - b.forward(x - 1e-6 * e)) / 2e-6 * G).sum() for e in np.eye(10)])
    np.testing.assert_allclose(b.forward_vjp(x, G), fd, rtol=1e-7, atol=1e-8)


@pytest.mark.gpu
def test_fit_sgd_first_step_matches_oracle_gradient(hip_lib):
    """One epoch, two minibatches of plain SGD: losses and parameter updates from the oracle's value-and-gradient."""
    model, params, props = _l63_problem(m=1)
    rng = np.random.default_rng(3)
    mdl = o.lorenz63_model(1)
    mdl = o.Model(mdl.drift, np.eye(3), np.eye(3), np.eye(1, 3), np.zeros(1), np.eye(1), np.zeros(3), 100 * np.eye(3))
    N, T, lr, bs = 6, 30, 0.05, 4
    t = o.irregular_times(rng, N, T, 0.05)
    y = o.simulate(mdl, t, rng)
    new, losses, ph, gh = model.fit_sgd(params, props, y, t[..., None], cd.EKFHyperParams(), optimizer=fit.SGD(lr),
                                        batch_size=bs, num_epochs=1, return_param_history=True, return_grad_history=True)
    th = mdl.drift.theta().copy()
    exp_losses = []
    for lo, hi in ((0, 4), (4, 6)):
        cur = o.Model(o.Lorenz63Drift(*th), mdl.L, mdl.Qc, mdl.H, mdl.bias, mdl.R, mdl.m0, mdl.P0)
        ll, g = o.ekf_loglik_grad(cur, t[lo:hi], y[lo:hi])
        scale = N / (hi - lo)
        exp_losses.append(-(ll.sum() * scale) / y.size)
        gl = -(g.sum(0) * scale) / y.size
        th = th - lr * gl
    np.testing.assert_allclose(losses, [np.mean(exp_losses)], rtol=1e-10)
    np.testing.assert_allclose([new.dynamics.drift.sigma, new.dynamics.drift.rho, new.dynamics.drift.beta], th, rtol=1e-10)
    np.testing.assert_allclose(gh[0], gl, rtol=1e-8)
    assert ph[0].dynamics.drift == new.dynamics.drift


@pytest.mark.gpu
def test_fit_sgd_recovers_lorenz63_parameters(hip_lib):
    """Parameter estimation as in the reference's Lorenz-63 tutorials: start from perturbed (sigma, rho, beta), Adam on
    the EKF marginal log-likelihood; the loss falls and every parameter ends closer to the truth."""
    model, params, props = _l63_problem(m=3)
    rng = np.random.default_rng(11)
    true = o.lorenz63_model(3)
    true = o.Model(true.drift, np.eye(3), np.eye(3), np.eye(3), np.zeros(3), np.eye(3), np.zeros(3), 100 * np.eye(3))
    N, T = 64, 150
    t = o.irregular_times(rng, N, T, 0.02)
    y = o.simulate(true, t, rng)
    start = params._replace(dynamics=params.dynamics._replace(drift=cd.LearnableLorenz63(8.0, 25.0, 2.0)))
    new, losses = model.fit_sgd(start, props, y, t[..., None], cd.EKFHyperParams(), optimizer=fit.Adam(0.05), batch_size=N,
                                num_epochs=150)
    assert losses[-1] < losses[0] and np.all(np.isfinite(losses))
    got = np.array([new.dynamics.drift.sigma, new.dynamics.drift.rho, new.dynamics.drift.beta])
    truth, init = np.array([10.0, 28.0, 8 / 3]), np.array([8.0, 25.0, 2.0])
    assert np.all(np.abs(got - truth) < 0.5 * np.abs(init - truth)), got
    ll_true = model.marginal_log_prob(params, y, t[..., None]).sum()
    ll_fit = model.marginal_log_prob(new, y, t[..., None]).sum()
    assert ll_fit > model.marginal_log_prob(start, y, t[..., None]).sum()
    assert ll_fit > ll_true - 0.01 * abs(ll_true)


@pytest.mark.gpu
def test_fit_sgd_lorenz96_forcing_beyond_eight_state_dimensions(hip_lib):
    """fit_sgd on a Lorenz-96 model at d = 20 (ten components observed): the forcing is recovered by Adam on the EKF marginal
    log-likelihood -- value and gradient from the wavefront-per-trajectory forward sweep and the workgroup-per-trajectory reverse
    sweep (ekf_adjoint_wg_kernel); the reference trains models of any size (ssm_temissions.py:492-600).  Then forcing AND emission
    covariance: the loss keeps falling and R moves towards the truth."""
    from cd_dynamax_amd import _ffi
    from helpers import lorenz96_model
    d, m = 20, 10
    true = lorenz96_model(d, m)
    model = cd.ContDiscreteNonlinearGaussianSSM(d, m)
    frozen = PP(trainable=False)

    def problem(forcing, r_scale, r_props):
        return model.initialize(
            key=0,
            initial_mean={"params": true.m0, "props": frozen},
            initial_cov={"params": true.P0, "props": frozen},
            dynamics_drift={"params": cd.LearnableLorenz96(forcing), "props": cd.LearnableLorenz96(PP())},
            dynamics_diffusion_coefficient={"params": cd.LearnableMatrix(np.eye(d)), "props": cd.LearnableMatrix(frozen)},
            dynamics_diffusion_cov={"params": cd.LearnableMatrix(np.eye(d)), "props": cd.LearnableMatrix(frozen)},
            emission_function={"params": cd.LearnableLinear(true.H, np.zeros(m)), "props": cd.LearnableLinear(frozen, frozen)},
            emission_cov={"params": cd.LearnableMatrix(r_scale * np.eye(m)), "props": cd.LearnableMatrix(r_props)},
        )

    rng = np.random.default_rng(96)
    N, T = 16, 60
    t = o.irregular_times(rng, N, T, 0.6)
    y = o.simulate(true, t, rng)
    start, props = problem(6.5, 1.0, frozen)
    new, losses = model.fit_sgd(start, props, y, t[..., None], cd.EKFHyperParams(), optimizer=fit.Adam(0.1), batch_size=N, num_epochs=60)
    assert _ffi.lib().cdkf_last_kernel().decode().startswith(("ekf_adjoint_wave2_l96_kernel<double", "ekf_adjoint_wave_l96_kernel<double", "ekf_adjoint_wg_kernel<double"))
    assert np.all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert abs(float(new.dynamics.drift.forcing) - 8.0) < 0.5 * abs(6.5 - 8.0), new.dynamics.drift.forcing
    # the first step's gradient is the oracle's
    ll, g = o.ekf_loglik_grad_adjoint(o.Model(o.Lorenz96Drift(6.5), true.L, true.Qc, true.H, true.bias, true.R, true.m0, true.P0), t, y)
    _, _, _, gh = model.fit_sgd(start, props, y, t[..., None], cd.EKFHyperParams(), optimizer=fit.SGD(1e-3), batch_size=N, num_epochs=1,
                                return_param_history=True, return_grad_history=True)
    np.testing.assert_allclose(gh[0], -g.sum(0) / y.size, rtol=1e-8)
    from cd_dynamax_amd.bijectors import RealToPSDBijector
    start2, props2 = problem(7.0, 2.0, PP(constrainer=RealToPSDBijector()))
    new2, losses2 = model.fit_sgd(start2, props2, y, t[..., None], cd.EKFHyperParams(), optimizer=fit.Adam(0.05), batch_size=N, num_epochs=40)
    assert np.all(np.isfinite(losses2)) and losses2[-1] < losses2[0]
    r_fit = np.asarray(new2.emissions.emission_cov.params)
    assert np.abs(np.diag(r_fit) - 1.0).mean() < np.abs(2.0 - 1.0) * 0.6, np.diag(r_fit)


@pytest.mark.gpu
def test_fit_sgd_mlp_drift(hip_lib):
    """BASELINE config 5 in miniature: SGD over the marginal log-likelihood of an MLP-drift model (partial observations).
    First step against the oracle's adjoint gradient, then Adam lowers the loss."""
    from helpers import mlp_model
    rng = np.random.default_rng(4)
    d, m = 4, 2
    true = mlp_model(rng, d, m, (16, 16))
    N, T = 12, 30
    t = o.irregular_times(rng, N, T, 0.03)
    y = o.simulate(true, t, rng)
    start = mlp_model(np.random.default_rng(5), d, m, (16, 16))          # different weights, same everything else
    P0 = params_from(start)
    frozen = PP(trainable=False)
    props = P0._replace(
        initial=P0.initial._replace(mean=cd.LearnableVector(frozen), cov=cd.LearnableMatrix(frozen)),
        dynamics=P0.dynamics._replace(drift=cd.LearnableMLP(*([PP()] * 6)), diffusion_coefficient=cd.LearnableMatrix(frozen),
                                      diffusion_cov=cd.LearnableMatrix(frozen), approx_order=frozen),
        emissions=P0.emissions._replace(emission_function=cd.LearnableLinear(frozen, frozen),
                                        emission_cov=cd.LearnableMatrix(frozen)))
    model = cd.ContDiscreteNonlinearGaussianSSM(d, m)
    hyp = cd.EKFHyperParams(state_order="first")
    lr = 0.1
    new, losses = model.fit_sgd(P0, props, y, t[..., None], hyp, optimizer=fit.SGD(lr), batch_size=N, num_epochs=1)
    ll, g = o.ekf_loglik_grad_adjoint(start, t, y)
    np.testing.assert_allclose(losses[0], -ll.sum() / y.size, rtol=1e-10)
    th1 = start.drift.theta() + lr * g.sum(0) / y.size
    got = np.concatenate([np.asarray(a).ravel() for a in new.dynamics.drift])
    np.testing.assert_allclose(got, th1, rtol=1e-8, atol=1e-10)
    new, losses = model.fit_sgd(P0, props, y, t[..., None], hyp, optimizer=fit.Adam(0.01), batch_size=4, num_epochs=25,
                                shuffle=True, key=1)
    assert np.all(np.isfinite(losses)) and losses[-1] < losses[0] and losses[-5:].mean() < losses[:5].mean()
    # the reference's default hyper-parameters (state_order='second': the mean also moves with 0.5 P grad(div f))
    new2, losses2 = model.fit_sgd(P0, props, y, t[..., None], cd.EKFHyperParams(), optimizer=fit.SGD(lr), batch_size=N, num_epochs=1)
    ll2, g2 = o.ekf_loglik_grad_adjoint(start, t, y, state_order="second")
    np.testing.assert_allclose(losses2[0], -ll2.sum() / y.size, rtol=1e-10)
    got2 = np.concatenate([np.asarray(a).ravel() for a in new2.dynamics.drift])
    np.testing.assert_allclose(got2, start.drift.theta() + lr * g2.sum(0) / y.size, rtol=1e-8, atol=1e-10)


@pytest.mark.gpu
def test_fit_sgd_mlp_drift_beyond_eight_state_dimensions(hip_lib):
    """VERDICT r3 "missing" 3: an MLP drift at d = 12 (hidden 32 / 24, five components observed) can be trained -- the workgroup reverse
    sweep with the network's reverse pass (ekf_adjoint_wg_kernel<R, 8, true>), the reference's default state_order='second'.  First SGD
    step = the step computed from the oracle's gradient; Adam then lowers the loss; fp32 gives the same first step to 1e-3."""
    from helpers import mlp_model
    rng = np.random.default_rng(14)
    d, m = 12, 5
    true = mlp_model(rng, d, m, (32, 24))
    N, T = 8, 20
    t = o.irregular_times(rng, N, T, 0.03)
    y = o.simulate(true, t, rng)
    start = mlp_model(np.random.default_rng(15), d, m, (32, 24))
    P0 = params_from(start)
    frozen = PP(trainable=False)
    props = P0._replace(
        initial=P0.initial._replace(mean=cd.LearnableVector(frozen), cov=cd.LearnableMatrix(frozen)),
        dynamics=P0.dynamics._replace(drift=cd.LearnableMLP(*([PP()] * 6)), diffusion_coefficient=cd.LearnableMatrix(frozen),
                                      diffusion_cov=cd.LearnableMatrix(frozen), approx_order=frozen),
        emissions=P0.emissions._replace(emission_function=cd.LearnableLinear(frozen, frozen),
                                        emission_cov=cd.LearnableMatrix(frozen)))
    model = cd.ContDiscreteNonlinearGaussianSSM(d, m)
    lr = 0.1
    new, losses = model.fit_sgd(P0, props, y, t[..., None], cd.EKFHyperParams(), optimizer=fit.SGD(lr), batch_size=N, num_epochs=1)
    ll, g = o.ekf_loglik_grad_adjoint(start, t, y, state_order="second")
    np.testing.assert_allclose(losses[0], -ll.sum() / y.size, rtol=1e-10)
    th1 = start.drift.theta() + lr * g.sum(0) / y.size
    got = np.concatenate([np.asarray(a).ravel() for a in new.dynamics.drift])
    np.testing.assert_allclose(got, th1, rtol=1e-8, atol=1e-10)
    new32, _ = model.fit_sgd(P0, props, y.astype(np.float32), t[..., None], cd.EKFHyperParams(), optimizer=fit.SGD(lr), batch_size=N, num_epochs=1)
    got32 = np.concatenate([np.asarray(a).ravel() for a in new32.dynamics.drift])
    assert np.abs(got32 - th1).max() < 1e-3 * np.abs(th1 - start.drift.theta()).max() + 1e-6
    new, losses = model.fit_sgd(P0, props, y, t[..., None], cd.EKFHyperParams(), optimizer=fit.Adam(0.01), batch_size=4, num_epochs=15,
                                shuffle=True, key=1)
    assert np.all(np.isfinite(losses)) and losses[-1] < losses[0] and losses[-3:].mean() < losses[:3].mean()


@pytest.mark.gpu
def test_fit_sgd_all_parameters_first_step_and_descent(hip_lib):
    """Every leaf trainable -- drift, initial mean, diffusion coefficient, emission weights / bias unconstrained, the three
    covariances through RealToPSDBijector as in the reference's default props -- on a linear model (the cdlgssm
    learn-parameters notebooks' set-up, d = 4, m = 2).  The first plain-SGD step must equal the step computed from the
    oracle's all-parameter adjoint pulled back through the bijectors; Adam then lowers the loss."""
    from cd_dynamax_amd.bijectors import RealToPSDBijector
    from helpers import linear_model
    rng = np.random.default_rng(8)
    d, m = 4, 2
    mdl = linear_model(rng, d, m)
    N, T = 10, 25
    t = o.irregular_times(rng, N, T, 0.05)
    y = o.simulate(mdl, t, rng)
    P0 = params_from(mdl)
    psd, free = PP(constrainer=RealToPSDBijector()), PP()
    props = P0._replace(
        initial=P0.initial._replace(mean=cd.LearnableVector(free), cov=cd.LearnableMatrix(psd)),
        dynamics=P0.dynamics._replace(drift=cd.LearnableLinear(free, free), diffusion_coefficient=cd.LearnableMatrix(free),
                                      diffusion_cov=cd.LearnableMatrix(psd), approx_order=PP(False)),
        emissions=P0.emissions._replace(emission_function=cd.LearnableLinear(free, free), emission_cov=cd.LearnableMatrix(psd)))
    model = cd.ContDiscreteNonlinearGaussianSSM(d, m)
    hyp = cd.EKFHyperParams(state_order="first")
    lr = 0.05
    new, losses = model.fit_sgd(P0, props, y, t[..., None], hyp, optimizer=fit.SGD(lr), batch_size=N, num_epochs=1)
    ll, g, ex = o.ekf_loglik_grad_adjoint(mdl, t, y, full=True)
    np.testing.assert_allclose(losses[0], -ll.sum() / y.size, rtol=1e-10)
    b = RealToPSDBijector()
    sz = y.size
    # unconstrained leaves: theta <- theta + lr * grad / size
    np.testing.assert_allclose(new.dynamics.drift.weights, mdl.drift.W + lr * g.sum(0)[:d * d].reshape(d, d) / sz, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(new.initial.mean.params, mdl.m0 + lr * ex["m0"].sum(0) / sz, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(new.dynamics.diffusion_coefficient.params, mdl.L + lr * ex["L"].sum(0) / sz, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(new.emissions.emission_function.weights, mdl.H + lr * ex["H"].sum(0) / sz, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(new.emissions.emission_function.bias, mdl.bias + lr * ex["bias"].sum(0) / sz, rtol=1e-8, atol=1e-11)
    # PSD leaves: the step is taken on the unconstrained vector
    for got, base, key in ((new.emissions.emission_cov.params, mdl.R, "R"), (new.initial.cov.params, mdl.P0, "P0"),
                           (new.dynamics.diffusion_cov.params, mdl.Qc, "Qc")):
        u = b.inverse(base)
        np.testing.assert_allclose(got, b.forward(u + lr * b.forward_vjp(u, ex[key].sum(0)) / sz), rtol=1e-8, atol=1e-11)
    new, losses = model.fit_sgd(P0, props, y, t[..., None], hyp, optimizer=fit.Adam(0.02), batch_size=5, num_epochs=30)
    assert np.all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert np.linalg.eigvalsh(new.emissions.emission_cov.params).min() > 0
    ll_new = model.marginal_log_prob(new, y, t[..., None], hyp).sum()
    assert ll_new > ll.sum()


@pytest.mark.gpu
def test_linear_model_gradient_against_exact_kalman_filter(hip_lib):
    """BASELINE config 1 (tracking model, d_x = 4, d_y = 2, regular grid: 100 Dormand-Prince steps per interval) through
    ContDiscreteLinearGaussianSSM.marginal_log_prob_and_grad: directional derivatives of the EXACT (matrix-exponential)
    Kalman filter's log-likelihood by central differences -- an implementation that shares nothing with the kernels or
    the oracle -- and then fit_sgd on the linear surface."""
    from cd_dynamax_amd.bijectors import RealToPSDBijector
    from helpers import closed_form_kf
    F = np.zeros((4, 4))
    F[0, 2] = F[1, 3] = 1.0
    H = np.eye(4)[:2]
    m0 = np.array([8.0, 10.0, 1.0, 0.0])
    rng = np.random.default_rng(1)
    T = 60
    t = np.arange(T, dtype=float)

    def mk(F_=F, R_=0.5 * np.eye(2), Qc_=0.1 * np.eye(4), H_=H, m0_=m0):
        return o.Model(o.LinearDrift(F_, np.zeros(4)), np.eye(4), Qc_, H_, np.zeros(2), R_, m0_, np.eye(4))

    mdl = mk()
    y = o.simulate(mdl, t[None], rng)[0]
    model = cd.ContDiscreteLinearGaussianSSM(state_dim=4, emission_dim=2)
    free, psd, frozen = PP(), PP(constrainer=RealToPSDBijector()), PP(False)
    params, props = model.initialize(
        initial_mean={"params": m0, "props": free}, initial_cov={"params": np.eye(4), "props": frozen},
        dynamics_weights={"params": F, "props": free}, dynamics_diffusion_coefficient={"params": np.eye(4), "props": frozen},
        dynamics_diffusion_cov={"params": 0.1 * np.eye(4), "props": psd}, emission_weights={"params": H, "props": free},
        emission_cov={"params": 0.5 * np.eye(2), "props": psd})
    hyp = cd.KFHyperParams(dt_final=1.0)
    ll, g = model.marginal_log_prob_and_grad(params, y, t[:, None], hyp)
    assert abs(ll - closed_form_kf(mdl, t, y, 1.0)["marginal_loglik"]) < 1e-7 * abs(ll)
    h = 1e-5
    for name, leaf, sym in (("F_", g.dynamics.weights, False), ("R_", g.emissions.cov, True), ("Qc_", g.dynamics.diffusion_cov, True),
                            ("H_", g.emissions.weights, False), ("m0_", g.initial.mean, False)):
        base = {"F_": F, "R_": 0.5 * np.eye(2), "Qc_": 0.1 * np.eye(4), "H_": H, "m0_": m0}[name]
        u = rng.standard_normal(base.shape)
        if sym:
            u = 0.5 * (u + u.T)
        fd = (closed_form_kf(mk(**{name: base + h * u}), t, y, 1.0)["marginal_loglik"]
              - closed_form_kf(mk(**{name: base - h * u}), t, y, 1.0)["marginal_loglik"]) / (2 * h)
        assert abs((np.asarray(leaf) * u).sum() - fd) < 1e-5 * abs(fd) + 1e-7, name
    start = params._replace(emissions=params.emissions._replace(cov=1.5 * np.eye(2)),
                            dynamics=params.dynamics._replace(diffusion_cov=0.3 * np.eye(4)))
    new, losses = model.fit_sgd(start, props, y, t[:, None], hyp, optimizer=fit.Adam(0.003), num_epochs=40)
    assert np.all(np.isfinite(losses)) and losses[-1] < losses[0] and losses[-5:].mean() < losses[:5].mean()
    assert isinstance(new, cd.ParamsCDLGSSM) and np.linalg.eigvalsh(new.emissions.cov).min() > 0
    np.testing.assert_array_equal(new.initial.cov, params.initial.cov)        # frozen leaf untouched
    # a dynamics bias is a trainable leaf like any other (round 4: tests/test_gpu_parity.py::test_linear_front_end_trains_...): from zero
    # it moves, the other leaves with it
    withb, lb = model.fit_sgd(params._replace(dynamics=params.dynamics._replace(bias=np.zeros(4))), props, y, t[:, None], hyp, num_epochs=3)
    assert np.all(np.isfinite(lb)) and np.abs(withb.dynamics.bias).max() > 0 and withb.emissions.bias is None
