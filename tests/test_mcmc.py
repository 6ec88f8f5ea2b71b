"""fit_mcmc (reference: ssm_temissions.py:601-777 over blackjax 0.9.6's hmc / nuts + window_adaptation).  The draws cannot be
compared sample by sample (JAX's PRNG is not reproduced), so the sampler is pinned distributionally on targets with known
moments (CPU), its pieces against their published definitions, and the model-facing path on the GPU against densities
recomputed through the plain filter and against the exact linear-Gaussian posterior surface."""
import numpy as np
import pytest

import cd_dynamax_amd as cd
import cdkf_oracle as o
from cd_dynamax_amd import mcmc
from cd_dynamax_amd.bijectors import RealToPSDBijector
from cd_dynamax_amd.params import ParameterProperties as PP


def _gaussian(mean, cov):
    prec = np.linalg.inv(cov)

    def f(q):
        d = q - mean
        return -0.5 * d @ prec @ d, -prec @ d
    return f


def test_adaptation_schedule_matches_stan_windows():
    """blackjax.adaptation.window_adaptation.schedule: 75 fast, slow windows 25 / 50 / 100 / 200 / 500, 50 fast for 1000
    steps; fewer than 20 steps never touch the mass matrix; short runs use the 15 % / 75 % / 10 % split."""
    s = mcmc.adaptation_schedule(1000)
    assert len(s) == 1000 and not any(slow for slow, _ in s[:75]) and not any(slow for slow, _ in s[950:])
    ends = [i for i, (_, e) in enumerate(s) if e]
    assert ends == [99, 149, 249, 449, 949]
    assert all(slow for slow, _ in s[75:950])
    assert mcmc.adaptation_schedule(4) == [(False, False)] * 4 and mcmc.adaptation_schedule(19) == [(False, False)] * 19
    s = mcmc.adaptation_schedule(100)
    assert len(s) == 100 and [i for i, (_, e) in enumerate(s) if e] == [89] and sum(slow for slow, _ in s) == 75
    for n in (20, 57, 150, 333):
        assert len(mcmc.adaptation_schedule(n)) == n


def test_dual_averaging_and_welford():
    da = mcmc._DualAveraging(0.5)
    assert abs(da.mu - np.log(5.0)) < 1e-15 and abs(da.current - 0.5) < 1e-15
    da.update(1.0)   # acceptance above target: the step grows towards 10 x the start
    assert da.current > 0.5
    for _ in range(200):
        da.update(0.0)
    assert da.current < 0.05 and 0 < da.final < 0.5
    w = mcmc._Welford(3)
    x = np.random.default_rng(0).standard_normal((50, 3)) * [1.0, 2.0, 3.0]
    for r in x:
        w.update(r)
    np.testing.assert_allclose(w.mean, x.mean(0), rtol=1e-12)
    np.testing.assert_allclose(w.regularised_variance(), 50 / 55 * x.var(0, ddof=1) + 1e-3 * 5 / 55, rtol=1e-12)


def test_psd_bijector_log_det_jacobian_against_numerical_jacobian():
    """forward_log_det_jacobian of the TFP chain (CholeskyOuterProduct o FillScaleTriL(Exp)) equals log|det| of the Jacobian
    of x -> vech(forward(x)), evaluated here by central differences; its gradient is constant."""
    b = RealToPSDBijector()
    rng = np.random.default_rng(5)
    for n in (1, 2, 3, 4):
        m = n * (n + 1) // 2
        x = rng.standard_normal(m) * 0.5
        r, c = np.tril_indices(n)
        J = np.zeros((m, m))
        h = 1e-6
        for k in range(m):
            e = np.zeros(m)
            e[k] = h
            J[:, k] = (b.forward(x + e)[r, c] - b.forward(x - e)[r, c]) / (2 * h)
        val, grad = b.forward_log_det_jacobian_and_grad(x)
        assert abs(val - np.linalg.slogdet(J)[1]) < 1e-6
        val2, _ = b.forward_log_det_jacobian_and_grad(x + 0.1 * grad)
        assert abs((val2 - val) - 0.1 * grad @ grad) < 1e-12


@pytest.mark.parametrize("algorithm,par", [("nuts", {}), ("hmc", {"num_integration_steps": 12})])
def test_samplers_recover_a_correlated_gaussian(algorithm, par):
    """Warm-up + sampling on N(mean, cov) with scales spanning 1 : 20 and correlation 0.6: posterior mean and covariance to
    Monte-Carlo accuracy, acceptance near the 0.8 target, an adapted mass matrix that tracks the marginal variances."""
    rng = np.random.default_rng(3)
    mean = np.array([1.0, -2.0, 0.5])
    sd = np.array([0.05, 1.0, 0.3])
    corr = np.array([[1, 0.6, 0], [0.6, 1, 0], [0, 0, 1.0]])
    cov = corr * np.outer(sd, sd)
    f = _gaussian(mean, cov)
    state, eps, inv_mass, wpos, wlps, winfo = mcmc.window_adaptation(rng, f, np.zeros(3), algorithm, 400, **par)
    pos, lps, info = mcmc.sample(rng, f, state, algorithm, eps, inv_mass, 3000, **par)
    assert wpos.shape == (400, 3) and pos.shape == (3000, 3) and info["divergences"] == 0
    np.testing.assert_allclose(inv_mass, sd ** 2, rtol=0.6)
    assert 0.6 < info["acceptance"].mean() < 0.97
    assert np.all(np.abs(pos.mean(0) - mean) < 5 * sd / np.sqrt(300)), pos.mean(0)
    assert np.all(np.abs(np.cov(pos.T) - cov) < 0.25 * np.outer(sd, sd)), np.cov(pos.T)
    np.testing.assert_allclose(lps[:5], [f(q)[0] for q in pos[:5]], rtol=1e-12)


def test_nuts_tree_is_bounded_and_reports_divergences():
    rng = np.random.default_rng(0)
    f = _gaussian(np.zeros(2), np.eye(2))
    h = mcmc._Hamiltonian(f, np.ones(2), 1e-4)          # tiny steps: the tree hits the doubling cap, 2^4 - 1 leapfrogs
    q, lp, g, acc, div = mcmc.nuts_step(rng, h, np.ones(2), *f(np.ones(2)), max_num_doublings=4)
    assert h.evals == 15 and not div and acc > 0.99
    h = mcmc._Hamiltonian(f, np.ones(2), 50.0)          # absurd step: energy error above the threshold
    q, lp, g, acc, div = mcmc.nuts_step(rng, h, np.ones(2), *f(np.ones(2)))
    assert div and np.array_equal(q, np.ones(2))
    with pytest.raises(NotImplementedError):
        mcmc._transition(rng, h, q, lp, g, "rmhmc", {})


def _l63(m=3):
    model = cd.ContDiscreteNonlinearGaussianSSM(3, m)
    frozen = PP(trainable=False)
    params, props = model.initialize(
        key=0, initial_mean={"params": np.zeros(3), "props": frozen}, initial_cov={"params": 100 * np.eye(3), "props": frozen},
        dynamics_drift={"params": cd.LearnableLorenz63(10.0, 28.0, 8 / 3), "props": cd.LearnableLorenz63(PP(), PP(), PP())},
        dynamics_diffusion_coefficient={"params": cd.LearnableMatrix(np.eye(3)), "props": cd.LearnableMatrix(frozen)},
        dynamics_diffusion_cov={"params": cd.LearnableMatrix(np.eye(3)), "props": cd.LearnableMatrix(frozen)},
        emission_function={"params": cd.LearnableLinear(np.eye(m, 3), np.zeros(m)), "props": cd.LearnableLinear(frozen, frozen)},
        emission_cov={"params": cd.LearnableMatrix(np.eye(m)), "props": cd.LearnableMatrix(frozen)})
    return model, params, props


def test_fit_mcmc_refusals_need_no_gpu():
    model, params, props = _l63()
    y, t = np.zeros((5, 3)), np.arange(5.0)[:, None]
    with pytest.raises(NotImplementedError, match="rmhmc"):
        model.fit_mcmc(params, props, y, t, mcmc_algorithm={"type": "rmhmc", "parameters": {"num_steps": 2}})
    with pytest.raises(NotImplementedError, match="EKF"):   # (the unscented filter is served from round 4 on; the ensemble filter is stochastic)
        model.fit_mcmc(params, props, y, t, cd.EnKFHyperParams())
    assert model.log_prior(params) == 0.0


@pytest.mark.gpu
def test_fit_mcmc_lorenz63_posterior(hip_lib):
    """The reference's default call shape (NUTS) on the Lorenz-63 drift: every returned log density is the batch marginal
    log-likelihood at that draw (recomputed through the plain filter), non-trainable leaves are broadcast, and the
    posterior concentrates around the data-generating (sigma, rho, beta)."""
    model, params, props = _l63()
    rng = np.random.default_rng(2)
    true = o.Model(o.lorenz63_model(3).drift, np.eye(3), np.eye(3), np.eye(3), np.zeros(3), np.eye(3), np.zeros(3), 100 * np.eye(3))
    N, T = 16, 120
    t = o.irregular_times(rng, N, T, 0.02)
    y = o.simulate(true, t, rng)
    start = params._replace(dynamics=params.dynamics._replace(drift=cd.LearnableLorenz63(9.0, 27.0, 2.4)))
    wp, sp, wlp, slp, info = model.fit_mcmc(start, props, y, t[..., None], n_mcmc_samples=60, verbose=False, key=1, return_info=True,
                                            mcmc_algorithm={"type": "nuts", "parameters": {"num_steps": 60}})
    assert wlp.shape == (60,) and slp.shape == (60,) and np.all(np.isfinite(slp))
    assert sp.dynamics.drift.sigma.shape == (60,) and wp.dynamics.drift.rho.shape == (60,)
    assert sp.initial.cov.params.shape == (60, 3, 3) and np.array_equal(sp.initial.cov.params[7], 100 * np.eye(3))
    for i in (0, 31, 59):
        p_i = start._replace(dynamics=start.dynamics._replace(drift=cd.LearnableLorenz63(
            sp.dynamics.drift.sigma[i], sp.dynamics.drift.rho[i], sp.dynamics.drift.beta[i])))
        assert abs(model.marginal_log_prob(p_i, y, t[..., None]).sum() - slp[i]) < 1e-9 * abs(slp[i])
    draws = np.stack([sp.dynamics.drift.sigma, sp.dynamics.drift.rho, sp.dynamics.drift.beta], 1)
    truth = np.array([10.0, 28.0, 8 / 3])
    assert np.all(np.abs(draws.mean(0) - truth) < 4 * draws.std(0) + 0.05), (draws.mean(0), draws.std(0))
    assert np.all(draws.std(0) < [1.5, 1.5, 1.0]) and info["sampling"]["acceptance"].mean() > 0.4
    assert slp.mean() > model.marginal_log_prob(start, y, t[..., None]).sum()
    # the reference's default mcmc_algorithm (4 warm-up steps) and the HMC variant of the oscillator notebook run as well
    out = model.fit_mcmc(start, props, y, t[..., None], n_mcmc_samples=3, verbose=False)
    assert out[2].shape == (4,) and out[3].shape == (3,)
    out = model.fit_mcmc(start, props, y, t[..., None], n_mcmc_samples=3, verbose=False,
                         mcmc_algorithm={"type": "hmc", "parameters": {"num_steps": 5, "num_integration_steps": 4}})
    assert out[1].dynamics.drift.beta.shape == (3,)


@pytest.mark.gpu
def test_fit_mcmc_linear_model_with_psd_constrainer(hip_lib):
    """ContDiscreteLinearGaussianSSM.fit_mcmc with a RealToPSDBijector-constrained emission covariance (the oscillator
    notebook's set-up in miniature): log densities equal exact Kalman log-likelihood + log|det J| at the draws."""
    from helpers import closed_form_kf
    F = np.array([[0.0, 1.0], [-1.0, -0.2]])
    rng = np.random.default_rng(4)
    T = 80
    t = np.cumsum(rng.uniform(0.05, 0.3, T))
    mk = lambda R_: o.Model(o.LinearDrift(F, np.zeros(2)), np.eye(2), 0.1 * np.eye(2), np.eye(2)[:1], np.zeros(1), R_,
                            np.zeros(2), np.eye(2))
    y = o.simulate(mk(0.2 * np.eye(1)), t[None], rng)[0]
    model = cd.ContDiscreteLinearGaussianSSM(state_dim=2, emission_dim=1)
    frozen = PP(False)
    params, props = model.initialize(
        initial_mean={"params": np.zeros(2), "props": frozen}, initial_cov={"params": np.eye(2), "props": frozen},
        dynamics_weights={"params": F, "props": frozen}, dynamics_diffusion_coefficient={"params": np.eye(2), "props": frozen},
        dynamics_diffusion_cov={"params": 0.1 * np.eye(2), "props": frozen}, emission_weights={"params": np.eye(2)[:1], "props": frozen},
        emission_cov={"params": 0.5 * np.eye(1), "props": PP(constrainer=RealToPSDBijector())})
    hyp = cd.KFHyperParams()
    wp, sp, wlp, slp = model.fit_mcmc(params, props, y, t[:, None], hyp, n_mcmc_samples=50, verbose=False, key=3,
                                      mcmc_algorithm={"type": "nuts", "parameters": {"num_steps": 50}})
    assert isinstance(sp, cd.ParamsCDLGSSM) and sp.emissions.cov.shape == (50, 1, 1) and sp.dynamics.weights.shape == (50, 2, 2)
    b = RealToPSDBijector()
    for i in (0, 25, 49):
        R_i = sp.emissions.cov[i]
        exact = closed_form_kf(mk(R_i), t, y, hyp.dt_final)["marginal_loglik"] + b.forward_log_det_jacobian_and_grad(b.inverse(R_i))[0]
        assert abs(exact - slp[i]) < 1e-6 * abs(exact) + 1e-6
    r = sp.emissions.cov[:, 0, 0]
    assert 0.05 < np.median(r) < 0.6 and r.min() > 0
