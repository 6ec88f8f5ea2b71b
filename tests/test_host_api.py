"""Host-side logic of the drop-in surface (cd_dynamax_amd/models.py), exercised on CPU by replacing the
C-ABI call with the oracle: argument handling, batching, layout transposes, output_fields, dispatch and
refusals behave like the reference's cdnlgssm_filter / cdnlgssm_smoother (models.py:658-764)."""
import numpy as np
import pytest

import cd_dynamax_amd as cd
import cdkf_oracle as o
from cd_dynamax_amd import _ffi, models
from helpers import linear_model, params_from

ORDER_NAMES = {0: "zeroth", 1: "first", 2: "second"}


def _mdl_from_block(blk):
    d = blk.state_dim
    th = blk.theta
    drift = {0: lambda: o.LinearDrift(th[: d * d].reshape(d, d), th[d * d:]), 1: lambda: o.Lorenz63Drift(*th),
             2: lambda: o.Lorenz96Drift(th[0])}[blk.c.drift_kind]()
    return o.Model(drift, blk.L, blk.Qc, blk.H, blk.h_bias, blk.R, blk.m0, blk.P0)


@pytest.fixture
def oracle_backend(monkeypatch):
    """Replaces _ffi.run_host (the only place models.py touches the library) with the oracle."""
    calls = []

    def fake(algo, blk, opts, t, y, want, dtype):
        calls.append((algo, opts.t_shared, opts.dt_final, opts.num_iter, tuple(want), np.dtype(dtype)))
        mdl = _mdl_from_block(blk)
        N, T, _ = y.shape
        tt = np.broadcast_to(t, (N, T)) if opts.t_shared else t
        kw = dict(dt0=opts.dt0, dt_final=opts.dt_final, max_steps=opts.max_steps, dtype=dtype)
        if algo == "ekf_filter":
            r = o.ekf_filter(mdl, tt, y, state_order=ORDER_NAMES[opts.state_order], num_iter=opts.num_iter, **kw)
            keys = ["filtered_means", "filtered_covariances", "predicted_means", "predicted_covariances"]
        elif algo == "ukf_filter":
            r = o.ukf_filter(mdl, tt, y, alpha=opts.ukf_alpha, beta=opts.ukf_beta, kappa=opts.ukf_kappa, **kw)
            keys = ["filtered_means", "filtered_covariances", "predicted_means", "predicted_covariances"]
        else:
            r = o.ekf_smoother(mdl, tt, y, state_order=ORDER_NAMES[opts.state_order], **kw)
            keys = ["filtered_means", "filtered_covariances", "smoothed_means", "smoothed_covariances"]
        return r["marginal_loglik"], [r[k] if w else None for k, w in zip(keys, want)], np.zeros(N, np.int32)

    def fake_smoother1(blk, opts, t, y, dtype):
        calls.append(("kf_smoother1", opts.t_shared, opts.dt_final, opts.num_iter, (), np.dtype(dtype)))
        mdl = _mdl_from_block(blk)
        N, T, _ = y.shape
        tt = np.broadcast_to(t, (N, T)) if opts.t_shared else t
        r = o.kf_smoother_type1(mdl, tt, y, dt0=opts.dt0, dt_final=opts.dt_final, max_steps=opts.max_steps, dtype=dtype)
        return (r["marginal_loglik"], r["filtered_means"], r["filtered_covariances"], r["smoothed_means"],
                r["smoothed_covariances"], r["smoothed_cross_covariances"], np.zeros(N, np.int32))

    def fake_pushforward(blk, opts, t, dtype):
        calls.append(("kf_pushforward", 0, opts.dt_final, opts.num_iter, (), np.dtype(dtype)))
        mdl = _mdl_from_block(blk)
        N, T = t.shape
        A = np.zeros((N, T - 1, mdl.d, mdl.d))
        Q = np.zeros_like(A)
        for k in range(T - 1):
            A[:, k], Q[:, k] = o.kf_pushforward(mdl, t[:, k], t[:, k + 1], opts.dt0, opts.max_steps)
        return A, Q

    monkeypatch.setattr(_ffi, "run_host", fake)
    monkeypatch.setattr(_ffi, "kf_smoother1", fake_smoother1)
    monkeypatch.setattr(_ffi, "kf_pushforward", fake_pushforward)
    monkeypatch.setattr(_ffi, "default_opts", lambda: _default())
    return calls


def _default():
    o_ = _ffi.CdkfOpts()
    o_.state_order, o_.num_iter, o_.device, o_.max_steps = 2, 1, -1, 100000
    o_.dt0, o_.dt_final, o_.cov_rescaling = 0.01, 1e-10, 1.0
    o_.ukf_alpha, o_.ukf_beta, o_.ukf_kappa = np.sqrt(3), 2.0, 1.0
    return o_


def test_single_trajectory_shapes_and_fields(oracle_backend):
    rng = np.random.default_rng(0)
    mdl = o.lorenz63_model(1)
    T = 12
    t = o.irregular_times(rng, 1, T, 0.1)
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    post = cd.cdnlgssm_filter(P, y[0], t[0][:, None])
    assert isinstance(post, cd.PosteriorGSSMFiltered)
    assert np.ndim(post.marginal_loglik) == 0
    assert post.filtered_means.shape == (T, 3) and post.predicted_covariances.shape == (T, 3, 3)
    ref = o.ekf_filter(mdl, t, y)
    np.testing.assert_allclose(post.filtered_means, ref["filtered_means"][0])
    post = cd.cdnlgssm_filter(P, y[0], t[0][:, None], output_fields=["filtered_means"])
    assert post.filtered_covariances is None and post.predicted_means is None and post.filtered_means is not None
    assert oracle_backend[-1][4] == (True, False, False, False)
    # 1-D emissions (emission_dim = 1) are accepted
    post = cd.cdnlgssm_filter(P, y[0, :, 0], t[0][:, None], output_fields=[])
    np.testing.assert_allclose(post.marginal_loglik, ref["marginal_loglik"][0])


def test_batched_and_shared_time_grids(oracle_backend):
    rng = np.random.default_rng(1)
    mdl = o.lorenz63_model(3)
    N, T = 3, 10
    t = o.irregular_times(rng, N, T, 0.1)
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    post = cd.cdnlgssm_filter(P, y, t[..., None])
    assert post.marginal_loglik.shape == (N,) and post.filtered_covariances.shape == (N, T, 3, 3)
    assert oracle_backend[-1][1] == 0
    post2 = cd.cdnlgssm_filter(P, y, t[0][:, None])  # one grid shared by the batch
    assert oracle_backend[-1][1] == 1
    ref = o.ekf_filter(mdl, np.broadcast_to(t[0], (N, T)), y)
    np.testing.assert_allclose(post2.filtered_means, ref["filtered_means"])


def test_t_emissions_none_is_arange_with_unit_last_interval(oracle_backend):
    rng = np.random.default_rng(2)
    mdl = linear_model(rng, 2, 2)
    T = 9
    tt = np.arange(T, dtype=float)[None]
    y = o.simulate(mdl, tt, rng)
    P = params_from(mdl)
    a = cd.cdnlgssm_filter(P, y[0])
    assert oracle_backend[-1][1:3] == (1, 1.0)
    b = cd.cdnlgssm_filter(P, y[0], tt[0][:, None], cd.EKFHyperParams(dt_final=1.0))
    np.testing.assert_allclose(a.predicted_covariances, b.predicted_covariances)
    np.testing.assert_allclose(a.marginal_loglik, b.marginal_loglik)


def test_dispatch_by_hyperparams_class(oracle_backend):
    rng = np.random.default_rng(3)
    mdl = o.lorenz63_model(3)
    t = o.irregular_times(rng, 1, 6, 0.05)
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    cd.cdnlgssm_filter(P, y[0], t[0][:, None], cd.UKFHyperParams())
    assert oracle_backend[-1][0] == "ukf_filter"
    cd.cdnlgssm_filter(P, y[0], t[0][:, None], cd.EKFHyperParams(state_order="first"), num_iter=3)
    assert oracle_backend[-1][0] == "ekf_filter" and oracle_backend[-1][3] == 3
    sm = cd.cdnlgssm_smoother(P, y[0], t[0][:, None], cd.EKFHyperParams(), num_iter=5)
    assert isinstance(sm, cd.PosteriorGSSMSmoothed) and oracle_backend[-1][0] == "ekf_smoother"
    assert oracle_backend[-1][3] == 1  # the smoother's filter runs with num_iter = 1 (inference_ekf.py:489-495)
    assert sm.smoothed_cross_covariances is None
    with pytest.raises(ValueError, match="UKS not implemented yet"):
        cd.cdnlgssm_smoother(P, y[0], t[0][:, None], cd.UKFHyperParams())
    with pytest.raises(ValueError, match="EnKS not implemented yet"):
        cd.cdnlgssm_smoother(P, y[0], t[0][:, None], cd.EnKFHyperParams())
    with pytest.raises(NotImplementedError):
        cd.cdnlgssm_filter(P, y[0], t[0][:, None], cd.EnKFHyperParams())
    with pytest.raises(ValueError, match="state_order"):
        cd.cdnlgssm_filter(P, y[0], t[0][:, None], cd.EKFHyperParams(state_order="third"))
    with pytest.raises(NotImplementedError, match="diffeqsolve_settings"):
        cd.cdnlgssm_filter(P, y[0], t[0][:, None], cd.EKFHyperParams(diffeqsolve_settings={"stepsize_controller": "PIDController"}))
    with pytest.raises(NotImplementedError, match="choose from"):
        cd.cdnlgssm_filter(P, y[0], t[0][:, None], cd.EKFHyperParams(diffeqsolve_settings={"solver": "Kvaerno5"}))


def test_dtype_follows_emissions(oracle_backend):
    rng = np.random.default_rng(4)
    mdl = o.lorenz63_model(3)
    t = o.irregular_times(rng, 1, 6, 0.05)
    y = o.simulate(mdl, t, rng)
    P = params_from(mdl)
    post = cd.cdnlgssm_filter(P, y[0].astype(np.float32), t[0][:, None])
    assert oracle_backend[-1][5] == np.float32 and post.filtered_means.dtype == np.float32
    post = cd.cdnlgssm_filter(P, y[0], t[0][:, None])
    assert oracle_backend[-1][5] == np.float64
    post = cd.cdnlgssm_filter(P, y[0], t[0][:, None], dtype=np.float32)
    assert post.filtered_means.dtype == np.float32


def test_unknown_drift_or_emission_is_refused_not_approximated():
    class MyDrift:
        def f(self, x, u=None, t=None):
            return -x

    mdl = o.lorenz63_model(3)
    P = params_from(mdl)
    bad = P._replace(dynamics=P.dynamics._replace(drift=MyDrift()))
    with pytest.raises(NotImplementedError, match="drift registry"):
        cd.cdnlgssm_filter(bad, np.zeros((4, 3)))
    bad = P._replace(emissions=P.emissions._replace(emission_function=MyDrift()))
    with pytest.raises(NotImplementedError, match="emission_function"):
        cd.cdnlgssm_filter(bad, np.zeros((4, 3)))


def test_model_class_surface(oracle_backend):
    model = cd.ContDiscreteNonlinearGaussianSSM(state_dim=3, emission_dim=3)
    params, props = model.initialize(
        dynamics_drift={"params": cd.LearnableLorenz63(10.0, 28.0, 8 / 3),
                        "props": cd.LearnableLorenz63(*([cd.ParameterProperties()] * 3))},
        dynamics_diffusion_coefficient={"params": cd.LearnableMatrix(np.eye(3)),
                                        "props": cd.LearnableMatrix(cd.ParameterProperties())},
        emission_function={"params": cd.LearnableLinear(np.eye(3), np.zeros(3)),
                           "props": cd.LearnableLinear(cd.ParameterProperties(), cd.ParameterProperties())})
    assert isinstance(params, cd.ParamsCDNLGSSM) and params.initial.mean.f().shape == (3,)
    assert model.emission_shape == (3,) and model.inputs_shape is None
    rng = np.random.default_rng(5)
    t = o.irregular_times(rng, 2, 7, 0.05)
    y = rng.standard_normal((2, 7, 3))
    ll = model.marginal_log_prob(params, y, t[..., None])
    assert ll.shape == (2,) and oracle_backend[-1][4] == (False,) * 4
    f = model.filter(params, y[0], t[0][:, None])
    s = model.smoother(params, y[0], t[0][:, None])
    np.testing.assert_allclose(f.filtered_means, s.filtered_means)
    np.testing.assert_allclose(ll[0], f.marginal_loglik)
    # defaults of initialize(): linear drift -0.1 I, L = Qc = 0.1 I (models.py:186-243)
    p0, _ = cd.ContDiscreteNonlinearGaussianSSM(2, 4).initialize()
    np.testing.assert_array_equal(p0.dynamics.drift.weights, -0.1 * np.eye(2))
    assert p0.emissions.emission_function.weights.shape == (4, 2)


def test_run_host_transposes_to_time_major(monkeypatch):
    """_ffi.run_host hands the library time-major buffers and returns reference-shaped views."""
    seen = {}

    class FakeLib:
        def cdkf_last_error(self):
            return b""

        def cdkf_preferred_layout(self, mdl):
            return _ffi.LAYOUT_TCN

        def __getattr__(self, name):
            def fn(mdl, opts, N, T, t, y, ll, a1, a2, a3, a4, st):
                seen["layout"], seen["N"], seen["T"] = opts._obj.layout, N, T
                return 0
            return fn

    monkeypatch.setattr(_ffi, "lib", lambda: FakeLib())
    blk = _ffi.ModelBlock(_ffi.DRIFT_LORENZ63, [10, 28, 8 / 3], np.eye(3), np.eye(3), np.eye(3), np.zeros(3), np.eye(3),
                          np.zeros(3), np.eye(3))
    opts = _default()
    ll, outs, st = _ffi.run_host("ekf_filter", blk, opts, np.zeros((5, 7)), np.zeros((5, 7, 3)), [True, True, False, False],
                                 np.float64)
    assert seen == {"layout": _ffi.LAYOUT_TCN, "N": 5, "T": 7}
    assert outs[0].shape == (5, 7, 3) and outs[1].shape == (5, 7, 3, 3) and outs[2] is None
    assert outs[0].base is not None and outs[0].base.shape == (7, 3, 5)  # a view of the native [T,d,N] buffer


def test_linear_model_front_end(oracle_backend):
    """BASELINE config 1 surface: ContDiscreteLinearGaussianSSM.filter / smoother / marginal_log_prob map onto the EKF
    with a LearnableLinear drift (state_order='first'); unsupported reference features are refused, not approximated."""
    from helpers import closed_form_kf
    model = cd.ContDiscreteLinearGaussianSSM(state_dim=4, emission_dim=2)
    F = np.zeros((4, 4))
    F[0, 2] = F[1, 3] = 1.0
    params, props = model.initialize(
        initial_mean={"params": np.array([8.0, 10.0, 1.0, 0.0]), "props": cd.ParameterProperties()},
        dynamics_weights={"params": F, "props": cd.ParameterProperties()},
        dynamics_diffusion_coefficient={"params": np.eye(4), "props": cd.ParameterProperties()},
        emission_weights={"params": np.eye(4)[:2], "props": cd.ParameterProperties()},
        emission_cov={"params": 0.5 * np.eye(2), "props": cd.ParameterProperties()})
    assert isinstance(params, cd.ParamsCDLGSSM) and params.dynamics.bias is None
    rng = np.random.default_rng(0)
    T = 30
    y = rng.standard_normal((T, 2))
    post = model.filter(params, y, filter_hyperparams=cd.KFHyperParams(dt_final=1.0))
    assert oracle_backend[-1][0] == "ekf_filter" and oracle_backend[-1][1:3] == (1, 1.0)
    mdl = o.Model(o.LinearDrift(F, np.zeros(4)), np.eye(4), 0.1 * np.eye(4), np.eye(4)[:2], np.zeros(2), 0.5 * np.eye(2),
                  np.array([8.0, 10.0, 1.0, 0.0]), np.eye(4))
    ref = closed_form_kf(mdl, np.arange(T, dtype=float), y, dt_final=1.0)
    np.testing.assert_allclose(post.filtered_means, ref["filtered_means"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(model.marginal_log_prob(params, y, filter_hyperparams=cd.KFHyperParams(dt_final=1.0)),
                               ref["marginal_loglik"], rtol=1e-8)
    sm = model.smoother(params, y, smoother_type="cd_smoother_2")
    assert oracle_backend[-1][0] == "ekf_smoother" and sm.smoothed_means.shape == (T, 4)
    sm1 = model.smoother(params, y)  # the reference's default: smoother type 1 (discrete RTS on the pushed-forward (A, Q))
    assert oracle_backend[-1][0] == "kf_smoother1" and sm1.smoothed_cross_covariances.shape == (T - 1, 4, 4)
    assert sm1.smoothed_means.shape == (T, 4) and np.ndim(sm1.marginal_loglik) == 0
    # a dynamics bias and inputs (un-integrated in the reference): offsets from the pushed-forward matrices, host logic of linear.py
    b, B, D, u = 0.1 * np.ones(4), rng.standard_normal((4, 1)), rng.standard_normal((2, 1)), rng.standard_normal((T, 1))
    pb = params._replace(dynamics=params.dynamics._replace(bias=b, input_weights=B),
                         emissions=params.emissions._replace(input_weights=D))
    pin = cd.cdlgssm_filter(pb, y, inputs=u, filter_hyperparams=cd.KFHyperParams(dt_final=1.0))
    assert any(c[0] == "kf_pushforward" for c in oracle_backend)
    rin = o.kf_filter_inputs(mdl, np.arange(T, dtype=float)[None], y[None], b, B, D, u[None], dt_final=1.0)
    for k in ("filtered_means", "predicted_means", "filtered_covariances"):
        np.testing.assert_allclose(getattr(pin, k), rin[k][0], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(pin.marginal_loglik, rin["marginal_loglik"][0], rtol=1e-9)
    with pytest.raises(NotImplementedError, match="smoothers take no inputs"):
        model.smoother(pb, y, inputs=u)
