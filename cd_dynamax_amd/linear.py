"""Front-end for the reference's continuous-discrete LINEAR Gaussian SSM (BASELINE.json config 1) on top of the HIP
EKF kernels: for a linear drift the extended Kalman filter IS the Kalman filter, which is exactly what the
reference's own tests assert (src/test_scripts/cdnlgssm_test_filter_linear_TRegular.py:314-324: EKF first / second
order == cdlgssm_filter within rtol 1e-5).

Mirrors /root/reference/src/continuous_discrete_linear_gaussian_ssm/:
  ``KFHyperParams`` inference.py:34-38, ``ParamsCDLGSSMDynamics`` / ``ParamsCDLGSSM`` inference.py:56-103,
  ``cdlgssm_filter`` inference.py:555-632, ``cdlgssm_smoother`` inference.py:694-800,
  ``ContDiscreteLinearGaussianSSM.{initialize, filter, smoother, marginal_log_prob}`` models.py:42-365.

Differences (what cannot be reproduced is refused loudly rather than approximated):
  * dynamics bias / inputs: the reference adds ``B u + b`` to the pushed-forward mean WITHOUT integrating it (inference.py
    ``_predict``: mu = F m + B u + b) and ``D u + d`` to the emission mean.  The filter (and ``marginal_log_prob``) reproduce
    that through offsets: with s_0 = 0, s_k+1 = A_k s_k + B u_k + b (A_k = the pushed-forward matrix of interval k, from
    cdkf_kf_pushforward_*; state_dim <= 8) the state is x = z + s where z runs the bias-free filter on y - H s - D u, so the
    covariances and the log-likelihood are untouched and s is added back to the means.  The smoothers take no inputs (the
    reference's own do not either: inference.py:635 "TODO: incorporate inputs!").  Their gradient -- and fit_sgd over them -- comes from
    the reverse sweep with jumps of the predicted mean (``_loglik_and_grad_with_offsets``, cdkf_ekf_loglik_grad_jumps_*).
  * the moments are integrated directly (dP/dt = F P + P F^T + L Qc L^T) instead of pushing (A, Q) forward and
    forming A P A^T + Q: identical up to the O(dt0^6) difference of two 5th-order solutions (~1e-12 relative).
  * ``smoother_type='cd_smoother_2'`` (Sarkka Alg. 3.18, inference.py:636-690) is the EKF smoother of the hot path;
    the reference's default ``'cd_smoother_1'`` (discrete RTS on the pushed-forward (A, Q)) has its own two kernels
    (cdkf_rts1_kernels.h) for state_dim <= 8; its forward pass is the moment-integrating filter above.
"""
from __future__ import annotations

from typing import Any, NamedTuple, Optional

import numpy as np

from .models import ContDiscreteNonlinearGaussianSSM, cdnlgssm_filter, cdnlgssm_loglik_and_grad_all, cdnlgssm_smoother
from .params import (EKFHyperParams, LearnableLinear, LearnableMatrix, LearnableVector, ParameterProperties,
                     ParamsCDNLGSSM, ParamsCDNLGSSMDynamics, ParamsCDNLGSSMEmissions, ParamsLGSSMInitial,
                     PosteriorGSSMFiltered, PosteriorGSSMSmoothed)


class KFHyperParams(NamedTuple):
    dt_final: float = 1e-10
    diffeqsolve_settings: dict = {}


class ParamsCDLGSSMDynamics(NamedTuple):
    weights: Any
    bias: Any
    input_weights: Any
    diffusion_coefficient: Any
    diffusion_cov: Any


class ParamsLGSSMEmissions(NamedTuple):
    weights: Any
    bias: Any
    input_weights: Any
    cov: Any


class ParamsCDLGSSM(NamedTuple):
    initial: ParamsLGSSMInitial
    dynamics: ParamsCDLGSSMDynamics
    emissions: ParamsLGSSMEmissions


def _has_offsets(params: ParamsCDLGSSM, inputs) -> bool:
    b = params.dynamics.bias
    return (inputs is not None and np.asarray(inputs).size > 0) or (b is not None and bool(np.any(np.asarray(b) != 0)))


def _as_nonlinear(params: ParamsCDLGSSM, inputs, offsets_handled: bool = False) -> ParamsCDNLGSSM:
    if not offsets_handled and _has_offsets(params, inputs):
        raise NotImplementedError(
            "dynamics bias / inputs: the reference adds B u + b to the pushed-forward mean without integrating it "
            "(continuous_discrete_linear_gaussian_ssm/inference.py, _predict); the HIP path reproduces that in cdlgssm_filter / "
            "marginal_log_prob only (the reference's smoothers take no inputs either)")
    F = np.asarray(params.dynamics.weights, dtype=np.float64)
    d = F.shape[0]
    H = np.asarray(params.emissions.weights, dtype=np.float64)
    hb = params.emissions.bias
    hb = np.zeros(H.shape[0]) if hb is None else np.asarray(hb, dtype=np.float64)
    return ParamsCDNLGSSM(
        initial=ParamsLGSSMInitial(LearnableVector(np.asarray(params.initial.mean)), LearnableMatrix(np.asarray(params.initial.cov))),
        dynamics=ParamsCDNLGSSMDynamics(LearnableLinear(F, np.zeros(d)), LearnableMatrix(np.asarray(params.dynamics.diffusion_coefficient)),
                                        LearnableMatrix(np.asarray(params.dynamics.diffusion_cov)), 1.0),
        emissions=ParamsCDNLGSSMEmissions(LearnableLinear(H, hb), LearnableMatrix(np.asarray(params.emissions.cov))))


def _from_nonlinear(nl: ParamsCDNLGSSM, like: ParamsCDLGSSM) -> ParamsCDLGSSM:
    """Inverse of ``_as_nonlinear`` (values or gradients): fields the HIP path does not touch keep ``like``'s entries."""
    return ParamsCDLGSSM(
        initial=ParamsLGSSMInitial(mean=nl.initial.mean.params, cov=nl.initial.cov.params),
        dynamics=ParamsCDLGSSMDynamics(weights=nl.dynamics.drift.weights, bias=like.dynamics.bias,
                                       input_weights=like.dynamics.input_weights,
                                       diffusion_coefficient=nl.dynamics.diffusion_coefficient.params,
                                       diffusion_cov=nl.dynamics.diffusion_cov.params),
        emissions=ParamsLGSSMEmissions(weights=nl.emissions.emission_function.weights,
                                       bias=nl.emissions.emission_function.bias if like.emissions.bias is not None else None,
                                       input_weights=like.emissions.input_weights,
                                       cov=nl.emissions.emission_cov.params))


def _props_as_nonlinear(props: ParamsCDLGSSM, params: ParamsCDLGSSM) -> ParamsCDNLGSSM:
    frozen = ParameterProperties(trainable=False)
    # a leaf that is absent (None) or empty (input weights with input_dim = 0) has nothing to train
    is_on = lambda p, v: isinstance(p, ParameterProperties) and p.trainable and v is not None and np.size(v) > 0
    if (is_on(props.dynamics.bias, params.dynamics.bias) or is_on(props.dynamics.input_weights, params.dynamics.input_weights)
            or is_on(props.emissions.input_weights, params.emissions.input_weights)):
        raise NotImplementedError(
            "the dynamics bias and the input weights are trained through ContDiscreteLinearGaussianSSM.fit_sgd "
            "(_fit_sgd_with_offsets: state_dim, emission_dim <= 8, the default solver); this route has no place for them")
    pp = lambda p: p if isinstance(p, ParameterProperties) else frozen
    return ParamsCDNLGSSM(
        initial=ParamsLGSSMInitial(LearnableVector(pp(props.initial.mean)), LearnableMatrix(pp(props.initial.cov))),
        dynamics=ParamsCDNLGSSMDynamics(LearnableLinear(pp(props.dynamics.weights), frozen),
                                        LearnableMatrix(pp(props.dynamics.diffusion_coefficient)),
                                        LearnableMatrix(pp(props.dynamics.diffusion_cov)), frozen),
        emissions=ParamsCDNLGSSMEmissions(
            LearnableLinear(pp(props.emissions.weights), pp(props.emissions.bias) if params.emissions.bias is not None else frozen),
            LearnableMatrix(pp(props.emissions.cov))))


def _hyper(filter_hyperparams: Optional[KFHyperParams]) -> EKFHyperParams:
    hp = filter_hyperparams if filter_hyperparams is not None else KFHyperParams()
    return EKFHyperParams(dt_final=hp.dt_final, state_order="first", diffeqsolve_settings=hp.diffeqsolve_settings)


def _filter_with_offsets(params: ParamsCDLGSSM, emissions, t_emissions, filter_hyperparams, inputs, dtype) -> PosteriorGSSMFiltered:
    """cdlgssm_filter with a dynamics bias and / or inputs (inference.py:596-620: u_k enters the emission at k and the predict from
    k to k+1, both un-integrated): the bias-free filter in the offset coordinates described in the module docstring."""
    from . import _ffi
    from .models import _model_block, _opts, _prepare
    nl = _as_nonlinear(params, inputs, offsets_handled=True)
    mdl = _model_block(nl)
    hyper = _hyper(filter_hyperparams)
    opts = _opts(hyper, 1)
    y, t, batched, dtype = _prepare(emissions, t_emissions, hyper, opts, dtype)
    if not _ffi.lib().cdkf_kf_smoother1_supported(_ffi.C.byref(mdl.c)):
        raise NotImplementedError(f"a dynamics bias / inputs run on the HIP path for state_dim <= 8 (got {mdl.state_dim})")
    N, T, m = y.shape
    d = mdl.state_dim
    tt = np.broadcast_to(t, (N, T)) if t.ndim == 1 else t
    # one more interval: the last predict runs over dt_final (1 on the regular grid of t_emissions=None)
    t_ext = np.concatenate([tt, tt[:, -1:] + opts.dt_final], axis=1).astype(np.float64)
    A, _ = _ffi.kf_pushforward(mdl, opts, t_ext, np.float64)                       # [N, T, d, d]
    H = np.asarray(params.emissions.weights, dtype=np.float64)
    b = params.dynamics.bias
    c = np.zeros((N, T, d)) + (0.0 if b is None else np.asarray(b, dtype=np.float64))
    yy = y.astype(np.float64)
    if inputs is not None and np.asarray(inputs).size:
        u = np.asarray(inputs, dtype=np.float64)
        u = np.broadcast_to(u if u.ndim == 3 else u[None], (N, T, u.shape[-1]))
        B, Dm = params.dynamics.input_weights, params.emissions.input_weights
        if B is not None and np.size(B):
            c = c + u @ np.asarray(B, dtype=np.float64).T
        if Dm is not None and np.size(Dm):
            yy = yy - u @ np.asarray(Dm, dtype=np.float64).T
    s = np.zeros((N, T + 1, d))
    for k in range(T):
        s[:, k + 1] = np.einsum("nij,nj->ni", A[:, k], s[:, k]) + c[:, k]
    yy = yy - s[:, :T] @ H.T
    post = cdnlgssm_filter(nl, yy.astype(dtype) if batched else yy[0].astype(dtype), t_emissions, hyper, dtype=dtype)
    sh = lambda a, off: None if a is None else (np.asarray(a) + (off if batched else off[0]).astype(np.asarray(a).dtype))
    return post._replace(filtered_means=sh(post.filtered_means, s[:, :T]), predicted_means=sh(post.predicted_means, s[:, 1:]))


def _trains_offsets(props: ParamsCDLGSSM, params: ParamsCDLGSSM) -> bool:
    is_on = lambda p, v: isinstance(p, ParameterProperties) and p.trainable and v is not None and np.size(v) > 0
    return (is_on(props.dynamics.bias, params.dynamics.bias) or is_on(props.dynamics.input_weights, params.dynamics.input_weights)
            or is_on(props.emissions.input_weights, params.emissions.input_weights))


def _loglik_and_grad_with_offsets(params: ParamsCDLGSSM, emissions, t_emissions, filter_hyperparams, inputs, dtype):
    """(marginal log-likelihood, gradient tree) of the linear model WITH a dynamics bias / inputs -- every leaf, the bias and the three
    input-weight matrices included (the reference: jax.value_and_grad of marginal_log_prob, in which they are ordinary trainable leaves,
    continuous_discrete_linear_gaussian_ssm/models.py:116-139, 167).  The reference's predict adds B u_k + b to the pushed-forward mean
    un-integrated (inference.py:185-205): on the device that is a JUMP of the predicted mean behind every interval
    (cdkf_ekf_loglik_grad_jumps_*), whose reverse sweep hands back the cotangent of every jump and of every observation; the chain to
    b, B (jump_k = B u_k + b) and D (the filter sees y_k - D u_k) is three small contractions here."""
    from . import _ffi
    from .models import _grads_tree, _model_block, _opts, _prepare
    nl = _as_nonlinear(params, inputs, offsets_handled=True)
    mdl = _model_block(nl)
    hyper = _hyper(filter_hyperparams)
    opts = _opts(hyper, 1)
    y, t, batched, dtype = _prepare(emissions, t_emissions, hyper, opts, dtype)
    N, T, m = y.shape
    d = mdl.state_dim
    b = params.dynamics.bias
    jumps = np.zeros((N, T, d)) + (0.0 if b is None else np.asarray(b, dtype=np.float64))
    yy = y.astype(np.float64)
    u = None
    B, Dm = params.dynamics.input_weights, params.emissions.input_weights
    if inputs is not None and np.asarray(inputs).size:
        u = np.asarray(inputs, dtype=np.float64)
        u = np.broadcast_to(u if u.ndim == 3 else u[None], (N, T, u.shape[-1]))
        if B is not None and np.size(B):
            jumps = jumps + u @ np.asarray(B, dtype=np.float64).T
        if Dm is not None and np.size(Dm):
            yy = yy - u @ np.asarray(Dm, dtype=np.float64).T
    try:
        ll, gth, gm, gj, gy, _ = _ffi.loglik_grad_jumps(mdl, opts, t, yy, jumps, dtype)
    except _ffi.CdkfUnsupported as e:
        raise NotImplementedError(f"gradient of the linear model with a dynamics bias / inputs: {e}") from e
    g = _from_nonlinear(_grads_tree(nl, mdl, gth, gm), params)
    gj, gy = gj.astype(np.float64), gy.astype(np.float64)
    g_b = None if b is None else gj.sum(axis=1)
    g_B = None if B is None else (np.einsum("nkd,nku->ndu", gj, u) if (u is not None and np.size(B)) else np.zeros((N,) + np.shape(B)))
    g_D = None if Dm is None else (-np.einsum("nkm,nku->nmu", gy, u) if (u is not None and np.size(Dm)) else np.zeros((N,) + np.shape(Dm)))
    pick = (lambda a: a) if batched else (lambda a: None if a is None else a[0])
    g = g._replace(dynamics=g.dynamics._replace(bias=pick(g_b), input_weights=pick(g_B)),
                   emissions=g.emissions._replace(input_weights=pick(g_D)))
    if not batched:
        ll = ll[0]
    return ll, g


def _fit_sgd_with_offsets(model, params, props, emissions, t_emissions, filter_hyperparams, inputs, optimizer, batch_size, num_epochs,
                          shuffle, return_param_history, return_grad_history, key, dtype):
    """``SSM.fit_sgd`` (ssm_temissions.py:492-600) for a linear model with a dynamics bias / inputs, any leaf trainable: the loop of
    ``cd_dynamax_amd.fit.fit_sgd`` (same loss scaling, unconstrained space, optimiser) around ``_loglik_and_grad_with_offsets``; the
    sweeps take host arrays per step (this front-end is BASELINE config 1: one or a few trajectories)."""
    from .fit import Adam, _Trainable
    optimizer = Adam(1e-3) if optimizer is None else optimizer
    # a leaf that is absent (None) or empty (input weights with input_dim = 0) has nothing to train
    frozen = ParameterProperties(trainable=False)
    props = _tree_map2(lambda p, v: p if (isinstance(p, ParameterProperties) and v is not None and np.size(v) > 0) else
                       (frozen if isinstance(p, ParameterProperties) else p), props, params)
    tr = _Trainable(params, props)
    y = np.asarray(emissions)
    batched = y.ndim == 3
    y3 = y if batched else y[None]
    N = y3.shape[0]
    size = float(y3.size)
    tt = None if t_emissions is None else np.asarray(t_emissions)
    uu = None if inputs is None else np.asarray(inputs)
    num_batches = -(-N // batch_size)
    if batch_size >= N:
        shuffle = False
    rng = np.random.default_rng(key if isinstance(key, (int, np.integer)) else 0)
    per_traj = lambda a, idx: a if a is None else (a[idx] if (batched and a.ndim == y.ndim) else a)
    u = tr.to_unconstrained(params)
    state = optimizer.init(u)
    losses, param_hist, grad_hist = [], [], []
    for _ in range(num_epochs):
        order = rng.permutation(N) if shuffle else np.arange(N)
        avg, itr = 0.0, 0
        g_u = np.zeros(tr.size)
        for bi in range(num_batches):
            idx = order[bi * batch_size:(bi + 1) * batch_size]
            cur = tr.from_unconstrained(params, u)
            yb = y3[idx] if batched else y
            ll, g = _loglik_and_grad_with_offsets(cur, yb, per_traj(tt, idx), filter_hyperparams, per_traj(uu, idx), dtype)
            if batched:  # sum the per-trajectory leaves over the minibatch
                g = _tree_map(lambda a: None if a is None else np.asarray(a).sum(axis=0), g)
                ll = float(np.sum(ll))
            scale = N / len(idx)
            loss = -(float(ll) * scale) / size
            g_u = -(tr.pull_back(g, u) * scale) / size
            upd, state = optimizer.update(g_u, state)
            u = u + upd
            avg = (avg * itr + loss) / (itr + 1)
            itr += 1
        losses.append(avg)
        if return_param_history:
            param_hist.append(tr.from_unconstrained(params, u))
        if return_grad_history:
            grad_hist.append(g_u.copy())
    out = [tr.from_unconstrained(params, u), np.asarray(losses)]
    if return_param_history:
        out.append(param_hist)
    if return_grad_history:
        out.append(grad_hist)
    return tuple(out)


def _tree_map2(fn, tree, other):
    if isinstance(tree, tuple) and hasattr(tree, "_fields") and not isinstance(tree, ParameterProperties):
        return type(tree)(*[_tree_map2(fn, v, getattr(other, f)) for f, v in zip(tree._fields, tree)])
    return fn(tree, other)


def _tree_map(fn, tree):
    if isinstance(tree, tuple) and hasattr(tree, "_fields"):
        return type(tree)(*[_tree_map(fn, v) for v in tree])
    return fn(tree)


def cdlgssm_filter(params: ParamsCDLGSSM, emissions, t_emissions=None, filter_hyperparams: Optional[KFHyperParams] = None,
                   inputs=None, dtype=None) -> PosteriorGSSMFiltered:
    """Continuous-discrete Kalman filter (reference: inference.py:555-632)."""
    if _has_offsets(params, inputs):
        return _filter_with_offsets(params, emissions, t_emissions, filter_hyperparams, inputs, dtype)
    return cdnlgssm_filter(_as_nonlinear(params, inputs), emissions, t_emissions, _hyper(filter_hyperparams), dtype=dtype)


def cdlgssm_smoother(params: ParamsCDLGSSM, emissions, t_emissions=None, filter_hyperparams: Optional[KFHyperParams] = None,
                     inputs=None, smoother_type: Optional[str] = "cd_smoother_1", dtype=None) -> PosteriorGSSMSmoothed:
    """Continuous-discrete Kalman smoother (reference: inference.py:694-823): ``cd_smoother_1`` (the default; discrete RTS
    on the pushed-forward (A, Q), Sarkka Alg. 3.17, with ``smoothed_cross_covariances``; state_dim <= 8 on the HIP path)
    or ``cd_smoother_2`` (Alg. 3.18, the continuous-time backward ODE = the EKF smoother of the hot path)."""
    if smoother_type == "cd_smoother_1":
        from . import _ffi
        from .models import _model_block, _opts, _prepare, _squeeze
        nl = _as_nonlinear(params, inputs)
        mdl = _model_block(nl)
        hyper = _hyper(filter_hyperparams)
        opts = _opts(hyper, 1)
        y, t, batched, dtype = _prepare(emissions, t_emissions, hyper, opts, dtype)
        if not _ffi.lib().cdkf_kf_smoother1_supported(_ffi.C.byref(mdl.c)):
            raise NotImplementedError(
                f"smoother_type='cd_smoother_1' runs on the HIP path for state_dim <= 8 (got {mdl.state_dim}); pass "
                "smoother_type='cd_smoother_2' (Sarkka Alg. 3.18)")
        ll, fm, fP, sm, sP, cr, _ = _ffi.kf_smoother1(mdl, opts, t, y, dtype)
        sq = lambda a: _squeeze(a, batched)
        return PosteriorGSSMSmoothed(marginal_loglik=ll if batched else ll[0], filtered_means=sq(fm), filtered_covariances=sq(fP),
                                     smoothed_means=sq(sm), smoothed_covariances=sq(sP), smoothed_cross_covariances=sq(cr))
    if smoother_type != "cd_smoother_2":
        raise ValueError(f"unknown smoother_type {smoother_type!r}")
    return cdnlgssm_smoother(_as_nonlinear(params, inputs), emissions, t_emissions, _hyper(filter_hyperparams), dtype=dtype)


class ContDiscreteLinearGaussianSSM:
    """Continuous-discrete linear Gaussian SSM (reference: models.py:42-365), inference surface only."""

    def __init__(self, state_dim: int, emission_dim: int, input_dim: int = 0, has_dynamics_bias: bool = False,
                 has_emissions_bias: bool = False, diffeqsolve_settings: dict = {}):
        self.state_dim, self.emission_dim, self.input_dim = state_dim, emission_dim, input_dim
        self.has_dynamics_bias, self.has_emissions_bias = has_dynamics_bias, has_emissions_bias
        self._diffeqsolve_settings = diffeqsolve_settings

    @property
    def emission_shape(self):
        return (self.emission_dim,)

    @property
    def inputs_shape(self):
        return (self.input_dim,) if self.input_dim > 0 else None

    def initialize(self, key=None, initial_mean=None, initial_cov=None, dynamics_weights=None, dynamics_bias=None,
                   dynamics_input_weights=None, dynamics_diffusion_coefficient=None, dynamics_diffusion_cov=None,
                   emission_weights=None, emission_bias=None, emission_input_weights=None, emission_cov=None):
        """Dict-based interface of models.py:110-243 ({"params": ..., "props": ...} per argument); defaults:
        F = -0.1 I, L = Qc = 0.1 I, H ~ N(0, 1), R = 0.1 I, m0 = 0, P0 = I."""
        d, m, u = self.state_dim, self.emission_dim, self.input_dim
        rng = np.random.default_rng(0 if key is None else key)
        pp = ParameterProperties

        def pick(arg, default):
            return arg if arg is not None else {"params": default, "props": pp()}

        args = dict(
            initial_mean=pick(initial_mean, np.zeros(d)), initial_cov=pick(initial_cov, np.eye(d)),
            dynamics_weights=pick(dynamics_weights, -0.1 * np.eye(d)),
            dynamics_bias=pick(dynamics_bias, np.zeros(d) if self.has_dynamics_bias else None),
            dynamics_input_weights=pick(dynamics_input_weights, np.zeros((d, u))),
            dynamics_diffusion_coefficient=pick(dynamics_diffusion_coefficient, 0.1 * np.eye(d)),
            dynamics_diffusion_cov=pick(dynamics_diffusion_cov, 0.1 * np.eye(d)),
            emission_weights=pick(emission_weights, rng.standard_normal((m, d))),
            emission_bias=pick(emission_bias, np.zeros(m) if self.has_emissions_bias else None),
            emission_input_weights=pick(emission_input_weights, np.zeros((m, u))),
            emission_cov=pick(emission_cov, 0.1 * np.eye(m)))
        out = []
        for k in ("params", "props"):
            g = lambda name: args[name][k]
            out.append(ParamsCDLGSSM(
                initial=ParamsLGSSMInitial(mean=g("initial_mean"), cov=g("initial_cov")),
                dynamics=ParamsCDLGSSMDynamics(weights=g("dynamics_weights"), bias=g("dynamics_bias"),
                                               input_weights=g("dynamics_input_weights"),
                                               diffusion_coefficient=g("dynamics_diffusion_coefficient"),
                                               diffusion_cov=g("dynamics_diffusion_cov")),
                emissions=ParamsLGSSMEmissions(weights=g("emission_weights"), bias=g("emission_bias"),
                                               input_weights=g("emission_input_weights"), cov=g("emission_cov"))))
        return out[0], out[1]

    def marginal_log_prob(self, params, emissions, t_emissions=None, filter_hyperparams=None, inputs=None, dtype=None):
        if _has_offsets(params, inputs):
            return cdlgssm_filter(params, emissions, t_emissions, filter_hyperparams, inputs, dtype=dtype).marginal_loglik
        return cdnlgssm_filter(_as_nonlinear(params, inputs), emissions, t_emissions, _hyper(filter_hyperparams),
                               output_fields=[], dtype=dtype).marginal_loglik

    def marginal_log_prob_and_grad(self, params, emissions, t_emissions=None, filter_hyperparams=None, inputs=None,
                                   dtype=None):
        """(marginal_log_prob, its gradient as a ParamsCDLGSSM): the pytree ``jax.value_and_grad`` returns in the reference's
        fit_sgd (ssm_temissions.py:550-568); reverse sweep on the device, state and emission dimension <= 8.  The entries of
        the dynamics bias / input weights come from the reverse sweep with mean jumps (``_loglik_and_grad_with_offsets``) whenever the
        model has a bias or is given inputs; without either their entries are returned as they are in ``params``."""
        if _has_offsets(params, inputs) or params.dynamics.bias is not None:
            try:
                return _loglik_and_grad_with_offsets(params, emissions, t_emissions, filter_hyperparams, inputs, dtype)
            except NotImplementedError:
                if _has_offsets(params, inputs):
                    raise  # (a zero bias and no inputs: the bias-free reverse sweep below covers larger models and other solvers)
        ll, g = cdnlgssm_loglik_and_grad_all(_as_nonlinear(params, inputs), emissions, t_emissions, _hyper(filter_hyperparams),
                                             dtype=dtype)
        return ll, _from_nonlinear(g, params)

    def fit_sgd(self, params, props, emissions, t_emissions=None, filter_hyperparams=None, inputs=None, optimizer=None,
                batch_size: int = 1, num_epochs: int = 50, shuffle: bool = False, return_param_history: bool = False,
                return_grad_history: bool = False, key=0, dtype=None, allreduce=None, comm=None):
        """``SSM.fit_sgd`` (ssm_temissions.py:492-600) for the linear model: any of initial mean / cov, dynamics weights,
        diffusion coefficient / cov, emission weights / bias / cov may be trainable (``cd_dynamax_amd.fit.fit_sgd``)."""
        from .fit import fit_sgd
        if _has_offsets(params, inputs) or _trains_offsets(props, params):
            if allreduce is not None or comm is not None:
                raise NotImplementedError("fit_sgd of a linear model with a dynamics bias / inputs runs in one process")
            return _fit_sgd_with_offsets(self, params, props, emissions, t_emissions, filter_hyperparams, inputs, optimizer, batch_size,
                                         num_epochs, shuffle, return_param_history, return_grad_history, key, dtype)
        nl = ContDiscreteNonlinearGaussianSSM(self.state_dim, self.emission_dim)
        out = fit_sgd(nl, _as_nonlinear(params, inputs), _props_as_nonlinear(props, params), emissions, t_emissions,
                      _hyper(filter_hyperparams), None, optimizer, batch_size, num_epochs, shuffle, return_param_history,
                      return_grad_history, key, dtype, allreduce, comm)
        out = list(out)
        out[0] = _from_nonlinear(out[0], params)
        if return_param_history:
            out[2] = [_from_nonlinear(p, params) for p in out[2]]
        return tuple(out)

    def log_prior(self, params) -> float:
        return 0.0

    def fit_mcmc(self, initial_params, props, emissions, t_emissions=None, filter_hyperparams=None, inputs=None,
                 n_mcmc_samples: int = 500, mcmc_algorithm=None, verbose: bool = True, key=0, dtype=None,
                 return_info: bool = False):
        """``SSM.fit_mcmc`` (ssm_temissions.py:601-777) for the linear model (``cd_dynamax_amd.mcmc.fit_mcmc``); the sample
        sets come back in this model's parameter structure, every leaf with a leading sample axis."""
        from .mcmc import fit_mcmc
        nl = ContDiscreteNonlinearGaussianSSM(self.state_dim, self.emission_dim)
        out = list(fit_mcmc(nl, _as_nonlinear(initial_params, inputs), _props_as_nonlinear(props, initial_params), emissions,
                            t_emissions, _hyper(filter_hyperparams), None, n_mcmc_samples, mcmc_algorithm, verbose, key, dtype,
                            return_info))
        for i in (0, 1):
            count = out[i].initial.mean.params.shape[0]
            rep = lambda v: None if v is None else np.broadcast_to(np.asarray(v), (count,) + np.shape(v)).copy()
            like = initial_params._replace(
                dynamics=initial_params.dynamics._replace(bias=rep(initial_params.dynamics.bias),
                                                          input_weights=rep(initial_params.dynamics.input_weights)),
                emissions=initial_params.emissions._replace(input_weights=rep(initial_params.emissions.input_weights)))
            out[i] = _from_nonlinear(out[i], like)
        return tuple(out)

    def filter(self, params, emissions, t_emissions=None, filter_hyperparams=None, inputs=None, dtype=None):
        return cdlgssm_filter(params, emissions, t_emissions, filter_hyperparams, inputs, dtype=dtype)

    def smoother(self, params, emissions, t_emissions=None, filter_hyperparams=None, inputs=None,
                 smoother_type="cd_smoother_1", dtype=None):
        return cdlgssm_smoother(params, emissions, t_emissions, filter_hyperparams, inputs, smoother_type, dtype=dtype)
