"""Drop-in host surface of the CDNLGSSM filtering / smoothing hot path.

Mirrors /root/reference/src/continuous_discrete_nonlinear_gaussian_ssm/models.py:
``cdnlgssm_filter`` (:658-718), ``cdnlgssm_smoother`` (:720-764),
``ContDiscreteNonlinearGaussianSSM.marginal_log_prob`` (:393-408) and ``initialize`` (:169-291); the
``filter`` / ``smoother`` methods have the signature of ``SSM.filter/smoother``
(/root/reference/src/ssm_temissions.py:344-386, which only raise NotImplementedError there).

All arithmetic happens in the HIP library (``_ffi.py`` -> libcdkf_hip.so).  Inputs may carry a
leading trajectory axis: ``emissions [N,T,m]`` and ``t_emissions [N,T,1]`` (or a shared ``[T,1]``)
replace the reference's user-side ``jax.vmap`` (ssm_temissions.py:555-566); outputs then carry the
same leading axis and ``marginal_loglik`` has shape ``[N]``.
"""
from __future__ import annotations

from typing import List, Optional, Tuple, Union

import numpy as np

from . import _ffi
from . import device as _device
from .params import (EKFHyperParams, EnKFHyperParams, GSSMForecast, LearnableCustomDrift, LearnableCustomEmission, LearnableLinear,
                     LearnableLorenz63,
                     LearnableLorenz96,
                     LearnableMatrix, LearnableMLP, LearnableVector, ParameterProperties, ParamsCDNLGSSM,
                     ParamsCDNLGSSMDynamics, ParamsCDNLGSSMEmissions, ParamsLGSSMInitial, PosteriorGSSMFiltered,
                     PosteriorGSSMSmoothed, UKFHyperParams)

_FILTER_FIELDS = ["filtered_means", "filtered_covariances", "predicted_means", "predicted_covariances"]


def _model_block(params: ParamsCDNLGSSM) -> _ffi.ModelBlock:
    """ParamsCDNLGSSM -> cdkf_model.  Python callables cannot cross a C ABI, so the drift and the
    emission function must be instances of the registry classes; anything else is refused loudly."""
    drift = params.dynamics.drift
    hidden = (0, 0)
    if isinstance(drift, LearnableLinear):
        W = np.asarray(drift.weights, dtype=np.float64)
        kind, theta = _ffi.DRIFT_LINEAR, np.concatenate([W.ravel(), np.asarray(drift.bias, np.float64).ravel()])
    elif isinstance(drift, LearnableLorenz63):
        kind, theta = _ffi.DRIFT_LORENZ63, np.array([drift.sigma, drift.rho, drift.beta], dtype=np.float64)
    elif isinstance(drift, LearnableLorenz96):
        kind, theta = _ffi.DRIFT_LORENZ96, np.array([drift.forcing], dtype=np.float64)
    elif isinstance(drift, LearnableMLP):
        kind = _ffi.DRIFT_MLP_TANH
        hidden = (np.asarray(drift.W1).shape[0], np.asarray(drift.W2).shape[0])
        theta = np.concatenate([np.asarray(a, np.float64).ravel() for a in drift])
    elif isinstance(drift, LearnableCustomDrift):
        theta = np.atleast_1d(np.asarray(drift.theta, dtype=np.float64)).ravel()
        d = int(np.asarray(params.initial.mean.f()).shape[0])
        kind = _ffi.register_custom_drift(d, theta.size, drift.f_src, drift.jac_src, drift.divgrad_src)
    else:
        raise NotImplementedError(
            f"drift of type {type(drift).__name__} is not in the HIP drift registry "
            "(LearnableLinear, LearnableLorenz63, LearnableLorenz96, LearnableMLP) and is not a LearnableCustomDrift "
            "(C source compiled at run time; Python callables cannot cross the C ABI)")
    h = params.emissions.emission_function
    if isinstance(h, LearnableCustomEmission):
        # a custom emission runs on the run-time compiled kernels: the drift has to be source as well
        d = int(np.asarray(params.initial.mean.f()).shape[0])
        m = int(np.asarray(params.emissions.emission_cov.f()).shape[0])
        if kind < _ffi.DRIFT_CUSTOM_BASE:
            src = _builtin_drift_source(drift, d)
            if d > 6:   # (above six dimensions the run-time compiled kernels derive the Jacobian from f_src by dual numbers)
                src = (src[0], None, src[2])
            kind = _ffi.register_custom_drift(d, theta.size, *src)
            hidden = (0, 0)
        eta = np.atleast_1d(np.asarray(h.eta, dtype=np.float64)).ravel()
        if eta.size > m * d + m:
            raise ValueError(f"LearnableCustomEmission.eta has {eta.size} entries; at most emission_dim * (state_dim + 1) = {m * d + m}")
        eta = np.concatenate([eta, np.zeros(m * d + m - eta.size)])
        ek = _ffi.register_custom_emission(d, m, h.h_src, h.hjac_src)
        return _ffi.ModelBlock(
            kind, theta, params.dynamics.diffusion_coefficient.f(), params.dynamics.diffusion_cov.f(), eta[:m * d].reshape(m, d),
            eta[m * d:], params.emissions.emission_cov.f(), params.initial.mean.f(), params.initial.cov.f(), hidden, emission_kind=ek)
    if not isinstance(h, LearnableLinear):
        raise NotImplementedError(
            f"emission_function of type {type(h).__name__} is not supported: the HIP path implements the reference's "
            "LearnableLinear emission h(x) = weights @ x + bias, or a LearnableCustomEmission (C source compiled at run time)")
    return _ffi.ModelBlock(
        kind, theta, params.dynamics.diffusion_coefficient.f(), params.dynamics.diffusion_cov.f(), h.weights, h.bias,
        params.emissions.emission_cov.f(), params.initial.mean.f(), params.initial.cov.f(), hidden)


def _builtin_drift_source(drift, d: int):
    """(f_src, jac_src, divgrad_src) of a registry drift, for the run-time compiled kernels (needed when the EMISSION is
    user-defined: those kernels take the whole model as source)."""
    if isinstance(drift, LearnableLorenz63):
        return ("fx[0] = theta[0] * (x[1] - x[0]); fx[1] = x[0] * (theta[1] - x[2]) - x[1]; fx[2] = x[0] * x[1] - theta[2] * x[2];",
                "F[0][0] = -theta[0]; F[0][1] = theta[0]; F[1][0] = theta[1] - x[2]; F[1][1] = R(-1); F[1][2] = -x[0]; "
                "F[2][0] = x[1]; F[2][1] = x[0]; F[2][2] = -theta[2];", "")
    if isinstance(drift, LearnableLinear):
        f = " ".join(f"fx[{i}] = theta[{d * d + i}]" + "".join(f" + theta[{i * d + j}] * x[{j}]" for j in range(d)) + ";" for i in range(d))
        jac = " ".join(f"F[{i}][{j}] = theta[{i * d + j}];" for i in range(d) for j in range(d))
        return f, jac, ""
    if isinstance(drift, LearnableLorenz96):
        f = " ".join(f"fx[{i}] = (x[{(i + 1) % d}] - x[{(i - 2) % d}]) * x[{(i - 1) % d}] - x[{i}] + theta[0];" for i in range(d))
        jac = " ".join(f"F[{i}][{(i + 1) % d}] = x[{(i - 1) % d}]; F[{i}][{(i - 2) % d}] = -x[{(i - 1) % d}]; "
                       f"F[{i}][{(i - 1) % d}] = x[{(i + 1) % d}] - x[{(i - 2) % d}]; F[{i}][{i}] = R(-1);" for i in range(d))
        return f, jac, ""
    raise NotImplementedError(
        f"a LearnableCustomEmission needs the drift as source: {type(drift).__name__} has no source form here (use "
        "LearnableCustomDrift, LearnableLorenz63, LearnableLinear or LearnableLorenz96)")


def _opts(hyperparams, num_iter: int = 1):
    o = _ffi.default_opts()
    settings = dict(getattr(hyperparams, "diffeqsolve_settings", {}) or {})
    # keys that only matter for SDE solves / JAX autodiff in the reference and have no effect on this path
    for k in ("tol_vbt", "adjoint", "key", "debug"):
        settings.pop(k, None)
    unknown = set(settings) - {"dt0", "max_steps", "solver", "stepsize_controller"}
    if unknown:
        raise NotImplementedError(
            f"diffeqsolve_settings {sorted(unknown)} are not supported by the HIP path (explicit Runge-Kutta methods, fixed "
            "steps or PIDController; src/utils/diffrax_utils.py:40-57)")
    ctrl = settings.get("stepsize_controller")
    if ctrl is not None and type(ctrl).__name__ != "ConstantStepSize" and ctrl != "ConstantStepSize":
        if not (hasattr(ctrl, "rtol") and hasattr(ctrl, "atol")):
            raise NotImplementedError(f"diffeqsolve_settings['stepsize_controller'] = {ctrl!r}: ConstantStepSize or PIDController")
        defaults = dict(step_ts=None, jump_ts=None, force_dtmin=True, error_order=None)
        for k, v in defaults.items():
            if hasattr(ctrl, k) and getattr(ctrl, k) is not None and getattr(ctrl, k) != v:
                raise NotImplementedError(f"PIDController.{k} = {getattr(ctrl, k)!r}: only the default ({v!r}) is implemented")
        o.adaptive = 1
        o.rtol, o.atol = float(ctrl.rtol), float(ctrl.atol)
        o.pid_p, o.pid_i, o.pid_d = (float(getattr(ctrl, k, v)) for k, v in (("pcoeff", 0.0), ("icoeff", 1.0), ("dcoeff", 0.0)))
        if getattr(ctrl, "dtmin", None) is not None:
            o.dtmin = float(ctrl.dtmin)
        if getattr(ctrl, "dtmax", None) is not None:
            o.dtmax = float(ctrl.dtmax)
        for k, field in (("safety", "pid_safety"), ("factormin", "pid_factormin"), ("factormax", "pid_factormax")):   # (ABI 110)
            if getattr(ctrl, k, None) is not None:
                setattr(o, field, float(getattr(ctrl, k)))
    solver = settings.get("solver", "dopri5")
    name = solver.lower() if isinstance(solver, str) else type(solver).__name__.lower()   # 'tsit5' or a diffrax.Tsit5() object
    if name not in _ffi.SOLVERS:
        raise NotImplementedError(f"diffeqsolve_settings['solver'] = {solver!r}: choose from {sorted(_ffi.SOLVERS)}")
    o.solver = _ffi.SOLVERS[name]
    if o.adaptive and name not in ("dopri5", "tsit5", "bosh3", "heun"):
        raise NotImplementedError(f"adaptive stepping needs an embedded error estimate: dopri5, tsit5, bosh3 or heun (got {name})")
    o.dt0 = float(settings.get("dt0", 0.01))
    o.max_steps = int(settings.get("max_steps", 100000))
    o.dt_final = float(hyperparams.dt_final)
    o.num_iter = int(num_iter)
    if isinstance(hyperparams, EKFHyperParams):
        if hyperparams.state_order not in _ffi.ORDER:
            raise ValueError(f"EKF hyperparams.state_order = {hyperparams.state_order} not implemented yet")
        o.state_order = _ffi.ORDER[hyperparams.state_order]
        o.cov_rescaling = float(hyperparams.cov_rescaling)
    else:
        o.ukf_alpha, o.ukf_beta, o.ukf_kappa = float(hyperparams.alpha), float(hyperparams.beta), float(hyperparams.kappa)
        if getattr(hyperparams, "sigma_points", False):
            o.flags |= _ffi.FLAG_UKF_SIGMA_POINTS
    return o


def _prepare(emissions, t_emissions, hyperparams, opts, dtype):
    """Normalise (emissions, t_emissions) to y [N,T,m], t [N,T] or [T]; returns (y, t, batched)."""
    y = np.asarray(emissions)
    if dtype is None:
        dtype = np.float32 if y.dtype == np.float32 else np.float64
    dtype = np.dtype(dtype)
    if y.ndim == 1:
        y = y[:, None]
    batched = y.ndim == 3
    if not batched:
        y = y[None]
    N, T, _ = y.shape
    if t_emissions is None:
        # t0 = arange(T), t1 = arange(1, T+1): the last interval has length 1 (inference_ekf.py:247-250)
        t = np.arange(T, dtype=dtype)
        opts.dt_final = 1.0
        opts.t_shared = 1
    else:
        t = np.asarray(t_emissions, dtype=dtype)
        if t.ndim == 3 or (t.ndim == 2 and t.shape[-1] != 1 and batched):
            t = t.reshape(N, T)
            opts.t_shared = 0
        else:
            t = t.reshape(-1)
            if t.shape[0] != T:
                raise ValueError(f"t_emissions has {t.shape[0]} time points but emissions has {T}")
            opts.t_shared = 1
    return np.ascontiguousarray(y, dtype=dtype), np.ascontiguousarray(t), batched, dtype


def _squeeze(a, batched):
    return a if (batched or a is None) else a[0]


def _attach_inputs(mdl, opts, inputs, y, dtype):
    """``inputs`` ([T, d_u], shared by the batch, or [N, T, d_u]) -> cdkf_model.input_dim / cdkf_opts.inputs, resident where ``y`` is.
    The reference hands u = inputs[t0_idx] and the time to the drift and the emission with every call -- f(m, u, t), h(m, u, t)
    (inference_ekf.py:95, 101-114, 277-286; inference_ukf.py:142, 189): a LearnableCustomDrift / LearnableCustomEmission snippet reads
    them as ``u[i]`` and ``t``; the registry drifts ignore both, as the reference's own Learnable* classes do (cdnlgssm_utils.py:50-83)."""
    if inputs is None:
        return
    N, T = int(y.shape[0]), int(y.shape[1])
    if _device.is_device_tensor(y):
        import torch
        u = torch.as_tensor(inputs).to(device=y.device, dtype=y.dtype)
        u = u.reshape(T, -1) if u.ndim <= 2 else u
        if u.ndim == 2:
            u = u[None].expand(N, T, u.shape[-1])
        u = u.contiguous()
        ptr = u.data_ptr()
    else:
        u = np.asarray(inputs, dtype=dtype)
        u = u.reshape(T, -1) if u.ndim <= 2 else u
        if u.ndim == 2:
            u = np.broadcast_to(u, (N,) + u.shape)
        u = np.ascontiguousarray(u)
        ptr = u.ctypes.data
    if tuple(u.shape[:2]) != (N, T):
        raise ValueError(f"inputs has shape {tuple(u.shape)}; expected [T, d_u] or [N, T, d_u] with N = {N}, T = {T}")
    mdl.c.input_dim = int(u.shape[-1])
    opts.inputs = ptr
    mdl._inputs = u   # (keeps the array alive for the call)


def cdnlgssm_filter(
    params: ParamsCDNLGSSM,
    emissions,
    t_emissions=None,
    hyperparams: Optional[Union[EKFHyperParams, EnKFHyperParams, UKFHyperParams]] = EKFHyperParams(),
    inputs=None,
    num_iter: Optional[int] = 1,
    output_fields: Optional[List[str]] = _FILTER_FIELDS,
    dtype=None,
) -> PosteriorGSSMFiltered:
    """Continuous-discrete nonlinear filter; EKF or UKF by the class of ``hyperparams``
    (reference dispatch: models.py:689-716).  ``inputs`` ([T, d_u] or [N, T, d_u]) reach a drift / emission given as source
    (``u``, and the time ``t``: _attach_inputs); the registry drifts ignore them exactly as the reference's shipped Learnable*
    classes ignore ``u``."""
    if isinstance(hyperparams, EKFHyperParams):
        algo = "ekf_filter"
    elif isinstance(hyperparams, UKFHyperParams):
        algo = "ukf_filter"
    elif isinstance(hyperparams, EnKFHyperParams):
        raise NotImplementedError("the ensemble Kalman filter is stochastic and not part of the HIP hot path")
    else:
        raise TypeError(f"unknown filter hyperparams {type(hyperparams).__name__}")
    mdl = _model_block(params)
    opts = _opts(hyperparams, num_iter if algo == "ekf_filter" else 1)
    fields = list(output_fields) if output_fields is not None else []
    want = [f in fields for f in _FILTER_FIELDS]
    if _device.is_device_tensor(emissions):  # data already on the GPU: run in place, return device tensors
        y, t, batched, dtype = _device.prepare(emissions, t_emissions, opts)
        _attach_inputs(mdl, opts, inputs, y, dtype)
        ll, outs, _ = _device.run_device(algo, mdl, opts, t, y, want)
    else:
        y, t, batched, dtype = _prepare(emissions, t_emissions, hyperparams, opts, dtype)
        _attach_inputs(mdl, opts, inputs, y, dtype)
        ll, outs, _ = _ffi.run_host(algo, mdl, opts, t, y, want, dtype)
    out = {name: _squeeze(arr, batched) for name, arr in zip(_FILTER_FIELDS, outs) if arr is not None}
    return PosteriorGSSMFiltered(marginal_loglik=ll if batched else ll[0], **out)


def cdnlgssm_smoother(
    params: ParamsCDNLGSSM,
    emissions,
    t_emissions=None,
    hyperparams: Optional[Union[EKFHyperParams, EnKFHyperParams, UKFHyperParams]] = EKFHyperParams(),
    inputs=None,
    num_iter: Optional[int] = 1,
    dtype=None,
) -> PosteriorGSSMSmoothed:
    """Continuous-discrete EKF (RTS) smoother (reference: models.py:720-764; one pass,
    inference_ekf.py:578-586)."""
    if isinstance(hyperparams, EnKFHyperParams):
        raise ValueError("EnKS not implemented yet")
    if isinstance(hyperparams, UKFHyperParams):
        raise ValueError("UKS not implemented yet")
    if not isinstance(hyperparams, EKFHyperParams):
        raise TypeError(f"unknown smoother hyperparams {type(hyperparams).__name__}")
    if hyperparams.smooth_order != "first":
        raise ValueError(f"EKF hyperparams.smooth_order = {hyperparams.smooth_order} not implemented yet")
    mdl = _model_block(params)
    opts = _opts(hyperparams, 1)
    if _device.is_device_tensor(emissions):
        y, t, batched, dtype = _device.prepare(emissions, t_emissions, opts)
        _attach_inputs(mdl, opts, inputs, y, dtype)
        ll, outs, _ = _device.run_device("ekf_smoother", mdl, opts, t, y, [True] * 4)
    else:
        y, t, batched, dtype = _prepare(emissions, t_emissions, hyperparams, opts, dtype)
        _attach_inputs(mdl, opts, inputs, y, dtype)
        ll, outs, _ = _ffi.run_host("ekf_smoother", mdl, opts, t, y, [True] * 4, dtype)
    fm, fP, sm, sP = (_squeeze(a, batched) for a in outs)
    return PosteriorGSSMSmoothed(marginal_loglik=ll if batched else ll[0], filtered_means=fm, filtered_covariances=fP,
                                 smoothed_means=sm, smoothed_covariances=sP)


def _drift_like(drift, theta_grad):
    """Pack a gradient array [..., n_theta] into the drift's own NamedTuple (the pytree jax.grad would return)."""
    if isinstance(drift, LearnableLorenz63):
        return LearnableLorenz63(sigma=theta_grad[..., 0], rho=theta_grad[..., 1], beta=theta_grad[..., 2])
    if isinstance(drift, LearnableLorenz96):
        return LearnableLorenz96(forcing=theta_grad[..., 0])
    if isinstance(drift, LearnableCustomDrift):
        return drift._replace(theta=theta_grad.reshape(theta_grad.shape[:-1] + np.shape(drift.theta)))
    if isinstance(drift, LearnableMLP):
        lead, parts, off = theta_grad.shape[:-1], [], 0
        for a in drift:
            shp = np.asarray(a).shape
            parts.append(theta_grad[..., off:off + int(np.prod(shp))].reshape(lead + shp))
            off += int(np.prod(shp))
        return LearnableMLP(*parts)
    d = np.asarray(drift.weights).shape[0]
    return LearnableLinear(weights=theta_grad[..., : d * d].reshape(theta_grad.shape[:-1] + (d, d)),
                           bias=theta_grad[..., d * d:])


def _grads_tree(params: ParamsCDNLGSSM, mdl: _ffi.ModelBlock, g_theta, g_model=None) -> ParamsCDNLGSSM:
    """Device gradient blocks ([..., n_theta] and [..., d + 2 d^2 + m d + m + m^2]) -> a ParamsCDNLGSSM of gradients.
    Without the model block the non-drift leaves are zeros."""
    d, m = mdl.state_dim, mdl.emission_dim
    g_theta = np.asarray(g_theta)
    lead = g_theta.shape[:-1]
    if g_model is None:
        g_model = np.zeros(lead + (_ffi.model_grad_size(d, m),), g_theta.dtype)
    off = 0

    def take(*shape):
        nonlocal off
        n = int(np.prod(shape))
        out = g_model[..., off:off + n].reshape(lead + shape)
        off += n
        return out

    g_m0, g_P0, g_LQL, g_H, g_b, g_R = take(d), take(d, d), take(d, d), take(m, d), take(m), take(m, m)
    L, Qc = mdl.L.astype(g_model.dtype), mdl.Qc.astype(g_model.dtype)
    g_L = g_LQL @ L @ Qc.T + np.swapaxes(g_LQL, -1, -2) @ L @ Qc      # LQL = L Qc L^T
    g_Qc = L.T @ g_LQL @ L
    return ParamsCDNLGSSM(
        initial=ParamsLGSSMInitial(mean=LearnableVector(g_m0), cov=LearnableMatrix(g_P0)),
        dynamics=ParamsCDNLGSSMDynamics(drift=_drift_like(params.dynamics.drift, g_theta),
                                        diffusion_coefficient=LearnableMatrix(g_L), diffusion_cov=LearnableMatrix(g_Qc),
                                        approx_order=0.0),
        emissions=ParamsCDNLGSSMEmissions(emission_function=_emission_like(params.emissions.emission_function, g_H, g_b),
                                          emission_cov=LearnableMatrix(g_R)))


def _emission_like(h, g_H, g_b):
    """The emission leaf of a gradient tree: LearnableLinear(weights, bias), or -- for a LearnableCustomEmission, whose parameter vector
    eta travels in the H / bias block of the C model -- the same class with d ll / d eta."""
    if isinstance(h, LearnableCustomEmission):
        n = int(np.atleast_1d(np.asarray(h.eta)).size)
        flat = np.concatenate([g_H.reshape(g_H.shape[:-2] + (-1,)), g_b], axis=-1)
        return LearnableCustomEmission(eta=flat[..., :n], h_src=h.h_src, hjac_src=h.hjac_src, py_h=h.py_h)
    return LearnableLinear(weights=g_H, bias=g_b)


def cdnlgssm_loglik_and_grad(
    params: ParamsCDNLGSSM,
    emissions,
    t_emissions=None,
    hyperparams: EKFHyperParams = EKFHyperParams(),
    inputs=None,
    dtype=None,
):
    """EKF (or, with ``UKFHyperParams``, unscented) marginal log-likelihood and its gradient w.r.t. the drift parameters -- the drift block of what
    ``jax.value_and_grad(_loss_fn)`` returns in the reference's fit_sgd (ssm_temissions.py:550-568), per trajectory and
    un-negated / un-normalised.  Returns ``(ll, grad)``: ``ll`` as ``marginal_log_prob`` (``[N]`` for batched emissions),
    ``grad`` an instance of the drift's class whose fields hold d ll / d field (leading ``[N]`` when batched).

    LearnableLorenz63 / LearnableLinear at the register-resident shapes: forward sensitivities inside the sweep,
    ``state_order`` first or second.  Any registry drift with state_dim, emission_dim <= 8 (LearnableMLP: hidden <= 64; its
    ``state_order='second'`` mean term 0.5 P grad(div f) is reversed too): forward + reverse sweep (discrete adjoint).  LearnableLorenz96 /
    LearnableLinear beyond eight dimensions (up to 43 in float64, 62 in float32): the workgroup-per-trajectory reverse sweep.  Anything
    else raises (no finite-difference fallback)."""
    ukf = isinstance(hyperparams, UKFHyperParams)
    if not ukf and not isinstance(hyperparams, EKFHyperParams):
        raise NotImplementedError("gradients are provided for the EKF and the UKF marginal log-likelihood (the ensemble filter is stochastic)")
    mdl = _model_block(params)
    opts = _opts(hyperparams, 1)
    on_device = _device.is_device_tensor(emissions)
    if on_device:
        y, t, batched, dtype = _device.prepare(emissions, t_emissions, opts)
    else:
        y, t, batched, dtype = _prepare(emissions, t_emissions, hyperparams, opts, dtype)
    _attach_inputs(mdl, opts, inputs, y, dtype)
    supported = _ffi.lib().cdkf_ukf_grad_supported if ukf else _ffi.lib().cdkf_grad_supported
    if ukf and not supported(_ffi.C.byref(mdl.c), _ffi.C.byref(opts)) and \
            _ffi.lib().cdkf_ukf_grad_all_supported(_ffi.C.byref(mdl.c), _ffi.C.byref(opts)):
        # no forward-sensitivity kernel at this shape (Lorenz-96, larger linear models): the drift block of the reverse sweeps' result
        ll, g = cdnlgssm_loglik_and_grad_all(params, emissions, t_emissions, hyperparams, inputs, dtype)
        return ll, g.dynamics.drift
    if not supported(_ffi.C.byref(mdl.c), _ffi.C.byref(opts)):
        raise NotImplementedError(
            f"no {'unscented-filter ' if ukf else ''}gradient kernel for drift {type(params.dynamics.drift).__name__} with "
            f"state_dim={mdl.state_dim}, emission_dim={mdl.emission_dim}"
            + ("" if ukf else f", state_order={hyperparams.state_order}")
            + (" (the unscented gradient: closed forms for LearnableLorenz63 / LearnableLorenz96 / LearnableLinear, the tangent sweep of the literal "
               "recursion for everything else with state_dim, emission_dim <= 16 and the default solver)" if ukf else ""))
    if on_device:  # ll and the per-trajectory gradient stay on the device (torch tensors)
        ll, grad, _ = _device.loglik_grad_device(mdl, opts, t, y, False, ukf=ukf)
    else:
        ll, grad, _ = _ffi.loglik_grad(mdl, opts, t, y, dtype, ukf=ukf)
    if not batched:
        ll, grad = ll[0], grad[0]
    return ll, _drift_like(params.dynamics.drift, grad)


def cdnlgssm_loglik_and_grad_all(
    params: ParamsCDNLGSSM,
    emissions,
    t_emissions=None,
    hyperparams: Union[EKFHyperParams, UKFHyperParams] = EKFHyperParams(),
    inputs=None,
    dtype=None,
    num_iter: int = 1,
):
    """EKF (or, with ``UKFHyperParams``, unscented) marginal log-likelihood and its gradient w.r.t. EVERY parameter: returns ``(ll, grads)`` with ``grads`` a
    ``ParamsCDNLGSSM`` of the same structure as ``params`` (what ``jax.grad`` of ``marginal_log_prob`` returns in the
    reference, ssm_temissions.py:550-568), leaves carrying a leading ``[N]`` for batched emissions.

    One forward and one reverse sweep on the device (cdkf_ekf_loglik_grad_all_*): state and emission dimension <= 8 for any
    registry drift, ``state_order`` first or second.  Gradients of the symmetric matrices (initial covariance, diffusion
    covariance, emission covariance) are symmetric cotangents: exact for symmetric perturbations, i.e. for any symmetric
    parametrisation such as the reference's ``RealToPSDBijector``."""
    ukf = isinstance(hyperparams, UKFHyperParams)
    if not ukf and not isinstance(hyperparams, EKFHyperParams):
        raise NotImplementedError("gradients are provided for the EKF and the UKF marginal log-likelihood (the ensemble filter is stochastic)")
    mdl = _model_block(params)
    opts = _opts(hyperparams, 1 if ukf else num_iter)  # (num_iter: the EKF's update iterations, as cdnlgssm_filter takes them)
    on_device = _device.is_device_tensor(emissions)
    if on_device:
        y, t, batched, dtype = _device.prepare(emissions, t_emissions, opts)
    else:
        y, t, batched, dtype = _prepare(emissions, t_emissions, hyperparams, opts, dtype)
    _attach_inputs(mdl, opts, inputs, y, dtype)
    if ukf:
        # the unscented filter, every leaf (cdkf_ukf_loglik_grad_all_*): the reverse sweeps over its moment equations in closed form --
        # exact for the quadratic Lorenz-63 / Lorenz-96 drifts and the linear one -- and, for every other model (an MLP drift, a drift or
        # an emission given as source), forward mode through the literal sigma-point recursion (inference_ukf.py:93-203 differentiated by JAX)
        if not _ffi.lib().cdkf_ukf_grad_all_supported(_ffi.C.byref(mdl.c), _ffi.C.byref(opts)):
            raise NotImplementedError(
                f"no unscented-filter gradient kernel for drift {type(params.dynamics.drift).__name__} with state_dim={mdl.state_dim}, "
                f"emission_dim={mdl.emission_dim} (closed forms: LearnableLorenz63, LearnableLorenz96, LearnableLinear with a linear emission; "
                "the tangent sweep of the literal recursion: any drift / emission with state_dim, emission_dim <= 16; default solver)")
        if on_device:
            ll, gth, _, gm = (x.cpu().numpy() for x in _device.loglik_grad_device(mdl, opts, t, y, True, ukf=True))
        else:
            ll, gth, _, gm = _ffi.loglik_grad(mdl, opts, t, y, dtype, with_model=True, ukf=True)
        if not batched:
            ll, gth, gm = ll[0], gth[0], gm[0]
        return ll, _grads_tree(params, mdl, gth, gm)
    if not _ffi.lib().cdkf_grad_all_supported(_ffi.C.byref(mdl.c), _ffi.C.byref(opts)):
        raise NotImplementedError(
            f"no reverse-sweep kernel for drift {type(params.dynamics.drift).__name__} with state_dim={mdl.state_dim}, "
            f"emission_dim={mdl.emission_dim}, state_order={hyperparams.state_order} (needs state and emission dimensions <= 8 -- LearnableLorenz96 / LearnableLinear: <= 43 in float64, 62 in float32 --, "
            "MLP hidden layers <= 64, state_order 'first' or 'second'; num_iter > 1 for state and emission dimensions <= 8; beyond those, up to "
            "state / emission dimension 16 with the default solver: the tangent sweep of the literal recursion)")
    if on_device:  # the sweeps run on the device tensors; the (small) gradient blocks are packed on the host
        ll, gth, _, gm = (x.cpu().numpy() for x in _device.loglik_grad_device(mdl, opts, t, y, True))
    else:
        ll, gth, _, gm = _ffi.loglik_grad(mdl, opts, t, y, dtype, with_model=True)
    if not batched:
        ll, gth, gm = ll[0], gth[0], gm[0]
    return ll, _grads_tree(params, mdl, gth, gm)


def cdnlgssm_forecast(
    params: ParamsCDNLGSSM,
    init_forecast,
    t_init,
    t_forecast=None,
    hyperparams: Optional[Union[EKFHyperParams, EnKFHyperParams, UKFHyperParams]] = EKFHyperParams(),
    inputs=None,
    output_fields: Optional[List[str]] = ["forecasted_state_means", "forecasted_state_covariances"],
    dtype=None,
) -> GSSMForecast:
    """Forecast the Gaussian state distribution over ``t_forecast`` from ``init_forecast`` at ``t_init``: repeated
    ``_predict`` without measurement updates (reference: models.py:767-936 -> forecast_extended_kalman_filter,
    inference_ekf.py:679-766 / forecast_unscented_kalman_filter, inference_ukf.py:409-505).

    ``init_forecast`` is a ``(mean, covariance)`` pair or any object with ``.mean()`` / ``.covariance()`` (the reference
    passes a TFP MultivariateNormalFullCovariance); a leading trajectory axis on it and on ``t_init`` / ``t_forecast``
    forecasts a batch.  Only the distribution forecast is on the HIP path; path (SDE sample) forecasts and emission
    moments are not.
    """
    if t_forecast is None:
        raise ValueError("t_forecast must be provided for forecasting")
    if isinstance(hyperparams, EKFHyperParams):
        algo = "ekf_filter"
    elif isinstance(hyperparams, UKFHyperParams):
        algo = "ukf_filter"
    else:
        raise NotImplementedError("only EKF / UKF distribution forecasts are on the HIP path")
    if hasattr(init_forecast, "mean") and callable(init_forecast.mean):
        m_init, P_init = np.asarray(init_forecast.mean()), np.asarray(init_forecast.covariance())
    else:
        m_init, P_init = (np.asarray(a) for a in init_forecast)
    if m_init.ndim > 1:
        raise NotImplementedError("one initial distribution per call (the model block holds a single (m0, P0))")
    fields = list(output_fields) if output_fields is not None else []
    unknown = set(fields) - {"forecasted_state_means", "forecasted_state_covariances"}
    if unknown:
        raise NotImplementedError(f"forecast fields {sorted(unknown)} are not produced by the HIP path")
    start = params._replace(initial=ParamsLGSSMInitial(LearnableVector(m_init), LearnableMatrix(P_init)))
    mdl = _model_block(start)
    opts = _opts(hyperparams, 1)
    opts.forecast = 1
    tf = np.asarray(t_forecast, dtype=np.float64)
    batched = tf.ndim == 3
    tf = tf.reshape(tf.shape[0], -1) if batched else tf.reshape(1, -1)
    ti = np.broadcast_to(np.asarray(t_init, dtype=np.float64).reshape(-1, 1), (tf.shape[0], 1))
    t = np.concatenate([ti, tf], axis=1)  # [N, 1 + n]: t0 = [t_init, t_forecast[:-1]], t1 = t_forecast
    dtype = np.dtype(np.float64 if dtype is None else dtype)
    y = np.zeros((t.shape[0], t.shape[1], mdl.emission_dim), dtype)  # ignored in forecast mode
    opts.t_shared = 0
    want = [False, False, "forecasted_state_means" in fields, "forecasted_state_covariances" in fields]
    _, outs, _ = _ffi.run_host(algo, mdl, opts, t.astype(dtype), y, want, dtype)
    fm = None if outs[2] is None else _squeeze(outs[2][:, :-1], batched)
    fP = None if outs[3] is None else _squeeze(outs[3][:, :-1], batched)
    return GSSMForecast(forecasted_state_means=fm, forecasted_state_covariances=fP)


def cdnlgssm_emissions(params: ParamsCDNLGSSM, t_states, state_means, state_covs=None, inputs=None, hyperparams=None,
                       key=None, dtype=None):
    """Emission moments of Gaussian state marginals (reference: models.py:939-1047 ->
    emissions_extended_kalman_filter inference_ekf.py:768-855 / emissions_unscented_kalman_filter
    inference_ukf.py:507-612): ``(H m + b, H P H^T + R)`` per time point; with ``state_covs=None`` the states are point
    estimates and only the means are returned (second element None).  For the registry's linear emission the EKF and
    UKF versions coincide, so ``hyperparams`` selects nothing and ``t_states`` is accepted for signature parity; for a
    ``LearnableCustomEmission`` (state / emission dimension <= 16) ``hyperparams`` picks the extended (jacfwd) or the unscented (sigma
    points) version and ``t_states`` / ``inputs`` reach the statements (a kernel compiled at run time)."""
    if t_states is None:
        raise ValueError("t_states must be provided for forecasting")
    if isinstance(hyperparams, EnKFHyperParams):
        raise NotImplementedError("ensemble emissions are stochastic and not part of the HIP path")
    mu = np.asarray(state_means)
    if dtype is None:
        dtype = np.float32 if mu.dtype == np.float32 else np.float64
    mdl = _model_block(params)
    if isinstance(params.emissions.emission_function, LearnableCustomEmission):
        # an emission given as source: h(m, u, t) and jacfwd(h) P jacfwd(h)^T + R (extended), or the sigma points of (m, P) through h
        # (unscented) -- the two reference functions differ here; t_states [T, 1] (or [N, T, 1]) and inputs rows reach the statements
        ukf = isinstance(hyperparams, UKFHyperParams)
        opts = _opts(hyperparams if hyperparams is not None else EKFHyperParams(), 1)
        t = np.asarray(t_states, dtype=np.float64)
        t = t.reshape(t.shape[:-1]) if t.ndim >= 2 and t.shape[-1] == 1 else t
        return _ffi.custom_emission_moments(mdl, opts, ukf, t, None if inputs is None else np.asarray(inputs), mu,
                                            None if state_covs is None else np.asarray(state_covs), dtype)
    return _ffi.emission_moments(mdl, mu, None if state_covs is None else np.asarray(state_covs), dtype)


class ContDiscreteNonlinearGaussianSSM:
    """Continuous-discrete nonlinear Gaussian SSM (reference: models.py:117-408), restricted to the
    inference surface of the hot path: ``initialize``, ``marginal_log_prob``, ``filter``, ``smoother``."""

    def __init__(self, state_dim: int, emission_dim: int, input_dim: int = 0, diffeqsolve_settings: dict = {}):
        self.state_dim = state_dim
        self.emission_dim = emission_dim
        self.input_dim = 0
        self._diffeqsolve_settings = diffeqsolve_settings

    @property
    def emission_shape(self):
        return (self.emission_dim,)

    @property
    def inputs_shape(self):
        return (self.input_dim,) if self.input_dim > 0 else None

    @property
    def diffeqsolve_settings(self):
        return self._diffeqsolve_settings

    def initialize(
        self,
        key=None,
        initial_mean: dict = None,
        initial_cov: dict = None,
        dynamics_drift: dict = None,
        dynamics_diffusion_coefficient: dict = None,
        dynamics_diffusion_cov: dict = None,
        dynamics_approx_order: Optional[float] = 2.0,
        emission_function: dict = None,
        emission_cov: dict = None,
    ) -> Tuple[ParamsCDNLGSSM, ParamsCDNLGSSM]:
        """Same dict-based interface and defaults as models.py:169-291 (``key`` seeds the default
        emission weights through NumPy instead of jax.random)."""
        d, m = self.state_dim, self.emission_dim
        rng = np.random.default_rng(0 if key is None else key)
        _initial_mean = {"params": LearnableVector(np.zeros(d)), "props": LearnableVector(ParameterProperties(False))}
        _initial_cov = {"params": LearnableMatrix(np.eye(d)), "props": LearnableMatrix(ParameterProperties(False))}
        _drift = {"params": LearnableLinear(weights=-0.1 * np.eye(d), bias=np.zeros(d)),
                  "props": LearnableLinear(weights=ParameterProperties(False), bias=ParameterProperties(False))}
        _L = {"params": LearnableMatrix(0.1 * np.eye(d)), "props": LearnableMatrix(ParameterProperties(False))}
        _Q = {"params": LearnableMatrix(0.1 * np.eye(d)), "props": LearnableMatrix(ParameterProperties(False))}
        _h = {"params": LearnableLinear(weights=rng.standard_normal((m, d)), bias=np.zeros(m)),
              "props": LearnableLinear(weights=ParameterProperties(False), bias=ParameterProperties(False))}
        _R = {"params": LearnableMatrix(0.1 * np.eye(m)), "props": LearnableMatrix(ParameterProperties(False))}

        def wrap(x, x0, cls):
            if x is None:
                return x0
            # the reference also accepts raw arrays for the initial mean / cov in older call sites
            if not hasattr(x["params"], "f"):
                return {"params": cls(x["params"]), "props": cls(x["props"])}
            return x

        initial_mean = wrap(initial_mean, _initial_mean, LearnableVector)
        initial_cov = wrap(initial_cov, _initial_cov, LearnableMatrix)
        dynamics_drift = dynamics_drift if dynamics_drift is not None else _drift
        dynamics_diffusion_coefficient = wrap(dynamics_diffusion_coefficient, _L, LearnableMatrix)
        dynamics_diffusion_cov = wrap(dynamics_diffusion_cov, _Q, LearnableMatrix)
        approx = {"params": dynamics_approx_order if dynamics_approx_order is not None else 2.0,
                  "props": ParameterProperties(False)}
        emission_function = emission_function if emission_function is not None else _h
        emission_cov = wrap(emission_cov, _R, LearnableMatrix)
        out = {}
        for k in ("params", "props"):
            out[k] = ParamsCDNLGSSM(
                initial=ParamsLGSSMInitial(mean=initial_mean[k], cov=initial_cov[k]),
                dynamics=ParamsCDNLGSSMDynamics(drift=dynamics_drift[k],
                                                diffusion_coefficient=dynamics_diffusion_coefficient[k],
                                                diffusion_cov=dynamics_diffusion_cov[k], approx_order=approx[k]),
                emissions=ParamsCDNLGSSMEmissions(emission_function=emission_function[k], emission_cov=emission_cov[k]),
            )
        return out["params"], out["props"]

    def marginal_log_prob(self, params, emissions, t_emissions=None, filter_hyperparams=EKFHyperParams(), inputs=None,
                          dtype=None):
        """models.py:393-408: the filter's marginal log-likelihood (``[N]`` for batched emissions)."""
        post = cdnlgssm_filter(params=params, emissions=emissions, t_emissions=t_emissions,
                               hyperparams=filter_hyperparams, inputs=inputs, output_fields=[], dtype=dtype)
        return post.marginal_loglik

    def marginal_log_prob_and_grad(self, params, emissions, t_emissions=None, filter_hyperparams=EKFHyperParams(),
                                   inputs=None, dtype=None):
        """(marginal_log_prob, d marginal_log_prob / d drift parameters): see ``cdnlgssm_loglik_and_grad``."""
        return cdnlgssm_loglik_and_grad(params, emissions, t_emissions, filter_hyperparams, inputs, dtype=dtype)

    def fit_sgd(self, params, props, emissions, t_emissions=None, filter_hyperparams=None, inputs=None, optimizer=None,
                batch_size: int = 1, num_epochs: int = 50, shuffle: bool = False, return_param_history: bool = False,
                return_grad_history: bool = False, key=0, dtype=None, allreduce=None, comm=None):
        """ssm_temissions.py:492-600 for the trainable drift parameters; see ``cd_dynamax_amd.fit.fit_sgd``."""
        from .fit import fit_sgd
        return fit_sgd(self, params, props, emissions, t_emissions, filter_hyperparams, inputs, optimizer, batch_size,
                       num_epochs, shuffle, return_param_history, return_grad_history, key, dtype, allreduce, comm)

    def log_prior(self, params) -> float:
        """SSM.log_prior (ssm_temissions.py:152-161): the reference's models define no prior."""
        return 0.0

    def fit_mcmc(self, initial_params, props, emissions, t_emissions=None, filter_hyperparams=None, inputs=None,
                 n_mcmc_samples: int = 500, mcmc_algorithm=None, verbose: bool = True, key=0, dtype=None,
                 return_info: bool = False):
        """ssm_temissions.py:601-777 (HMC / NUTS with window adaptation); see ``cd_dynamax_amd.mcmc.fit_mcmc``."""
        from .mcmc import fit_mcmc
        return fit_mcmc(self, initial_params, props, emissions, t_emissions, filter_hyperparams, inputs, n_mcmc_samples,
                        mcmc_algorithm, verbose, key, dtype, return_info)

    def filter(self, params, emissions, t_emissions=None, filter_hyperparams=EKFHyperParams(), inputs=None,
               dtype=None) -> PosteriorGSSMFiltered:
        return cdnlgssm_filter(params, emissions, t_emissions, filter_hyperparams, inputs, dtype=dtype)

    def smoother(self, params, emissions, t_emissions=None, filter_hyperparams=EKFHyperParams(), inputs=None,
                 dtype=None) -> PosteriorGSSMSmoothed:
        return cdnlgssm_smoother(params, emissions, t_emissions, filter_hyperparams, inputs, dtype=dtype)
