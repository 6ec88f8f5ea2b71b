"""fit_sgd for the drift parameters: the training loop of /root/reference/src/ssm_temissions.py:492-600 (loss
:550-568, loop utils/optimize_utils.py:48-140 / dynamax run_sgd) with the value-and-gradient of the loss computed by
the HIP sweep ``cdkf_ekf_loglik_grad_*`` instead of ``jax.value_and_grad`` through the filter.

Scope: the trainable leaves must be fields of ``params.dynamics.drift`` (LearnableLorenz63 / LearnableLinear / LearnableMLP -- the
set-up of the reference's own SGD timer, test_scripts/timers/timer_sgd.py:38-66, which freezes everything else) and
carry no constrainer.  Any other trainable leaf raises NotImplementedError: there is no gradient for it here, and
silently freezing it would change the optimisation problem.

Data stay resident: every minibatch is uploaded once (time-major [T,w,B], the layout the gradient kernels coalesce
on), the per-trajectory log-likelihoods and gradients are reduced on the device, and 1 + n_theta doubles come back
per step.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, List, Optional

import numpy as np

from . import _ffi
from .params import EKFHyperParams, LearnableLinear, LearnableLorenz63, LearnableMLP, ParameterProperties


class Adam:
    """optax.adam(learning_rate, b1, b2, eps): m, v moment estimates with bias correction, update
    -lr * m_hat / (sqrt(v_hat) + eps).  (The reference's default optimizer, ssm_temissions.py:502.)"""

    def __init__(self, learning_rate: float = 1e-3, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8):
        self.lr, self.b1, self.b2, self.eps = learning_rate, b1, b2, eps

    def init(self, theta: np.ndarray):
        return {"count": 0, "mu": np.zeros_like(theta), "nu": np.zeros_like(theta)}

    def update(self, grads: np.ndarray, state):
        c = state["count"] + 1
        mu = self.b1 * state["mu"] + (1 - self.b1) * grads
        nu = self.b2 * state["nu"] + (1 - self.b2) * grads * grads
        mu_hat = mu / (1 - self.b1 ** c)
        nu_hat = nu / (1 - self.b2 ** c)
        return -self.lr * mu_hat / (np.sqrt(nu_hat) + self.eps), {"count": c, "mu": mu, "nu": nu}


class SGD:
    """optax.sgd(learning_rate): update = -lr * grads."""

    def __init__(self, learning_rate: float):
        self.lr = learning_rate

    def init(self, theta):
        return None

    def update(self, grads, state):
        return -self.lr * grads, state


def _leaves(tree, prefix=""):
    """(path, leaf) pairs of a props tree: NamedTuples are interior nodes, ParameterProperties are leaves."""
    if isinstance(tree, ParameterProperties):
        yield prefix, tree
    elif isinstance(tree, tuple) and hasattr(tree, "_fields"):
        for f in tree._fields:
            yield from _leaves(getattr(tree, f), f"{prefix}.{f}" if prefix else f)
    elif tree is None:
        return
    else:
        raise TypeError(f"unexpected entry of type {type(tree).__name__} at props.{prefix}")


def _drift_theta(drift) -> np.ndarray:
    if isinstance(drift, LearnableLorenz63):
        return np.array([drift.sigma, drift.rho, drift.beta], dtype=np.float64)
    if isinstance(drift, LearnableLinear):
        return np.concatenate([np.asarray(drift.weights, np.float64).ravel(), np.asarray(drift.bias, np.float64).ravel()])
    if isinstance(drift, LearnableMLP):
        return np.concatenate([np.asarray(a, np.float64).ravel() for a in drift])
    raise NotImplementedError(f"fit_sgd: no gradient kernel for a drift of type {type(drift).__name__}")


def _drift_from_theta(drift, theta: np.ndarray):
    if isinstance(drift, LearnableLorenz63):
        return LearnableLorenz63(sigma=float(theta[0]), rho=float(theta[1]), beta=float(theta[2]))
    if isinstance(drift, LearnableMLP):
        parts, off = [], 0
        for a in drift:
            shp = np.asarray(a).shape
            parts.append(theta[off:off + int(np.prod(shp))].reshape(shp).copy())
            off += int(np.prod(shp))
        return LearnableMLP(*parts)
    d = np.asarray(drift.weights).shape[0]
    return LearnableLinear(weights=theta[: d * d].reshape(d, d).copy(), bias=theta[d * d:].copy())


def _trainable_mask(params, props) -> np.ndarray:
    """Boolean mask over theta; raises for trainable leaves outside the drift or with a constrainer."""
    drift = params.dynamics.drift
    masks = {}
    for path, leaf in _leaves(props):
        if not leaf.trainable:
            continue
        if not path.startswith("dynamics.drift."):
            raise NotImplementedError(
                f"fit_sgd: props.{path} is trainable, but the HIP path differentiates the log-likelihood w.r.t. the drift "
                "parameters only; set trainable=False on it (as test_scripts/timers/timer_sgd.py does)")
        if leaf.constrainer is not None:
            raise NotImplementedError(f"fit_sgd: constrainer on props.{path} is not supported")
        masks[path.split(".")[-1]] = True
    if isinstance(drift, LearnableLorenz63):
        return np.array([masks.get(k, False) for k in ("sigma", "rho", "beta")])
    if isinstance(drift, LearnableLinear):
        d = np.asarray(drift.weights).shape[0]
        return np.concatenate([np.full(d * d, masks.get("weights", False)), np.full(d, masks.get("bias", False))])
    if isinstance(drift, LearnableMLP):
        return np.concatenate([np.full(np.asarray(getattr(drift, f)).size, masks.get(f, False)) for f in drift._fields])
    raise NotImplementedError(f"fit_sgd: no gradient kernel for a drift of type {type(drift).__name__}")


class _ResidentBatch:
    """One minibatch on the device: t [T,B] (or [T] shared), y [T,m,B], and the output buffers."""

    def __init__(self, y: np.ndarray, t: np.ndarray, t_shared: bool, n_theta: int, dtype):
        B, T, m = y.shape
        self.B, self.T = B, T
        self.t = _ffi.DeviceArray.from_numpy(np.ascontiguousarray(t if t_shared else t.T, dtype=dtype))
        self.y = _ffi.DeviceArray.from_numpy(np.ascontiguousarray(y.transpose(1, 2, 0), dtype=dtype))
        self.ll = _ffi.DeviceArray((B,), dtype)
        self.grad = _ffi.DeviceArray((B, n_theta), dtype)
        self.status = _ffi.DeviceArray((B,), np.int32)
        self.sums = _ffi.DeviceArray((1 + n_theta,), np.float64)

    def value_and_grad(self, mdl: _ffi.ModelBlock, opts, suffix: str):
        L = _ffi.lib()
        n_theta = mdl.theta.size
        _ffi.check(getattr(L, f"cdkf_ekf_loglik_grad_{suffix}_dev")(
            C.byref(mdl.c), C.byref(opts), self.B, self.T, self.t.ptr, self.y.ptr, self.ll.ptr, self.grad.ptr,
            self.status.ptr, None))
        sums = self.sums.ptr.value
        _ffi.check(getattr(L, f"cdkf_ll_sum_{suffix}_dev")(self.ll.ptr, self.B, C.c_void_p(sums), None))
        _ffi.check(getattr(L, f"cdkf_grad_sum_{suffix}_dev")(self.grad.ptr, self.B, n_theta, C.c_void_p(sums + 8), None))
        _ffi.check(L.cdkf_synchronize(None))
        out = self.sums.numpy()
        return float(out[0]), out[1:]

    def free(self):
        for a in (self.t, self.y, self.ll, self.grad, self.status, self.sums):
            a.free()


def fit_sgd(model, params, props, emissions, t_emissions=None, filter_hyperparams: Optional[Any] = None, inputs=None,
            optimizer=None, batch_size: int = 1, num_epochs: int = 50, shuffle: bool = False,
            return_param_history: bool = False, return_grad_history: bool = False, key=0, dtype=None,
            allreduce=None):
    """Minimise ``-(log_prior + sum_n ll_n * scale) / emissions.size`` over the trainable drift parameters
    (ssm_temissions.py:550-568).  Returns ``(params, losses)`` (+ parameter / gradient histories when requested, one
    entry per epoch as the reference's scan returns them, optimize_utils.py:128-131).

    ``key``: seed of the NumPy generator that permutes the sequences when ``shuffle`` (JAX's PRNG stream is not
    reproduced).  ``allreduce``: optional callable summing a float64 array over data-parallel ranks
    (``distributed.allreduce_sum_array``) -- each rank then passes its own block of sequences and every rank applies the
    same update."""
    from .models import _model_block, _opts, _prepare
    hyper = EKFHyperParams() if filter_hyperparams is None else filter_hyperparams
    if not isinstance(hyper, EKFHyperParams):
        raise NotImplementedError("fit_sgd: gradients are provided for the EKF marginal log-likelihood only")
    optimizer = Adam(1e-3) if optimizer is None else optimizer
    mask = _trainable_mask(params, props)
    opts = _opts(hyper, 1)
    y, t, batched, dtype = _prepare(emissions, t_emissions, hyper, opts, dtype)
    opts.layout = _ffi.LAYOUT_TCN
    suffix = "f32" if dtype == np.float32 else "f64"
    mdl0 = _model_block(params)
    if not _ffi.lib().cdkf_grad_supported(C.byref(mdl0.c), C.byref(opts)):
        raise NotImplementedError(
            f"fit_sgd: no gradient kernel for drift {type(params.dynamics.drift).__name__} with state_dim="
            f"{mdl0.state_dim}, emission_dim={mdl0.emission_dim}, state_order={hyper.state_order}")
    N = y.shape[0]
    size = float(y.size)
    n_theta = mdl0.theta.size
    t_shared = bool(opts.t_shared)
    num_batches = -(-N // batch_size)
    if batch_size >= N:
        shuffle = False
    rng = np.random.default_rng(key if isinstance(key, (int, np.integer)) else 0)

    def build(idx):
        return _ResidentBatch(y[idx], t if t_shared else t[idx], t_shared, n_theta, dtype)

    order = np.arange(N)
    resident: List[_ResidentBatch] = []
    if not shuffle:
        resident = [build(order[b * batch_size:(b + 1) * batch_size]) for b in range(num_batches)]

    theta = _drift_theta(params.dynamics.drift)
    state = optimizer.init(theta)
    losses, theta_hist, grad_hist = [], [], []
    cur = params
    try:
        for _ in range(num_epochs):
            if shuffle:
                perm = rng.permutation(N)
                for b in resident:
                    b.free()
                resident = [build(perm[b * batch_size:(b + 1) * batch_size]) for b in range(num_batches)]
            avg = 0.0
            g_loss = np.zeros(n_theta)
            for itr, batch in enumerate(resident):
                cur = params._replace(dynamics=params.dynamics._replace(drift=_drift_from_theta(params.dynamics.drift, theta)))
                ll_sum, g_sum = batch.value_and_grad(_model_block(cur), opts, suffix)
                if allreduce is not None:
                    red = allreduce(np.concatenate([[ll_sum], g_sum]))
                    ll_sum, g_sum = float(red[0]), red[1:]
                scale = N / batch.B
                loss = -(ll_sum * scale) / size
                g_loss = np.where(mask, -(g_sum * scale) / size, 0.0)
                upd, state = optimizer.update(g_loss, state)
                theta = theta + np.where(mask, upd, 0.0)
                avg = (avg * itr + loss) / (itr + 1)
            losses.append(avg)
            theta_hist.append(theta.copy())
            grad_hist.append(g_loss.copy())
    finally:
        for b in resident:
            b.free()
    drift0 = params.dynamics.drift
    final = params._replace(dynamics=params.dynamics._replace(drift=_drift_from_theta(drift0, theta)))
    out = [final, np.asarray(losses)]
    if return_param_history:
        out.append([_drift_from_theta(drift0, th) for th in theta_hist])
    if return_grad_history:
        out.append([_drift_from_theta(drift0, g) for g in grad_hist])
    return tuple(out)
