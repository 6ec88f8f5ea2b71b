"""fit_sgd for the drift parameters: the training loop of /root/reference/src/ssm_temissions.py:492-600 (loss
:550-568, loop utils/optimize_utils.py:48-140 / dynamax run_sgd) with the value-and-gradient of the loss computed by
the HIP sweep ``cdkf_ekf_loglik_grad_*`` instead of ``jax.value_and_grad`` through the filter.

Scope.  Trainable drift leaves only (the set-up of the reference's own SGD timer, test_scripts/timers/timer_sgd.py:38-66):
the drift-gradient entry point -- forward sensitivities for the register-resident Lorenz-63 / linear shapes, the reverse
sweep otherwise.  Any other trainable leaf (initial mean / covariance, diffusion coefficient / covariance, emission
weights / bias / covariance): the all-parameter reverse sweep (state and emission dimension <= 8; Lorenz-96 and linear drifts up to 43 in float64, 62 in float32).  Constrainers: ``None``
or ``bijectors.RealToPSDBijector`` (the reference's choice for covariances); optimisation runs in the unconstrained space
as in the reference (to_unconstrained / from_unconstrained, dynamax/parameters.py:53-90).  What the kernels cannot
differentiate raises NotImplementedError -- silently freezing a leaf would change the optimisation problem.

Data stay resident: every minibatch is uploaded once (time-major [T,w,B], the layout the gradient kernels coalesce
on), the per-trajectory log-likelihoods and gradients are reduced on the device, and 1 + n_theta doubles come back
per step.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, List, Optional

import numpy as np

from . import _ffi
from .params import EKFHyperParams, ParameterProperties, UKFHyperParams


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
    """(path, leaf) pairs of a props tree: NamedTuples are interior nodes, ParameterProperties are leaves; anything else
    (e.g. the float ``approx_order``) is not a parameter."""
    if isinstance(tree, ParameterProperties):
        yield prefix, tree
    elif isinstance(tree, tuple) and hasattr(tree, "_fields"):
        for f in tree._fields:
            yield from _leaves(getattr(tree, f), f"{prefix}.{f}" if prefix else f)


def _get(tree, path: str):
    for f in path.split("."):
        tree = getattr(tree, f)
    return tree


def _set(tree, path: str, value):
    head, _, rest = path.partition(".")
    return tree._replace(**{head: _set(getattr(tree, head), rest, value) if rest else value})


def _drift_theta(drift) -> np.ndarray:
    return np.concatenate([np.asarray(a, np.float64).ravel() for a in drift])


def _drift_from_theta(drift, theta: np.ndarray):
    parts, off = [], 0
    for a in drift:
        shp = np.asarray(a).shape
        v = theta[off:off + int(np.prod(shp, dtype=np.int64))].reshape(shp)
        parts.append(float(v) if shp == () else v.copy())
        off += int(np.prod(shp, dtype=np.int64))
    return type(drift)(*parts)


class _Trainable:
    """The trainable leaves of (params, props), flattened to one unconstrained vector."""

    def __init__(self, params, props):
        self.items = []  # (path, constrainer, shape, slice into the unconstrained vector)
        off = 0
        for path, leaf in _leaves(props):
            if not leaf.trainable:
                continue
            value = np.asarray(_get(params, path), np.float64)
            c = leaf.constrainer
            if c is not None and not (hasattr(c, "forward") and hasattr(c, "inverse") and hasattr(c, "forward_vjp")):
                raise NotImplementedError(
                    f"fit_sgd: constrainer {type(c).__name__} on props.{path} is not supported (use None or "
                    "cd_dynamax_amd.bijectors.RealToPSDBijector)")
            n = value.size if c is None else np.asarray(c.inverse(value)).size
            self.items.append((path, c, value.shape, slice(off, off + n)))
            off += n
        if not self.items:
            raise ValueError("fit_sgd: no trainable parameters")
        self.size = off
        self.drift_only = all(p.startswith("dynamics.drift.") and c is None for p, c, _, _ in self.items)

    def to_unconstrained(self, params) -> np.ndarray:
        u = np.zeros(self.size)
        for path, c, _, sl in self.items:
            value = np.asarray(_get(params, path), np.float64)
            u[sl] = value.ravel() if c is None else c.inverse(value)
        return u

    def from_unconstrained(self, params, u: np.ndarray):
        for path, c, shape, sl in self.items:
            value = u[sl].reshape(shape) if c is None else c.forward(u[sl])
            params = _set(params, path, float(value) if np.ndim(value) == 0 else np.array(value))
        return params

    def pull_back(self, grads, u: np.ndarray) -> np.ndarray:
        """Gradient tree (same structure as params) -> gradient w.r.t. the unconstrained vector."""
        g = np.zeros(self.size)
        for path, c, _, sl in self.items:
            leaf = np.asarray(_get(grads, path), np.float64)
            g[sl] = leaf.ravel() if c is None else c.forward_vjp(u[sl], leaf)
        return g


class _ResidentBatch:
    """One minibatch on the device: t [T,B] (or [T] shared), y [T,m,B], and the output buffers."""

    def __init__(self, y: np.ndarray, t: np.ndarray, t_shared: bool, n_theta: int, n_model: int, dtype):
        B, T, m = y.shape
        self.B, self.T, self.n_theta, self.n_model = B, T, n_theta, n_model
        self.t = _ffi.DeviceArray.from_numpy(np.ascontiguousarray(t if t_shared else t.T, dtype=dtype))
        self.y = _ffi.DeviceArray.from_numpy(np.ascontiguousarray(y.transpose(1, 2, 0), dtype=dtype))
        self.ll = _ffi.DeviceArray((B,), dtype)
        self.grad = _ffi.DeviceArray((B, n_theta), dtype)
        self.gmodel = _ffi.DeviceArray((B, n_model), dtype) if n_model else None
        self.status = _ffi.DeviceArray((B,), np.int32)
        # [sum ll | sum d ll/d theta | sum d ll/d model block | B]: the block a data-parallel step all-reduces in ONE collective
        self.sums = _ffi.DeviceArray((2 + n_theta + n_model,), np.float64)
        self.ones = _ffi.DeviceArray.from_numpy(np.ones(B, dtype))  # the minibatch size reaches slot -1 as a device-side sum

    ukf = False  # fit_sgd(filter_hyperparams=UKFHyperParams()): the unscented filter's objective (cdkf_ukf_loglik_grad_*)

    def _launch(self, mdl: _ffi.ModelBlock, opts, suffix: str):
        """Sweeps + device-side sums of this minibatch into ``self.sums`` (asynchronous, default stream)."""
        L = _ffi.lib()
        if self.ukf and self.n_model:
            _ffi.check(getattr(L, f"cdkf_ukf_loglik_grad_all_{suffix}_dev")(
                C.byref(mdl.c), C.byref(opts), self.B, self.T, self.t.ptr, self.y.ptr, self.ll.ptr, self.grad.ptr,
                self.gmodel.ptr, self.status.ptr, None))
        elif self.ukf:
            _ffi.check(getattr(L, f"cdkf_ukf_loglik_grad_{suffix}_dev")(
                C.byref(mdl.c), C.byref(opts), self.B, self.T, self.t.ptr, self.y.ptr, self.ll.ptr, self.grad.ptr,
                self.status.ptr, None))
        elif self.n_model:
            _ffi.check(getattr(L, f"cdkf_ekf_loglik_grad_all_{suffix}_dev")(
                C.byref(mdl.c), C.byref(opts), self.B, self.T, self.t.ptr, self.y.ptr, self.ll.ptr, self.grad.ptr,
                self.gmodel.ptr, self.status.ptr, None))
        else:
            _ffi.check(getattr(L, f"cdkf_ekf_loglik_grad_{suffix}_dev")(
                C.byref(mdl.c), C.byref(opts), self.B, self.T, self.t.ptr, self.y.ptr, self.ll.ptr, self.grad.ptr,
                self.status.ptr, None))
        sums = self.sums.ptr.value
        _ffi.check(getattr(L, f"cdkf_ll_sum_{suffix}_dev")(self.ll.ptr, self.B, C.c_void_p(sums), None))
        _ffi.check(getattr(L, f"cdkf_grad_sum_{suffix}_dev")(self.grad.ptr, self.B, self.n_theta, C.c_void_p(sums + 8), None))
        if self.n_model:
            _ffi.check(getattr(L, f"cdkf_grad_sum_{suffix}_dev")(self.gmodel.ptr, self.B, self.n_model,
                                                                C.c_void_p(sums + 8 * (1 + self.n_theta)), None))
        _ffi.check(getattr(L, f"cdkf_ll_sum_{suffix}_dev")(self.ones.ptr, self.B,
                                                         C.c_void_p(sums + 8 * (1 + self.n_theta + self.n_model)), None))

    def value_and_grad(self, mdl: _ffi.ModelBlock, opts, suffix: str):
        """(sum ll, sum d ll/d theta [n_theta], sum d ll/d model block [n_model]) over the minibatch, reduced on the device."""
        self._launch(mdl, opts, suffix)
        _ffi.check(_ffi.lib().cdkf_synchronize(None))
        out = self.sums.numpy()
        return float(out[0]), out[1:1 + self.n_theta], out[1 + self.n_theta:-1]

    def value_and_grad_allreduced(self, mdl: _ffi.ModelBlock, opts, suffix: str, comm):
        """The data-parallel SGD objective with nothing returning to the host between the sweep and the reduced sums
        (ssm_temissions.py:555-568: ``vmap(...).sum()`` over ALL sequences): sweeps -> cdkf_ll_sum / cdkf_grad_sum -> ONE in-place
        ``ncclAllReduce`` of 2 + n_theta + n_model doubles over RCCL, all on one stream; then a single copy of the reduced block.
        Returns (sum ll, grad theta, grad model, global minibatch size)."""
        self._launch(mdl, opts, suffix)
        return _allreduce_block(self.sums, self.n_theta, comm)

    def free(self):
        for a in (self.t, self.y, self.ll, self.grad, self.gmodel, self.status, self.sums, self.ones):
            if a is not None:
                a.free()


def _allreduce_block(sums: "_ffi.DeviceArray", n_theta: int, comm):
    comm.allreduce_sum_dev(sums.ptr, int(np.prod(sums.shape)), None)
    _ffi.check(_ffi.lib().cdkf_synchronize(None))
    out = sums.numpy()
    return float(out[0]), out[1:1 + n_theta], out[1 + n_theta:-1], int(round(out[-1]))


class _EmptyPiece:
    """A rank's share of a data-parallel step in which it has no sequence: zeros, through the same device collective."""

    def __init__(self, n_theta: int, n_model: int):
        self.B, self.n_theta = 0, n_theta
        self.sums = _ffi.DeviceArray((2 + n_theta + n_model,), np.float64)

    def value_and_grad_allreduced(self, mdl, opts, suffix, comm):
        _ffi.check(_ffi.lib().cdkf_memset(self.sums.ptr, 0, self.sums.nbytes))
        return _allreduce_block(self.sums, self.n_theta, comm)

    def free(self):
        self.sums.free()


def fit_sgd(model, params, props, emissions, t_emissions=None, filter_hyperparams: Optional[Any] = None, inputs=None,
            optimizer=None, batch_size: int = 1, num_epochs: int = 50, shuffle: bool = False,
            return_param_history: bool = False, return_grad_history: bool = False, key=0, dtype=None,
            allreduce=None, comm=None):
    """Minimise ``-(log_prior + sum_n ll_n * scale) / emissions.size`` over the trainable parameters, in the unconstrained
    space (ssm_temissions.py:548-583).  Returns ``(params, losses)`` (+ parameter / gradient histories when requested, one
    entry per epoch as the reference's scan returns them, optimize_utils.py:128-131; gradients are those of the loss
    w.r.t. the unconstrained parameters, packed like ``params`` for unconstrained leaves).

    ``key``: seed of the NumPy generator that permutes the sequences when ``shuffle`` (JAX's PRNG stream is not
    reproduced).  ``allreduce``: optional callable summing a float64 array over data-parallel ranks
    (``distributed.allreduce_sum_array``) -- each rank then passes its own block of sequences (blocks may differ in size or be
    empty) and every rank applies the same update: that of ONE process holding the concatenated data whose step-b minibatch is the
    union of every rank's b-th piece (``np.array_split`` of the rank's block into ceil(N_total / batch_size) pieces; with
    ``batch_size >= N_total`` that is the plain single-process full-batch fit, otherwise the minibatch composition differs from
    ``batch_size`` consecutive slices of the concatenation).
    ``comm``: a ``distributed.Comm`` instead of the callable.  With an RCCL communicator (``Comm(..., device=d)``) a step's reduction
    stays on the device: sweeps -> ``cdkf_ll_sum`` / ``cdkf_grad_sum`` -> one in-place ``cdkf_ll_allreduce`` of 2 + n_theta + n_model
    doubles on the same stream, and only the reduced block is copied back; a host-only ``Comm`` sums the
    copied-back block over the library's TCP rendezvous (CPU tests)."""
    from .models import _grads_tree, _model_block, _opts, _prepare
    hyper = EKFHyperParams() if filter_hyperparams is None else filter_hyperparams
    ukf = isinstance(hyper, UKFHyperParams)
    if not ukf and not isinstance(hyper, EKFHyperParams):
        raise NotImplementedError("fit_sgd: gradients are provided for the EKF and the UKF marginal log-likelihood")
    optimizer = Adam(1e-3) if optimizer is None else optimizer
    tr = _Trainable(params, props)
    opts = _opts(hyper, 1)
    y, t, batched, dtype = _prepare(emissions, t_emissions, hyper, opts, dtype)
    opts.layout = _ffi.LAYOUT_TCN
    suffix = "f32" if dtype == np.float32 else "f64"
    mdl0 = _model_block(params)
    ukf_all = False
    if ukf:  # the forward-sensitivity kernel where it exists and suffices (drift block, register shapes), else the reverse sweeps
        ukf_all = not tr.drift_only or not _ffi.lib().cdkf_ukf_grad_supported(C.byref(mdl0.c), C.byref(opts))
    check = ((_ffi.lib().cdkf_ukf_grad_all_supported if ukf_all else _ffi.lib().cdkf_ukf_grad_supported) if ukf else
             (_ffi.lib().cdkf_grad_supported if tr.drift_only else _ffi.lib().cdkf_grad_all_supported))
    if not check(C.byref(mdl0.c), C.byref(opts)):
        what = ("unscented-filter drift gradient" if ukf else "drift gradient") if tr.drift_only else "all-parameter reverse-sweep"
        raise NotImplementedError(
            f"fit_sgd: no {what} kernel for drift {type(params.dynamics.drift).__name__} with state_dim={mdl0.state_dim}, "
            f"emission_dim={mdl0.emission_dim}, state_order={getattr(hyper, 'state_order', '-')}"
            + ("" if tr.drift_only else f"; trainable: {[p for p, _, _, _ in tr.items]}"))
    N = y.shape[0]
    size = float(y.size)
    n_theta = mdl0.theta.size
    n_model = 0 if (tr.drift_only and not ukf_all) else _ffi.model_grad_size(mdl0.state_dim, mdl0.emission_dim)
    t_shared = bool(opts.t_shared)
    if comm is not None:
        if allreduce is not None:
            raise ValueError("fit_sgd: pass either allreduce= (a host callable) or comm= (a distributed.Comm), not both")
        allreduce = comm.allreduce_sum_host
    on_device = comm is not None and bool(getattr(comm, "_comm", None))  # the step's reduction runs over RCCL, in place
    if allreduce is not None:
        # Data-parallel: this rank holds its block of the N_total sequences.  The loss is the single-process one,
        # -(sum over the GLOBAL minibatch of ll * N_total / B_global) / emissions.size of the whole data set, so N and size are
        # summed once here, every rank takes the same number of optimiser steps per epoch (ceil(N_total / batch_size): the
        # collective is called the same number of times everywhere), a rank's share of a step is the b-th of that many nearly equal
        # pieces of its block (possibly empty: it then contributes zeros), and the minibatch size travels with the sums.
        tot = np.asarray(allreduce(np.array([float(N), size])), np.float64)
        N_total, size = int(round(tot[0])), float(tot[1])
        num_batches = -(-N_total // batch_size)
        if batch_size >= N_total:
            shuffle = False
        pieces = lambda idx: np.array_split(idx, num_batches)
    else:
        N_total = N
        num_batches = -(-N // batch_size)
        if batch_size >= N:
            shuffle = False
        pieces = lambda idx: [idx[b * batch_size:(b + 1) * batch_size] for b in range(num_batches)]
    rng = np.random.default_rng(key if isinstance(key, (int, np.integer)) else 0)

    def build(idx):
        if len(idx) == 0:  # this rank has no sequence in that step
            return _EmptyPiece(n_theta, n_model) if on_device else None
        rb = _ResidentBatch(y[idx], t if t_shared else t[idx], t_shared, n_theta, n_model, dtype)
        if ukf:
            rb.ukf = True
        return rb

    order = np.arange(N)
    resident: List[Optional[_ResidentBatch]] = []
    if not shuffle:
        resident = [build(idx) for idx in pieces(order)]

    u = tr.to_unconstrained(params)
    state = optimizer.init(u)
    losses, param_hist, grad_hist = [], [], []
    cur = params
    try:
        for _ in range(num_epochs):
            if shuffle:
                perm = rng.permutation(N)
                for b in resident:
                    if b is not None:
                        b.free()
                resident = [build(idx) for idx in pieces(perm)]
            avg, itr = 0.0, 0
            g_u = np.zeros(tr.size)
            for batch in resident:
                cur = tr.from_unconstrained(params, u)
                mdl = _model_block(cur)
                if on_device:
                    ll_sum, g_th, g_md, B = batch.value_and_grad_allreduced(mdl, opts, suffix, comm)
                    if B == 0:
                        continue
                elif batch is None:
                    ll_sum, g_th, g_md, B = 0.0, np.zeros(n_theta), np.zeros(n_model), 0
                else:
                    ll_sum, g_th, g_md = batch.value_and_grad(mdl, opts, suffix)
                    B = batch.B
                if allreduce is not None and not on_device:
                    red = np.asarray(allreduce(np.concatenate([[ll_sum, float(B)], g_th, g_md])), np.float64)
                    ll_sum, B, g_th, g_md = float(red[0]), int(round(red[1])), red[2:2 + n_theta], red[2 + n_theta:]
                    if B == 0:  # no rank had a sequence left for this step (more steps than sequences per rank): nothing to do
                        continue
                grads = _grads_tree(cur, mdl, g_th, g_md if n_model else None)
                scale = N_total / B
                loss = -(ll_sum * scale) / size
                g_u = -(tr.pull_back(grads, u) * scale) / size
                upd, state = optimizer.update(g_u, state)
                u = u + upd
                avg = (avg * itr + loss) / (itr + 1)
                itr += 1
            losses.append(avg)
            if return_param_history:
                param_hist.append(tr.from_unconstrained(params, u))
            if return_grad_history:
                grad_hist.append(g_u.copy())
    finally:
        for b in resident:
            if b is not None:
                b.free()
    out = [tr.from_unconstrained(params, u), np.asarray(losses)]
    if return_param_history:
        out.append(param_hist)
    if return_grad_history:
        out.append(grad_hist)
    return tuple(out)
