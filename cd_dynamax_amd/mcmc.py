"""fit_mcmc: posterior sampling of the trainable parameters with HMC / NUTS, the second caller of the marginal
log-likelihood and its gradient (/root/reference/src/ssm_temissions.py:601-777).  The reference builds
``_logprob(unc) = log_prior + sum_n marginal_log_prob_n + log_det_jac_constrain`` (:656-679), hands it to
``blackjax.window_adaptation(blackjax.<type>, ...)`` (:683-706) and scans the adapted kernel (:713-727).  Here the value and
gradient of that density come from ONE launch of the HIP sweep per leapfrog step on the resident batch (cdkf_ekf_loglik_grad*,
like fit_sgd), and the sampler around it is plain host code:

* ``hmc``  -- velocity-Verlet, ``num_integration_steps`` leapfrogs, Metropolis test (divergence threshold 1000);
* ``nuts`` -- multinomial NUTS, tree doubling (<= 10), generalised U-turn test on ``rho - (p_left + p_right) / 2`` as in
  blackjax 0.9.6 (the reference's pin, hduq_cd_dynamax_requirements.txt:10), progressive sampling (uniform inside a
  subtree, biased at the top);
* Stan's window adaptation as blackjax schedules it: dual averaging of the step size towards an acceptance of 0.8
  (t0 = 10, gamma = 0.05, kappa = 0.75, restarted at every slow-window end), diagonal mass matrix from Welford
  variances over doubling windows, shrunk by ``n / (n + 5)`` towards 1e-3; fewer than 20 warm-up steps adapt the step size only.

Randomness is a NumPy ``Generator`` seeded with ``key``: JAX's PRNG stream is not reproduced, so draws differ from the
reference's sample by sample; parity is distributional (tests/test_mcmc.py: Gaussian targets recover mean and covariance, the
linear-model posterior agrees with the exact Kalman-filter likelihood).
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Callable, Optional, Tuple

import numpy as np

from . import _ffi
from .fit import _Trainable, _ResidentBatch, _get
from .params import EKFHyperParams, UKFHyperParams

LogDensity = Callable[[np.ndarray], Tuple[float, np.ndarray]]


# ---- one Hamiltonian trajectory -----------------------------------------------------------------------------------------
class _Hamiltonian:
    """-log density + p' M^-1 p / 2 with a diagonal inverse mass matrix; counts density evaluations."""

    def __init__(self, logdensity: LogDensity, inv_mass: np.ndarray, step_size: float, divergence_threshold: float = 1000.0):
        self.f, self.inv_mass, self.eps, self.thr = logdensity, inv_mass, step_size, divergence_threshold
        self.evals = 0

    def value_and_grad(self, q):
        self.evals += 1
        lp, g = self.f(q)
        if not np.isfinite(lp) or not np.all(np.isfinite(g)):
            return -np.inf, np.zeros_like(q)
        return float(lp), np.asarray(g, np.float64)

    def momentum(self, rng, n):
        return rng.standard_normal(n) / np.sqrt(self.inv_mass)

    def energy(self, lp, p):
        return -lp + 0.5 * float(np.dot(p * self.inv_mass, p))

    def leapfrog(self, q, p, g, eps):
        p = p + 0.5 * eps * g
        q = q + eps * self.inv_mass * p
        lp, g = self.value_and_grad(q)
        p = p + 0.5 * eps * g
        return q, p, g, lp


def hmc_step(rng, h: _Hamiltonian, q, lp, g, num_integration_steps: int):
    """One HMC transition; returns (q, lp, g, acceptance probability, diverged)."""
    p0 = h.momentum(rng, q.size)
    e0 = h.energy(lp, p0)
    q1, p1, g1, lp1 = q, p0, g, lp
    for _ in range(num_integration_steps):
        q1, p1, g1, lp1 = h.leapfrog(q1, p1, g1, h.eps)
    de = h.energy(lp1, p1) - e0
    de = np.inf if np.isnan(de) else de
    p_acc = float(min(1.0, np.exp(-de)))
    if rng.uniform() < p_acc:
        return q1, lp1, g1, p_acc, de > h.thr
    return q, lp, g, p_acc, de > h.thr


class _Tree:
    __slots__ = ("left", "right", "prop", "logw", "rho", "turning", "diverging", "sum_acc", "n")


def _is_turning(inv_mass, p_left, p_right, rho):
    r = rho - 0.5 * (p_left + p_right)
    return np.dot(inv_mass * p_left, r) <= 0 or np.dot(inv_mass * p_right, r) <= 0


def _leaf(h, e0, q, p, g, lp, direction):
    q1, p1, g1, lp1 = h.leapfrog(q, p, g, direction * h.eps)
    de = h.energy(lp1, p1) - e0
    de = np.inf if np.isnan(de) else de
    t = _Tree()
    t.left = t.right = (q1, p1, g1, lp1)
    t.prop = (q1, lp1, g1)
    t.logw = -de
    t.rho = p1.copy()
    t.turning = False
    t.diverging = bool(de > h.thr)
    t.sum_acc = float(min(1.0, np.exp(-de)))
    t.n = 1
    return t


def _subtree(rng, h, e0, start, direction, depth):
    """2**depth leapfrogs from ``start`` in ``direction``; stops early at a divergence or an internal U-turn."""
    if depth == 0:
        return _leaf(h, e0, *start, direction)
    a = _subtree(rng, h, e0, start, direction, depth - 1)
    if a.turning or a.diverging:
        return a
    b = _subtree(rng, h, e0, a.right if direction > 0 else a.left, direction, depth - 1)
    t = _Tree()
    t.left, t.right = (a.left, b.right) if direction > 0 else (b.left, a.right)
    t.logw = float(np.logaddexp(a.logw, b.logw))
    t.rho = a.rho + b.rho
    t.sum_acc, t.n = a.sum_acc + b.sum_acc, a.n + b.n
    t.diverging = b.diverging
    t.prop = a.prop
    if not (b.turning or b.diverging) and np.log(rng.uniform()) < b.logw - t.logw:  # uniform over the merged leaves
        t.prop = b.prop
    t.turning = b.turning or _is_turning(h.inv_mass, t.left[1], t.right[1], t.rho)
    return t


def nuts_step(rng, h: _Hamiltonian, q, lp, g, max_num_doublings: int = 10):
    """One NUTS transition; returns (q, lp, g, mean acceptance probability over the tree, diverged)."""
    p0 = h.momentum(rng, q.size)
    e0 = h.energy(lp, p0)
    left = right = (q, p0, g, lp)
    prop, logw, rho = (q, lp, g), 0.0, p0.copy()
    sum_acc, n, diverged = 0.0, 0, False
    for depth in range(max_num_doublings):
        direction = 1 if rng.uniform() < 0.5 else -1
        sub = _subtree(rng, h, e0, right if direction > 0 else left, direction, depth)
        sum_acc, n = sum_acc + sub.sum_acc, n + sub.n
        if sub.diverging or sub.turning:
            diverged = sub.diverging
            break
        if np.log(rng.uniform()) < sub.logw - logw:  # biased progressive sampling: favour the new half
            prop = sub.prop
        logw = float(np.logaddexp(logw, sub.logw))
        rho = rho + sub.rho
        if direction > 0:
            right = sub.right
        else:
            left = sub.left
        if _is_turning(h.inv_mass, left[1], right[1], rho):
            break
    return prop[0], prop[1], prop[2], sum_acc / max(n, 1), diverged


# ---- window adaptation ---------------------------------------------------------------------------------------------------
def adaptation_schedule(num_steps: int, initial_buffer_size: int = 75, final_buffer_size: int = 50, first_window_size: int = 25):
    """[(slow stage?, slow window ends here?)] per warm-up step: a fast buffer, doubling slow windows, a fast buffer."""
    if num_steps < 20:
        return [(False, False)] * num_steps
    if initial_buffer_size + first_window_size + final_buffer_size > num_steps:
        initial_buffer_size = int(0.15 * num_steps)
        final_buffer_size = int(0.1 * num_steps)
        first_window_size = num_steps - initial_buffer_size - final_buffer_size
    sched = [(False, False)] * initial_buffer_size
    final_start = num_steps - final_buffer_size
    start, size = initial_buffer_size, first_window_size
    while start < final_start:
        nxt = size
        if 3 * size <= final_start - start:
            nxt = 2 * size
        else:
            size = final_start - start
        sched += [(True, False)] * (size - 1) + [(True, True)]
        start, size = start + size, nxt
    return sched + [(False, False)] * (num_steps - final_start)


class _DualAveraging:
    """Nesterov dual averaging of log(step size) (Hoffman & Gelman 2014, section 3.2.1; blackjax / Stan constants)."""

    def __init__(self, step_size, target=0.8, t0=10, gamma=0.05, kappa=0.75):
        self.target, self.t0, self.gamma, self.kappa = target, t0, gamma, kappa
        self.restart(step_size)

    def restart(self, step_size):
        self.mu = np.log(10.0 * step_size)
        self.log_x, self.log_x_avg, self.step, self.avg_err = np.log(step_size), 0.0, 0, 0.0

    def update(self, p_accept):
        p_accept = 0.0 if np.isnan(p_accept) else p_accept
        self.step += 1
        w = 1.0 / (self.step + self.t0)
        self.avg_err = (1 - w) * self.avg_err + w * (self.target - p_accept)
        self.log_x = self.mu - np.sqrt(self.step) / self.gamma * self.avg_err
        eta = self.step ** (-self.kappa)
        self.log_x_avg = eta * self.log_x + (1 - eta) * self.log_x_avg

    @property
    def current(self):
        return float(np.exp(self.log_x))

    @property
    def final(self):
        return float(np.exp(self.log_x_avg))


class _Welford:
    def __init__(self, n):
        self.n = n
        self.reset()

    def reset(self):
        self.count, self.mean, self.m2 = 0, np.zeros(self.n), np.zeros(self.n)

    def update(self, x):
        self.count += 1
        d = x - self.mean
        self.mean = self.mean + d / self.count
        self.m2 = self.m2 + d * (x - self.mean)

    def regularised_variance(self):
        var = self.m2 / max(self.count - 1, 1)
        return (self.count / (self.count + 5.0)) * var + 1e-3 * (5.0 / (self.count + 5.0))


def find_reasonable_step_size(rng, logdensity, q, lp, g, inv_mass, step_size=1.0, target=0.8, max_iter=100):
    """Double / halve the step size until the one-leapfrog acceptance probability crosses the target."""
    h = _Hamiltonian(logdensity, inv_mass, step_size)
    direction = 0
    for _ in range(max_iter):
        p0 = h.momentum(rng, q.size)
        q1, p1, _, lp1 = h.leapfrog(q, p0, g, h.eps)
        de = h.energy(lp1, p1) - h.energy(lp, p0)
        p_acc = 0.0 if np.isnan(de) else min(1.0, np.exp(-de))
        new_dir = 1 if p_acc > target else -1
        if direction and new_dir != direction:
            break
        direction = new_dir
        h.eps *= 2.0 ** direction
    return h.eps


def window_adaptation(rng, logdensity: LogDensity, q0, algorithm: str, num_steps: int, initial_step_size: float = 1.0,
                      target_acceptance_rate: float = 0.8, progress=None, **parameters):
    """Warm-up: returns (q, lp, g), step size, inverse mass matrix (diagonal), warm-up positions [num_steps, n], their log
    densities, and a dict of statistics."""
    q = np.asarray(q0, np.float64).copy()
    lp, g = logdensity(q)
    if not np.isfinite(lp):
        raise FloatingPointError("fit_mcmc: the log density is not finite at the initial parameters")
    g = np.asarray(g, np.float64)
    n = q.size
    inv_mass = np.ones(n)
    eps = find_reasonable_step_size(rng, logdensity, q, lp, g, inv_mass, initial_step_size, target_acceptance_rate)
    da, wf = _DualAveraging(eps, target_acceptance_rate), _Welford(n)
    pos, lps, acc, ndiv, evals = np.zeros((num_steps, n)), np.zeros(num_steps), [], 0, 0
    for i, (slow, window_end) in enumerate(adaptation_schedule(num_steps)):
        h = _Hamiltonian(logdensity, inv_mass, da.current)
        q, lp, g, p_acc, div = _transition(rng, h, q, lp, g, algorithm, parameters)
        evals += h.evals
        ndiv += int(div)
        acc.append(p_acc)
        da.update(p_acc)
        if slow:
            wf.update(q)
        if window_end:
            inv_mass = wf.regularised_variance()
            wf.reset()
            da.restart(da.final)
        pos[i], lps[i] = q, lp
        if progress:
            progress(i, num_steps, "warm-up")
    step_size = da.final if num_steps else eps
    return (q, lp, g), step_size, inv_mass, pos, lps, {"acceptance": np.asarray(acc), "divergences": ndiv, "evaluations": evals}


def _transition(rng, h, q, lp, g, algorithm, parameters):
    if algorithm == "hmc":
        return hmc_step(rng, h, q, lp, g, int(parameters.get("num_integration_steps", 10)))
    if algorithm == "nuts":
        return nuts_step(rng, h, q, lp, g, int(parameters.get("max_num_doublings", 10)))
    raise NotImplementedError(f"fit_mcmc: mcmc_algorithm type {algorithm!r} (supported: 'hmc', 'nuts')")


def sample(rng, logdensity: LogDensity, state, algorithm: str, step_size: float, inv_mass, num_samples: int, progress=None,
           **parameters):
    """num_samples transitions of the adapted kernel: positions [num_samples, n], log densities, statistics."""
    q, lp, g = state
    pos, lps, acc, ndiv, evals = np.zeros((num_samples, q.size)), np.zeros(num_samples), [], 0, 0
    for i in range(num_samples):
        h = _Hamiltonian(logdensity, inv_mass, step_size)
        q, lp, g, p_acc, div = _transition(rng, h, q, lp, g, algorithm, parameters)
        evals += h.evals
        ndiv += int(div)
        acc.append(p_acc)
        pos[i], lps[i] = q, lp
        if progress:
            progress(i, num_samples, "sampling")
    return pos, lps, {"acceptance": np.asarray(acc), "divergences": ndiv, "evaluations": evals}


# ---- the model-facing entry point ----------------------------------------------------------------------------------------
def _logdet_and_grad(tr: _Trainable, u: np.ndarray):
    """log_det_jac_constrain (dynamax/parameters.py:99-126) of the trainable leaves and its gradient w.r.t. u."""
    total, grad = 0.0, np.zeros(tr.size)
    for _, c, _, sl in tr.items:
        if c is None:
            continue
        if not hasattr(c, "forward_log_det_jacobian_and_grad"):
            raise NotImplementedError(f"fit_mcmc: constrainer {type(c).__name__} has no forward_log_det_jacobian_and_grad")
        v, gv = c.forward_log_det_jacobian_and_grad(u[sl])
        total += float(v)
        grad[sl] = gv
    return total, grad


def fit_mcmc(model, initial_params, props, emissions, t_emissions=None, filter_hyperparams: Optional[Any] = None, inputs=None,
             n_mcmc_samples: int = 500, mcmc_algorithm=None, verbose: bool = True, key=0, dtype=None, return_info: bool = False):
    """``SSM.fit_mcmc`` (ssm_temissions.py:601-777).  Returns ``(warmup_param_samples, mcmc_param_samples,
    warmup_log_probs, mcmc_log_probs)``: parameter sets whose leaves carry a leading sample axis (``num_steps`` warm-up draws,
    ``n_mcmc_samples`` draws; leaves that are not trainable are broadcast, :757-775) and the log densities of the draws in
    the unconstrained space (``-potential_energy``, :749-750).  ``mcmc_algorithm = {"type": "nuts" | "hmc", "parameters":
    {"num_steps": warm-up steps, ["num_integration_steps": ...], ...}}`` as the reference passes it to
    ``blackjax.window_adaptation``.  ``return_info=True`` appends a dict (step size, inverse mass matrix, acceptance rates,
    divergences, number of sweeps)."""
    from .models import _grads_tree, _model_block, _opts, _prepare
    if mcmc_algorithm is None:
        mcmc_algorithm = {"type": "nuts", "parameters": {"num_steps": 4}}
    algo = str(mcmc_algorithm["type"]).lower()
    if algo not in ("hmc", "nuts"):
        raise NotImplementedError(f"fit_mcmc: mcmc_algorithm type {mcmc_algorithm['type']!r} (supported: 'hmc', 'nuts')")
    par = dict(mcmc_algorithm.get("parameters", {}))
    num_steps = int(par.pop("num_steps", 1000))
    hyper = EKFHyperParams() if filter_hyperparams is None else filter_hyperparams
    ukf = isinstance(hyper, UKFHyperParams)
    if not ukf and not isinstance(hyper, EKFHyperParams):
        raise NotImplementedError("fit_mcmc: gradients are provided for the EKF and the UKF marginal log-likelihood")
    prior0 = float(model.log_prior(initial_params)) if hasattr(model, "log_prior") else 0.0
    if prior0 != 0.0:
        raise NotImplementedError("fit_mcmc: a model log_prior other than the reference's 0.0 has no gradient here")
    tr = _Trainable(initial_params, props)
    opts = _opts(hyper, 1)
    y, t, _, dtype = _prepare(emissions, t_emissions, hyper, opts, dtype)
    opts.layout = _ffi.LAYOUT_TCN
    suffix = "f32" if dtype == np.float32 else "f64"
    mdl0 = _model_block(initial_params)
    # (unscented: the reverse sweeps over the closed-form moment equations, every leaf -- cdkf_ukf_loglik_grad_all_*)
    check = (_ffi.lib().cdkf_ukf_grad_all_supported if ukf else
             (_ffi.lib().cdkf_grad_supported if tr.drift_only else _ffi.lib().cdkf_grad_all_supported))
    if not check(C.byref(mdl0.c), C.byref(opts)):
        raise NotImplementedError(
            f"fit_mcmc: no gradient kernel for drift {type(initial_params.dynamics.drift).__name__} with state_dim="
            f"{mdl0.state_dim}, emission_dim={mdl0.emission_dim}, state_order={getattr(hyper, 'state_order', '-')}; trainable: "
            f"{[p for p, _, _, _ in tr.items]}")
    n_theta = mdl0.theta.size
    n_model = 0 if (tr.drift_only and not ukf) else _ffi.model_grad_size(mdl0.state_dim, mdl0.emission_dim)
    batch = _ResidentBatch(y, t, bool(opts.t_shared), n_theta, n_model, dtype)
    batch.ukf = ukf

    def logdensity(u):
        try:
            cur = tr.from_unconstrained(initial_params, u)
            mdl = _model_block(cur)
        except (np.linalg.LinAlgError, FloatingPointError, OverflowError):
            return -np.inf, np.zeros_like(u)
        ll_sum, g_th, g_md = batch.value_and_grad(mdl, opts, suffix)
        ld, g_ld = _logdet_and_grad(tr, u)
        grads = _grads_tree(cur, mdl, g_th, g_md if n_model else None)
        return ll_sum + ld, tr.pull_back(grads, u) + g_ld

    def progress(i, n, what):
        if verbose and (i + 1 == n or (i + 1) % max(n // 10, 1) == 0):
            print(f"fit_mcmc {what}: {i + 1}/{n}", flush=True)

    rng = np.random.default_rng(key if isinstance(key, (int, np.integer)) else 0)
    try:
        with np.errstate(over="ignore", invalid="ignore"):
            state, eps, inv_mass, wpos, wlps, winfo = window_adaptation(rng, logdensity, tr.to_unconstrained(initial_params),
                                                                        algo, num_steps, progress=progress, **par)
            pos, lps, sinfo = sample(rng, logdensity, state, algo, eps, inv_mass, n_mcmc_samples, progress=progress, **par)
    finally:
        batch.free()

    def stack(positions):
        out = initial_params
        drawn = [tr.from_unconstrained(initial_params, u) for u in positions]
        count = len(positions)
        from .fit import _leaves, _set
        for path, leaf in _leaves(props):
            if leaf.trainable:
                value = np.stack([np.asarray(_get(d, path), np.float64) for d in drawn]) if count else np.zeros(
                    (0,) + np.shape(_get(initial_params, path)))
            else:
                base = np.asarray(_get(initial_params, path))
                value = np.broadcast_to(base, (count,) + base.shape).copy()
            out = _set(out, path, value)
        return out

    result = (stack(wpos), stack(pos), wlps, lps)
    if return_info:
        result += ({"step_size": eps, "inverse_mass_matrix": inv_mass, "warmup": winfo, "sampling": sinfo},)
    return result
