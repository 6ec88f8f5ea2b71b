"""CPU oracle (NumPy) for the cd_dynamax CDNLGSSM EKF / UKF / EKF-smoother hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``cd_dynamax_amd/`` may import this module; it is
used by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
as the checker the HIP kernels are compared with.

This is a restatement, batched over a leading trajectory axis N, of the reference algorithm
(all paths relative to /root/reference):

* ``src/continuous_discrete_nonlinear_gaussian_ssm/inference_ekf.py:46-148``  (_predict)
* ``.../inference_ekf.py:153-199``  (_condition_on), ``:202-326`` (extended_kalman_filter)
* ``.../inference_ekf.py:363-448``  (_smooth), ``:450-539`` (extended_kalman_smoother)
* ``.../inference_ukf.py:45-89, 93-159, 162-203, 206-308``  (UKF)
* ``src/utils/diffrax_utils.py:40-165``  (diffeqsolve wrapper: Dopri5, ConstantStepSize, dt0=0.01)
* ``dynamax/utils/utils.py:202-211``  (psd_solve, symmetrize)
* ``.../cdnlgssm_utils.py:38-83``  (LearnableLinear / LearnableLorenz63 drifts)

Third-party arithmetic that is NOT under /root/reference and is restated from its published
algorithm (pinned versions from ``hduq_cd_dynamax_requirements.txt``):

* diffrax 0.4.0 -- ``Dopri5`` Butcher tableau (Dormand & Prince 1980), ``ConstantStepSize``
  (next step = previous end + dt0), the integration loop's ``tnext = min(t0 + dt0, t1)`` start,
  ``tprev = min(tprev, t1)`` and ``_clip_to_end`` (clip when ``tnext > t1 - tol``, tol = 1e-10 in
  float64 / 1e-6 in float32), loop condition ``tprev < t1``.
* jax 0.4.13 -- ``jnp.linalg.cholesky`` / ``cho_factor`` (lower, NaN on a non-positive pivot),
  ``jnp.trace`` on a rank-3 array (traces axes 0 and 1).
* tensorflow-probability 0.20.1 -- ``MultivariateNormalFullCovariance.log_prob``:
  ``L = chol(S)`` (no jitter), ``-0.5*|L^-1 (y-mu)|^2 - sum(log diag L) - 0.5*m*log(2*pi)``.

PARITY PINNING.  The JAX reference cannot be imported in the build container (no jax / diffrax /
tfp).  The oracle is pinned by (tests/test_oracle.py):
  1. the reference's own known-answer constants for the Dopri5 push-forward,
     ``src/test_scripts/cdlgssm_test_filter_TRegular.py:59-60`` (reproduced in float32);
  2. the reference's test equalities EKF(first, second) == UKF == CD Kalman filter on linear
     models (``src/test_scripts/cdnlgssm_test_filter_linear_TRegular.py:314-324, 414-424``), checked
     here against an independent closed-form (matrix-exponential / Van Loan) Kalman filter;
  3. EKF smoother == closed-form RTS smoother on linear models
     (``src/test_scripts/cdnlgssm_test_smoother_linear_TRegular.py:222-232``, soft in the reference).
  4. the one reference-recorded statement about the adaptive path: in the tutorial
     ``src/notebooks/tutorial/diffeqsolve_settings_analysis.ipynb:385-386`` the marginal log-likelihood under the default
     settings (Dopri5, constant dt0 = 0.01) and under ``Tsit5 + PIDController(atol=1e-9, rtol=1e-9)`` print the SAME float32
     value (-14591.8759765625) for a Lorenz-63 model observed through H = [1, 0, 0] at mean gap 0.005 -- the two solves agree
     to float32 resolution (< 1 ulp = 6.7e-8 relative).  tests/test_oracle.py and tests/test_gpu_parity.py hold this oracle and
     the HIP path to that bound on seeded synthetic data of the same model and time density.
Nonlinear (Lorenz / MLP) outputs are NOT pinned by anything else the reference ships ("parity unpinned"
for those beyond the items above); they are pinned to this fp64 restatement.

Per restated third-party routine (none of them is in the mount; each is restated from memory of its published algorithm), what
holds it to the reference:

  routine                                              pinned by
  ---------------------------------------------------  ---------------------------------------------------------------------
  diffrax Dopri5 tableau + stage association           pin 1 bit-exactly in float32 (100 steps of F = -0.1 I); order conditions
  diffrax fixed-step loop (tnext start, tprev clamp)   pin 1 (regular grid only); pins 2, 3 on regular and irregular grids
  diffrax ``_clip_to_end`` tolerance (1e-10 / 1e-6)     NOTHING from the reference (a wrong tolerance changes which side of an
                                                       interval end a last sliver step falls on: <= 1 ulp of state per interval)
  diffrax Tsit5 / Bosh3 / Heun / Midpoint / Ralston    their order conditions and observed order of convergence (tests/test_oracle.py);
    tableaus                                           Tsit5 additionally by pin 4
  diffrax PIDController (error norm, factor formula,   pin 4 only, and only in the regime where the tolerance is met with room to
    safety 0.9, factormin 0.2, factormax 10,           spare: it bounds the RESULT, not the accept / reject sequence.  A wrong
    rejected-step handling, embedded error weights)    recollection that still converges (e.g. another safety factor) passes.
  jax ``cholesky`` NaN semantics, ``trace`` axes        the trace quirk by SURVEY.md section 0.5's derivation; NaN semantics by nothing
  TFP ``MultivariateNormalFullCovariance.log_prob``     pins 2 - 4 (the log-likelihoods of the closed-form Kalman filter / the notebook)
  dynamax ``psd_solve`` / ``symmetrize``                IN the mount (dynamax/utils/utils.py:202-211): restated from source
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np

# --------------------------------------------------------------------------------------
# Dormand-Prince 5(4) tableau (diffrax 0.4.0 ``Dopri5``; only the 5th-order solution is used
# because the step-size controller is ``ConstantStepSize`` -- diffrax_utils.py:47)
# --------------------------------------------------------------------------------------
DOPRI5_A = (
    (),
    (1 / 5,),
    (3 / 40, 9 / 40),
    (44 / 45, -56 / 15, 32 / 9),
    (19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729),
    (9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656),
)
DOPRI5_B = (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84)

# Other explicit Runge-Kutta methods a caller can select through diffeqsolve_settings['solver'] (the reference forwards the
# diffrax solver object to dfx.diffeqsolve, src/utils/diffrax_utils.py:40-57, 150-163).  With ConstantStepSize only the
# solution weights matter (the embedded error estimate is unused); diffrax 0.4.0 is not in the mount, so these are the
# published tableaus: Euler; Heun = explicit trapezoid (diffrax.Heun); explicit midpoint; Ralston (2nd order, minimal error
# bound); Bogacki-Shampine 3(2) (diffrax.Bosh3); Tsitouras 5(4) (diffrax.Tsit5; Tsitouras 2011, Table 1).
TABLEAUS = {
    "dopri5": (DOPRI5_A, DOPRI5_B),
    "euler": (((),), (1.0,)),
    "heun": (((), (1.0,)), (0.5, 0.5)),
    "midpoint": (((), (0.5,)), (0.0, 1.0)),
    "ralston": (((), (2 / 3,)), (0.25, 0.75)),
    "bosh3": (((), (0.5,), (0.0, 0.75)), (2 / 9, 1 / 3, 4 / 9)),
    "tsit5": (((), (0.161,), (-0.008480655492356989, 0.335480655492357),
               (2.8971530571054935, -6.359448489975075, 4.3622954328695815),
               (5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525),
               (5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383)),
              (0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774)),
}
# Embedded error weights b_sol - b_hat (last entry: the first-same-as-last stage f(y_new)) and the order used by the step-size
# controller (diffrax: error_order = solver.order for ODEs), for the methods that carry an error estimate.
ERROR_WEIGHTS = {
    "dopri5": ((35 / 384 - 5179 / 57600, 0.0, 500 / 1113 - 7571 / 16695, 125 / 192 - 393 / 640, -2187 / 6784 + 92097 / 339200,
                11 / 84 - 187 / 2100, -1 / 40), 5),
    "tsit5": ((-0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995, -0.1447110071732629, 0.5823571654525552,
               -0.45808210592918697, 0.015151515151515152), 5),
    "bosh3": ((2 / 9 - 7 / 24, 1 / 3 - 1 / 4, 4 / 9 - 1 / 3, -1 / 8), 3),
    "heun": ((0.5, -0.5, 0.0), 2),
}
_ACTIVE = [("dopri5", None)]


class use_solver:
    """``with use_solver('tsit5'): ...`` -- every diffeqsolve call inside integrates with that tableau.  ``adaptive``: None
    (fixed steps of dt0, diffrax.ConstantStepSize) or a dict(rtol, atol[, pcoeff=0, icoeff=1, dcoeff=0, dtmin, dtmax, safety=0.9,
    factormin=0.2, factormax=10]) for diffrax.PIDController."""

    def __init__(self, name, adaptive=None):
        if name not in TABLEAUS:
            raise ValueError(f"unknown solver {name!r}")
        if adaptive is not None and name not in ERROR_WEIGHTS:
            raise ValueError(f"solver {name!r} has no embedded error estimate")
        self.name, self.adaptive = name, adaptive

    def __enter__(self):
        _ACTIVE.append((self.name, self.adaptive))

    def __exit__(self, *exc):
        _ACTIVE.pop()


# --------------------------------------------------------------------------------------
# Inputs and time.  The reference evaluates the drift and the emission as f(m, u, t), h(m, u, t) (and their jacfwd) with
# u = inputs[t0_idx] held over the interval [t_k, t_{k+1}] and t the solver's stage time (inference_ekf.py:95, 101-114, 277-286;
# inference_ukf.py:142, 189; cdnlgssm_utils.py:13-61: a LearnableFunction is any callable of (x, u, t)); in the reverse solves of the
# smoother t = t_{k+1} - s (diffrax_utils.py:13-25).  The registry drifts ignore both, as the reference's own do; a CallableDrift /
# callable emission built with ut=True receives them.  They travel in this context instead of through every signature: the filters set
# _CTX["u"] ([N, d_u]) per observation step, the Runge-Kutta steppers set _CTX["t"] ([N]) before every right-hand-side evaluation.
_CTX = {"u": None, "t": None}


def _stage_time(t, dt, c, tmap):
    if t is None:
        return
    ts = t + dt * t.dtype.type(c)
    _CTX["t"] = ts if tmap is None else tmap(ts)


def _ctx_rows(n):
    """(u, t) of the context broadcast to n rows (n = N, or N * (2 d + 1) when the sigma points of a batch are evaluated at once)."""
    u, t = _CTX["u"], _CTX["t"]
    if u is not None and u.shape[0] != n:
        u = np.repeat(u, n // u.shape[0], axis=0)
    if t is not None and np.ndim(t) and t.shape[0] != n:
        t = np.repeat(t, n // t.shape[0], axis=0)
    return u, t


# --------------------------------------------------------------------------------------
# Drift registry.  f: [N,d] -> [N,d]; jac: [N,d] -> [N,d,d] (dF_i/dx_j);
# divgrad: [N,d] -> [N,d], the vector  g_l = sum_i d^2 f_i / (dx_i dx_l)  that the reference's
# "second order" mean term 0.5*jnp.trace(H_t @ P) reduces to (0.5 * P @ g), SURVEY.md section 0.5.
# --------------------------------------------------------------------------------------
class LinearDrift:
    """f(x) = W x + b   (cdnlgssm_utils.py:50-61)."""

    kind = "linear"

    def __init__(self, weights, bias):
        self.W = np.asarray(weights)
        self.b = np.asarray(bias)

    def cast(self, dtype):
        return LinearDrift(self.W.astype(dtype), self.b.astype(dtype))

    def f(self, x):
        return x @ self.W.T + self.b

    def jac(self, x):
        return np.broadcast_to(self.W, x.shape[:-1] + self.W.shape).copy()

    def divgrad(self, x):
        return np.zeros_like(x)

    def theta(self):
        return np.concatenate([self.W.ravel(), self.b.ravel()]).astype(np.float64)


class Lorenz63Drift:
    """cdnlgssm_utils.py:63-83."""

    kind = "lorenz63"

    def __init__(self, sigma=10.0, rho=28.0, beta=8.0 / 3.0, dtype=np.float64):
        self.dtype = np.dtype(dtype).type
        self.sigma, self.rho, self.beta = (self.dtype(sigma), self.dtype(rho), self.dtype(beta))

    def cast(self, dtype):
        return Lorenz63Drift(self.sigma, self.rho, self.beta, dtype=dtype)

    def f(self, x):
        s, r, b = self.sigma, self.rho, self.beta
        return np.stack(
            [s * (x[..., 1] - x[..., 0]), x[..., 0] * (r - x[..., 2]) - x[..., 1], x[..., 0] * x[..., 1] - b * x[..., 2]],
            axis=-1,
        )

    def jac(self, x):
        s, r, b = self.sigma, self.rho, self.beta
        F = np.zeros(x.shape + (3,), dtype=x.dtype)
        F[..., 0, 0] = -s
        F[..., 0, 1] = s
        F[..., 1, 0] = r - x[..., 2]
        F[..., 1, 1] = -1
        F[..., 1, 2] = -x[..., 0]
        F[..., 2, 0] = x[..., 1]
        F[..., 2, 1] = x[..., 0]
        F[..., 2, 2] = -b
        return F

    def divgrad(self, x):
        return np.zeros_like(x)  # dF_ii/dx is constant

    def theta(self):
        return np.array([self.sigma, self.rho, self.beta], dtype=np.float64)


class Lorenz96Drift:
    """f_i = (x_{i+1} - x_{i-2}) x_{i-1} - x_i + F  (build-defined; not in the reference)."""

    kind = "lorenz96"

    def __init__(self, forcing=8.0, dtype=np.float64):
        self.dtype = np.dtype(dtype).type
        self.F = self.dtype(forcing)

    def cast(self, dtype):
        return Lorenz96Drift(self.F, dtype=dtype)

    def f(self, x):
        xp1 = np.roll(x, -1, axis=-1)
        xm1 = np.roll(x, 1, axis=-1)
        xm2 = np.roll(x, 2, axis=-1)
        return (xp1 - xm2) * xm1 - x + self.F

    def jac(self, x):
        d = x.shape[-1]
        J = np.zeros(x.shape + (d,), dtype=x.dtype)
        for i in range(d):  # indices may coincide for tiny d, hence +=
            J[..., i, (i + 1) % d] += x[..., (i - 1) % d]
            J[..., i, (i - 2) % d] -= x[..., (i - 1) % d]
            J[..., i, (i - 1) % d] += x[..., (i + 1) % d] - x[..., (i - 2) % d]
            J[..., i, i] -= 1
        return J

    def divgrad(self, x):
        return np.zeros_like(x)  # needs d >= 4; dF_ii/dx = 0

    def theta(self):
        return np.array([self.F], dtype=np.float64)


class MLPDrift:
    """f(x) = W3 tanh(W2 tanh(W1 x + b1) + b2) + b3 (2 hidden layers; build-defined stand-in for the
    reference's NeuralNetDrift tutorial class, whose notebooks are absent from the mount)."""

    kind = "mlp"

    def __init__(self, W1, b1, W2, b2, W3, b3):
        self.W1, self.b1, self.W2, self.b2, self.W3, self.b3 = (np.asarray(a) for a in (W1, b1, W2, b2, W3, b3))

    def cast(self, dtype):
        return MLPDrift(*(a.astype(dtype) for a in (self.W1, self.b1, self.W2, self.b2, self.W3, self.b3)))

    def _fwd(self, x):
        a1 = np.tanh(x @ self.W1.T + self.b1)
        a2 = np.tanh(a1 @ self.W2.T + self.b2)
        return a1, a2

    def f(self, x):
        _, a2 = self._fwd(x)
        return a2 @ self.W3.T + self.b3

    def jac(self, x):
        a1, a2 = self._fwd(x)
        d1 = 1 - a1 * a1  # [N,h1]
        d2 = 1 - a2 * a2  # [N,h2]
        # J = W3 diag(d2) W2 diag(d1) W1
        B = self.W2[None] * d1[:, None, :]  # [N,h2,h1]
        B = d2[:, :, None] * B
        C = np.einsum("oh,nhk->nok", self.W3, B)  # [N,d,h1]
        return np.einsum("nok,kj->noj", C, self.W1)

    def divgrad(self, x):
        # g_l = d/dx_l  sum_i J_ii(x); analytic: differentiate tr(W3 D2 W2 D1 W1).
        a1, a2 = self._fwd(x)
        d1 = 1 - a1 * a1
        d2 = 1 - a2 * a2
        dd1 = -2 * a1 * d1  # d(d1)/d(z1)
        dd2 = -2 * a2 * d2
        W1, W2, W3 = self.W1, self.W2, self.W3
        # tr(J) = sum_{p,q} (W1 W3)_{q p}... write  tr = sum_{p,q} M_pq d2_p W2_pq d1_q,  M = (W1 @ W3).T -> M_pq = sum_i W3_ip W1_qi
        M = (W1 @ W3).T  # [h2,h1]
        G = M * W2  # [h2,h1]
        # d tr / d z1_q (direct)  = sum_p G_pq d2_p dd1_q
        # d tr / d z2_p = sum_q G_pq dd2_p d1_q ; z2 = W2 a1 + b2 -> dz2_p/dz1_q = W2_pq d1_q
        t_direct = (d2 @ G) * dd1  # [N,h1]
        s2 = dd2 * (d1 @ G.T)  # [N,h2]
        t_chain = (s2 @ W2) * d1  # [N,h1]
        return (t_direct + t_chain) @ W1  # [N,d]

    def theta(self):
        return np.concatenate([a.ravel() for a in (self.W1, self.b1, self.W2, self.b2, self.W3, self.b3)]).astype(np.float64)


class CallableDrift:
    """A drift given by vectorised callables -- the oracle-side twin of a run-time compiled custom drift
    (cdkf_custom_drift_register): f(x, theta) -> [N,d], jac(x, theta) -> [N,d,d], divgrad(x, theta) -> [N,d] (or None
    when grad(div f) = 0 is not claimed: state_order 'second' is then unavailable)."""

    kind = "custom"

    def __init__(self, theta, f, jac, divgrad=None, dtype=np.float64, vjp=None, gvjp=None, ut=False, dtheta=None):
        """ut=True: the callables are f(x, theta, u, t), jac(x, theta, u, t), ... with u [N, d_u] the interval's inputs and t [N] the
        stage time (module header: _CTX)."""
        self.dtype = np.dtype(dtype)
        self.th = np.asarray(theta, dtype=self.dtype)
        self.ut = ut
        self._f, self._jac, self._g = f, jac, divgrad
        # vjp(x [d], lam [d], G [d,d], theta) -> (xbar [d], thetabar): the gradient of lam . f + <G, F> (drift_vjp) written out by hand --
        # what ekf_loglik_grad_adjoint needs of a drift it has no formulas for; gvjp(x, u [d], theta) -> (xbar, thetabar): the gradient
        # of u . grad(div f) (divgrad_vjp: state_order 'second')
        self._vjp, self._gvjp = vjp, gvjp
        self._dth = dtheta   # dtheta(x, theta[, u, t]) -> [N, P, d]: d f / d theta (ukf_loglik_grad_all_literal's forward mode)

    def cast(self, dtype):
        return CallableDrift(self.th, self._f, self._jac, self._g, dtype=dtype, vjp=self._vjp, gvjp=self._gvjp, ut=self.ut, dtheta=self._dth)

    def _extra(self, x):
        return _ctx_rows(x.shape[0]) if self.ut else ()

    def f(self, x):
        return np.asarray(self._f(x, self.th, *self._extra(x)), dtype=x.dtype)

    def jac(self, x):
        return np.asarray(self._jac(x, self.th, *self._extra(x)), dtype=x.dtype)

    def divgrad(self, x):
        if self._g is None:
            raise NotImplementedError("custom drift without grad(div f): state_order 'second' is unavailable")
        return np.asarray(self._g(x, self.th, *self._extra(x)), dtype=x.dtype)

    def theta(self):
        return self.th.astype(np.float64)


class Model:
    """The pieces of ParamsCDNLGSSM the hot path touches (cdnlgssm_utils.py:88-209): drift, L, Qc,
    linear emission h(x) = H x + bias (LearnableLinear), R, initial mean / covariance."""

    def __init__(self, drift, L, Qc, H, bias, R, m0, P0, emission=None, emission_ut=False):
        """``emission``: optional pair of vectorised callables (h(x, eta) -> [N,m], jac(x, eta) -> [N,m,d]) for a
        non-linear emission function (the reference accepts any callable, cdnlgssm_utils.py:38-61, and linearises it with
        jacfwd, inference_ekf.py:258); its parameter vector eta is concat(H.ravel(), bias) -- the storage a run-time
        compiled custom emission (cdkf_custom_emission_register) reads."""
        self.drift = drift
        self.L, self.Qc, self.H, self.bias, self.R, self.m0, self.P0 = (
            np.asarray(a, dtype=np.float64) for a in (L, Qc, H, bias, R, m0, P0)
        )
        self.d = self.m0.shape[0]
        self.m = self.H.shape[0]
        self.emission = emission
        self.emission_ut = emission_ut   # the pair is h(x, eta, u, t), jac(x, eta, u, t): inputs and observation time (module header: _CTX)

    def cast(self, dtype):
        mdl = Model.__new__(Model)
        mdl.drift = self.drift.cast(dtype)
        for k in ("L", "Qc", "H", "bias", "R", "m0", "P0"):
            setattr(mdl, k, getattr(self, k).astype(dtype))
        mdl.d, mdl.m = self.d, self.m
        mdl.emission = self.emission
        mdl.emission_ut = getattr(self, "emission_ut", False)
        return mdl

    def h(self, x):
        """Emission mean at x [..., d] -> [..., m]."""
        if self.emission is None:
            return x @ self.H.T + self.bias
        eta = np.concatenate([self.H.ravel(), self.bias])
        extra = _ctx_rows(x.shape[0]) if getattr(self, "emission_ut", False) else ()
        return np.asarray(self.emission[0](x, eta, *extra), dtype=x.dtype)

    def Hjac(self, x):
        """Emission Jacobian at x [N, d] -> [N, m, d]."""
        if self.emission is None:
            return np.broadcast_to(self.H, x.shape[:-1] + self.H.shape)
        eta = np.concatenate([self.H.ravel(), self.bias])
        extra = _ctx_rows(x.shape[0]) if getattr(self, "emission_ut", False) else ()
        return np.asarray(self.emission[1](x, eta, *extra), dtype=x.dtype)


# --------------------------------------------------------------------------------------
# small dense linear algebra, batched over the leading axis, NaN-propagating like jax/LAPACK-on-XLA
# --------------------------------------------------------------------------------------
def symmetrize(A):
    """dynamax/utils/utils.py:209-211."""
    return 0.5 * (A + np.swapaxes(A, -1, -2))


def cholesky_lower(A):
    """Lower Cholesky, batched; a non-positive pivot yields NaN (jnp.linalg.cholesky semantics)."""
    A = np.asarray(A)
    n = A.shape[-1]
    Lm = np.zeros_like(A)
    with np.errstate(invalid="ignore", divide="ignore"):
        for j in range(n):
            s = A[..., j, j] - np.sum(Lm[..., j, :j] ** 2, axis=-1)
            piv = np.sqrt(s)  # NaN if s < 0
            Lm[..., j, j] = piv
            for i in range(j + 1, n):
                s = A[..., i, j] - np.sum(Lm[..., i, :j] * Lm[..., j, :j], axis=-1)
                Lm[..., i, j] = s / piv
    return Lm


def solve_lower(Lm, B):
    """Solve L X = B (B: [..., n, k])."""
    n = Lm.shape[-1]
    X = np.zeros_like(B)
    with np.errstate(invalid="ignore", divide="ignore"):
        for i in range(n):
            s = B[..., i, :] - np.einsum("...j,...jk->...k", Lm[..., i, :i], X[..., :i, :])
            X[..., i, :] = s / Lm[..., i, i][..., None]
    return X


def solve_upper_from_lower(Lm, B):
    """Solve L^T X = B."""
    n = Lm.shape[-1]
    X = np.zeros_like(B)
    with np.errstate(invalid="ignore", divide="ignore"):
        for i in range(n - 1, -1, -1):
            s = B[..., i, :] - np.einsum("...j,...jk->...k", Lm[..., i + 1 :, i], X[..., i + 1 :, :])
            X[..., i, :] = s / Lm[..., i, i][..., None]
    return X


def psd_solve(A, B, diagonal_boost=1e-9):
    """dynamax/utils/utils.py:202-207: symmetrize + boost*I, Cholesky (lower), cho_solve."""
    dtype = A.dtype
    n = A.shape[-1]
    A = symmetrize(A) + dtype.type(diagonal_boost) * np.eye(n, dtype=dtype)
    Lm = cholesky_lower(A)
    return solve_upper_from_lower(Lm, solve_lower(Lm, B))


def mvn_logpdf(y, mu, S):
    """TFP MultivariateNormalFullCovariance(mu, S).log_prob(y) (inference_ekf.py:286, inference_ukf.py:197)."""
    dtype = S.dtype
    m = S.shape[-1]
    Lm = cholesky_lower(S)
    z = solve_lower(Lm, (y - mu)[..., None])[..., 0]
    with np.errstate(invalid="ignore", divide="ignore"):
        logdet_half = np.sum(np.log(np.diagonal(Lm, axis1=-2, axis2=-1)), axis=-1)
    return dtype.type(-0.5) * np.sum(z * z, axis=-1) - logdet_half - dtype.type(0.5 * m * math.log(2 * math.pi))


# --------------------------------------------------------------------------------------
# diffeqsolve (diffrax_utils.py:40-165, ODE branch; diffrax 0.4.0 integrate loop)
# --------------------------------------------------------------------------------------
def _tree_axpy(y0, ks, coefs, dtype):
    """y0 + sum_j coefs[j] * ks[j]: the increment is summed first and added to y0 once (diffrax's
    ``y0 + a_lower[i] @ ks`` form), which is what keeps the float32 push-forward within 1-2 ulp of exact."""
    out = []
    for c, comp0 in enumerate(y0):
        acc = None
        for kj, a in zip(ks, coefs):
            if a != 0.0:
                term = dtype.type(a) * kj[c]
                acc = term if acc is None else acc + term
        out.append(comp0 + acc)
    return tuple(out)


def dopri5_step(rhs, y, dt, t=None, tmap=None):
    """One Dopri5 step of size dt (dt: [N]); k_j = dt * f(stage_j) (diffrax ODETerm.vf_prod).  t ([N], optional): the step's start --
    stage i is then evaluated with the context time tmap(t + c_i dt), c_i = sum_j a_ij (module header: _CTX)."""
    dtype = y[0].dtype
    A, B = TABLEAUS[_ACTIVE[-1][0]]
    ks = []
    for i in range(len(B)):
        yi = y if i == 0 else _tree_axpy(y, ks, A[i], dtype)
        _stage_time(t, dt, sum(A[i]), tmap)
        fi = rhs(yi)
        ks.append(tuple(dt.reshape((-1,) + (1,) * (c.ndim - 1)) * c for c in fi))
    return _tree_axpy(y, ks, B, dtype)


def diffeqsolve(rhs, t0, t1, y0, dt0=0.01, max_steps=100000, count_steps=None, err_components=None, dt_log=None, tmap=None):
    """Integrate the autonomous ODE y' = rhs(y) from t0 to t1 (both [N]) with fixed-step Dopri5.

    Mirrors diffrax 0.4.0: tprev=t0, tnext=min(t0+dt0, t1); while tprev < t1: step(tprev->tnext);
    tprev=min(tnext, t1); tnext=clip_to_end(tprev + dt0).  Under vmap the reference runs until every
    lane has finished with finished lanes masked; same here.  ``reverse=True`` solves in the
    reference integrate s from 0 to t1-t0 with rhs negated (diffrax_utils.py:13-25,131-135); for the
    autonomous right-hand sides of this path the caller passes t0=0, t1=t1-t0 and the negated rhs.
    """
    if _ACTIVE[-1][1] is not None:
        return _diffeqsolve_adaptive(rhs, t0, t1, y0, dt0, max_steps, count_steps, err_components, dt_log, tmap)
    dtype = y0[0].dtype
    tol = dtype.type(1e-10 if dtype == np.float64 else 1e-6)
    t0 = np.asarray(t0, dtype=dtype)
    t1 = np.asarray(t1, dtype=dtype)
    dt0 = dtype.type(dt0)
    tprev = t0.copy()
    tnext = np.minimum(t0 + dt0, t1)
    y = tuple(c.copy() for c in y0)
    nsteps = np.zeros(t0.shape, dtype=np.int64)
    for _ in range(int(max_steps)):
        active = tprev < t1
        if not active.any():
            break
        dt = np.where(active, tnext - tprev, dtype.type(0))
        ynew = dopri5_step(rhs, y, dt, tprev, tmap)
        y = tuple(np.where(active.reshape((-1,) + (1,) * (c.ndim - 1)), cn, c) for cn, c in zip(ynew, y))
        nsteps += active
        tprev_new = np.minimum(tnext, t1)
        tnext_new = tnext + dt0
        tnext_new = np.where(tnext_new > t1 - tol, t1, tnext_new)
        tprev = np.where(active, tprev_new, tprev)
        tnext = np.where(active, tnext_new, tnext)
    if count_steps is not None:
        count_steps.append(nsteps)
    return y


def _diffeqsolve_adaptive(rhs, t0, t1, y0, dt0, max_steps, count_steps, err_components=None, dt_log=None, tmap=None):
    """diffrax.PIDController around the embedded pair (diffrax 0.4.0 is not in the mount; restated from its published
    algorithm -- step_size_controller/adaptive.py and the integrate loop):
      y_error      = dt * sum_i (b_sol - b_hat)_i k_i, the last stage being f(y_candidate) (FSAL methods)
      scaled_error = rms over ALL entries of the state pytree (mean and full d x d covariance) of
                     y_error / (atol + max(|y0|, |y_candidate|) rtol);   keep the step iff scaled_error < 1
      factor       = clip(safety * e^-(i+p+d)/order * e_prev^(p+2d)/order * e_prevprev^-d/order, [1 if kept else factormin, factormax])
                     with e the scaled error (history updated on accepted steps only), safety 0.9, factormin 0.2, factormax 10
      next step    = previous attempted size * factor, clipped to [dtmin, dtmax] when those are given (a size proposed at or below
                     dtmin flags the step taken with it as kept whatever its error: force_dtmin), from t1 (kept) or again from t0
                     (rejected); the end is clipped as in the fixed-step loop, a REJECTED step that would cross the end is sent half-way there
    max_steps counts accepted and rejected steps."""
    name, ad = _ACTIVE[-1]
    A, B = TABLEAUS[name]
    Berr, order = ERROR_WEIGHTS[name]
    dtype = y0[0].dtype
    tol = dtype.type(1e-10 if dtype == np.float64 else 1e-6)
    rtol, atol = dtype.type(ad["rtol"]), dtype.type(ad["atol"])
    pc, ic, dc = (dtype.type(ad.get(k, v)) for k, v in (("pcoeff", 0.0), ("icoeff", 1.0), ("dcoeff", 0.0)))
    c1, c2, c3 = (ic + pc + dc) / order, -(pc + 2 * dc) / order, dc / order
    safety, fmin, fmax = (dtype.type(ad.get(k) or v) for k, v in (("safety", 0.9), ("factormin", 0.2), ("factormax", 10.0)))
    # dtmin / dtmax (PIDController.init and the end of adapt_step_size, force_dtmin=True): every proposed size -- the first one too -- is
    # clipped to [dtmin, dtmax]; a step proposed at or below dtmin is flagged, and the step taken under that flag is KEPT whatever its error
    dtmin, dtmax = dtype.type(ad.get("dtmin") or 0.0), dtype.type(ad.get("dtmax") or np.inf)
    t0 = np.asarray(t0, dtype=dtype)
    t1 = np.asarray(t1, dtype=dtype)
    N = t0.shape[0]
    tprev = t0.copy()
    dt_first = np.minimum(dtype.type(dt0), dtmax)
    at_min = np.full(N, bool(dt_first <= dtmin))
    tnext = np.minimum(t0 + np.maximum(dt_first, dtmin), t1)
    y = tuple(c.copy() for c in y0)
    inv1 = np.ones(N, dtype)
    inv2 = np.ones(N, dtype)
    nsteps = np.zeros(N, dtype=np.int64)
    bc = lambda v, c: v.reshape((-1,) + (1,) * (c.ndim - 1))
    # err_components: how many leading components form the reference's state pytree and enter the error norm (the forward-
    # sensitivity gradient integrates tangents alongside; JAX differentiates the solve with the controller under stop_gradient)
    nerr = len(y0) if err_components is None else err_components
    size = sum(int(np.prod(c.shape[1:])) for c in y0[:nerr])
    for _ in range(int(max_steps)):
        active = tprev < t1
        if not active.any():
            break
        dt = np.where(active, tnext - tprev, dtype.type(0))
        ks = []
        for i in range(len(B)):
            yi = y if i == 0 else _tree_axpy(y, ks, A[i], dtype)
            _stage_time(tprev, dt, sum(A[i]), tmap)
            ks.append(tuple(bc(dt, c) * c for c in rhs(yi)))
        ynew = _tree_axpy(y, ks, B, dtype)
        if len(Berr) > len(B) and Berr[len(B)] != 0.0:
            _stage_time(tprev, dt, 1.0, tmap)
            ks.append(tuple(bc(dt, c) * c for c in rhs(ynew)))
        yerr = tuple(sum(dtype.type(Berr[i]) * ks[i][c] for i in range(len(ks)) if Berr[i] != 0.0) for c in range(len(y)))
        sq = np.zeros(N, dtype)
        for c in range(nerr):
            sc = yerr[c] / (atol + np.maximum(np.abs(y[c]), np.abs(ynew[c])) * rtol)
            sq = sq + np.sum((sc * sc).reshape(N, -1), axis=1)
        scaled = np.sqrt(sq / dtype.type(size))
        keep = (scaled < 1) | at_min
        with np.errstate(divide="ignore", over="ignore", invalid="ignore"):
            inv = np.where(scaled == 0, dtype.type(np.inf), dtype.type(1) / scaled)
            factor = safety * inv ** c1
            if c2 != 0:
                factor = factor * inv1 ** c2
            if c3 != 0:
                factor = factor * inv2 ** c3
        # fmax / fmin: a NaN error estimate (e.g. a UKF stage covariance that lost positive definiteness because the attempted
        # step was far too long) rejects the step with factor = factormin.  diffrax's clip would propagate the NaN into the
        # step size and the solve would end in its max_steps error; the engine retries with a shorter step instead.
        factor = np.fmin(np.fmax(factor, np.where(keep, dtype.type(1), fmin)), fmax)
        dtn = np.fmin(dt * factor, dtmax)
        at_min = np.where(active, dtn <= dtmin, at_min)
        dtn = np.fmax(dtn, dtmin)
        nt0 = np.where(keep, tnext, tprev)
        nt1 = nt0 + dtn
        upd = active & keep
        if dt_log is not None and upd[0]:  # (batch of one: the accepted step sizes, for the reverse sweep's oracle)
            dt_log.append(float(dt[0]))
        y = tuple(np.where(bc(upd, c), cn, c) for cn, c in zip(ynew, y))
        inv2 = np.where(upd, inv1, inv2)
        inv1 = np.where(upd, inv, inv1)
        nsteps += active
        tprev_new = np.minimum(nt0, t1)
        tclip = np.where(keep, t1, tprev_new + dtype.type(0.5) * (t1 - tprev_new))
        tnext_new = np.where(nt1 > t1 - tol, tclip, nt1)
        tprev = np.where(active, tprev_new, tprev)
        tnext = np.where(active, tnext_new, tnext)
    if count_steps is not None:
        count_steps.append(nsteps)
    return y


# --------------------------------------------------------------------------------------
# EKF
# --------------------------------------------------------------------------------------
def _LQL(mdl):
    return mdl.L @ mdl.Qc @ mdl.L.T


def _inputs(inputs, N, T, dtype):
    """inputs [N, T, d_u] in the compute type; None -> zeros [N, T, 1] (_process_input, inference_ekf.py:32, 260)."""
    if inputs is None:
        return np.zeros((N, T, 1), dtype)
    u = np.asarray(inputs, dtype=dtype)
    return np.broadcast_to(u, (N,) + u.shape).copy() if u.ndim == 2 else u


def _set_step(u, t0s, k):
    """Context of observation step k: u = inputs[k] for the emission and the whole interval that follows, t = t_k for the emission
    (inference_ekf.py:277-286: h(pred_mean, u, t0), H(pred_mean, u, t0))."""
    _CTX["u"] = u[:, k]
    _CTX["t"] = t0s[:, k]


def ekf_predict(mdl, m, P, t0, t1, state_order="second", dt0=0.01, max_steps=100000, cov_rescaling=1.0):
    """inference_ekf.py:46-148."""
    dtype = m.dtype
    LQL = _LQL(mdl)
    drift = mdl.drift

    if state_order == "zeroth":
        (m1,) = diffeqsolve(lambda y: (drift.f(y[0]),), t0, t1, (m,), dt0, max_steps)
        dt = (np.asarray(t1, dtype=dtype) - np.asarray(t0, dtype=dtype))[:, None, None]
        Lr = mdl.L * dtype.type(cov_rescaling)
        return m1, P + np.sqrt(dt) * (Lr @ mdl.Qc @ Lr.T)

    def rhs(y):
        mm, PP = y
        F = drift.jac(mm)
        dm = drift.f(mm)
        if state_order == "second":
            # 0.5*jnp.trace(H_t @ P) with H_t[i,j,k] -> 0.5 * sum_k g_k P[k,:]   (inference_ekf.py:111-114)
            dm = dm + dtype.type(0.5) * np.einsum("nk,nkl->nl", drift.divgrad(mm), PP)
        elif state_order != "first":
            raise ValueError(f"EKF hyperparams.state_order = {state_order} not implemented yet")
        dP = F @ PP + PP @ np.swapaxes(F, -1, -2) + LQL
        return dm, dP

    return diffeqsolve(rhs, t0, t1, (m, P), dt0, max_steps)


def ekf_condition_on(mdl, m, P, y, num_iter=1):
    """inference_ekf.py:153-199: the emission is re-linearised at the current mean in every iteration."""
    R = mdl.R
    for _ in range(num_iter):
        H = mdl.Hjac(m)
        S = R + H @ P @ np.swapaxes(H, -1, -2)
        K = np.swapaxes(psd_solve(S, H @ P), -1, -2)
        Pn = P - K @ S @ np.swapaxes(K, -1, -2)
        mn = m + np.einsum("nij,nj->ni", K, y - mdl.h(m))
        m, P = mn, Pn
    return m, symmetrize(P)


def _t0_t1(t, dt_final, dtype):
    """inference_ekf.py:235-250: t1 = [t[1:], t[-1] + dt_final]."""
    t = np.asarray(t, dtype=dtype)
    t1 = np.concatenate([t[:, 1:], t[:, -1:] + dtype.type(dt_final)], axis=1)
    return t, t1


def ekf_filter(
    mdl: Model,
    t,
    y,
    state_order: str = "second",
    num_iter: int = 1,
    dt0: float = 0.01,
    dt_final: float = 1e-10,
    max_steps: int = 100000,
    cov_rescaling: float = 1.0,
    dtype=np.float64,
    inputs=None,
):
    """extended_kalman_filter (inference_ekf.py:202-326), batched: t [N,T], y [N,T,m], inputs [N,T,d_u] or None."""
    dtype = np.dtype(dtype)
    mdl = mdl.cast(dtype)
    y = np.asarray(y, dtype=dtype)
    N, T, _ = y.shape
    d = mdl.d
    t0s, t1s = _t0_t1(t, dt_final, dtype)
    u = _inputs(inputs, N, T, dtype)
    ll = np.zeros(N, dtype=dtype)
    pm = np.broadcast_to(mdl.m0, (N, d)).copy()
    pP = np.broadcast_to(mdl.P0, (N, d, d)).copy()
    out = {
        "filtered_means": np.zeros((N, T, d), dtype),
        "filtered_covariances": np.zeros((N, T, d, d), dtype),
        "predicted_means": np.zeros((N, T, d), dtype),
        "predicted_covariances": np.zeros((N, T, d, d), dtype),
    }
    for k in range(T):
        yk = y[:, k]
        _set_step(u, t0s, k)
        Hk = mdl.Hjac(pm)
        S = Hk @ pP @ np.swapaxes(Hk, -1, -2) + mdl.R
        ll = ll + mvn_logpdf(yk, mdl.h(pm), S)
        fm, fP = ekf_condition_on(mdl, pm, pP, yk, num_iter)
        pm, pP = ekf_predict(mdl, fm, fP, t0s[:, k], t1s[:, k], state_order, dt0, max_steps, cov_rescaling)
        out["filtered_means"][:, k] = fm
        out["filtered_covariances"][:, k] = fP
        out["predicted_means"][:, k] = pm
        out["predicted_covariances"][:, k] = pP
    out["marginal_loglik"] = ll
    return out


def ekf_smoother(mdl: Model, t, y, state_order="second", dt0=0.01, dt_final=1e-10, max_steps=100000, dtype=np.float64,
                 filtered: Optional[dict] = None, inputs=None):
    """extended_kalman_smoother (inference_ekf.py:450-539) + _smooth (:363-448), smooth_order='first'.

    The filter inside the smoother runs with num_iter=1 (inference_ekf.py:489-495).
    """
    dtype = np.dtype(dtype)
    if filtered is None:
        filtered = ekf_filter(mdl, t, y, state_order, 1, dt0, dt_final, max_steps, dtype=dtype, inputs=inputs)
    mdl = mdl.cast(dtype)
    t = np.asarray(t, dtype=dtype)
    fm, fP = filtered["filtered_means"], filtered["filtered_covariances"]
    N, T, d = fm.shape
    u = _inputs(inputs, N, T, dtype)
    LQL = _LQL(mdl)
    sm = np.zeros_like(fm)
    sP = np.zeros_like(fP)
    sm[:, -1] = fm[:, -1]
    sP[:, -1] = fP[:, -1]
    ms, Ps = fm[:, -1].copy(), fP[:, -1].copy()
    drift = mdl.drift
    for k in range(T - 2, -1, -1):
        mf, Pf = fm[:, k], fP[:, k]

        def rhs(yv, mf=mf, Pf=Pf):
            m_s, P_s = yv
            F = drift.jac(mf)
            aux = np.swapaxes(psd_solve(Pf, np.broadcast_to(LQL, Pf.shape).copy()), -1, -2)
            G = F + aux
            dm = drift.f(mf) + np.einsum("nij,nj->ni", G, m_s - mf)
            dP = G @ P_s + P_s @ np.swapaxes(G, -1, -2) - LQL
            return -dm, -dP  # reverse_rhs (diffrax_utils.py:13-25)

        t0, t1 = t[:, k], t[:, k + 1]
        _CTX["u"] = u[:, k]   # u = inputs[t0_idx] of the interval (inference_ekf.py:516); solver time s -> t1 - s (diffrax_utils.py:13-25)
        ms, Ps = diffeqsolve(rhs, np.zeros_like(t0), t1 - t0, (ms, Ps), dt0, max_steps, tmap=lambda sv, t1=t1: t1 - sv)
        sm[:, k] = ms
        sP[:, k] = Ps
    return {
        "marginal_loglik": filtered["marginal_loglik"],
        "filtered_means": fm,
        "filtered_covariances": fP,
        "smoothed_means": sm,
        "smoothed_covariances": sP,
    }


# --------------------------------------------------------------------------------------
# UKF (inference_ukf.py)
# --------------------------------------------------------------------------------------
def ukf_weights(n, alpha, beta, kappa, dtype):
    """_compute_lambda / _compute_weights (inference_ukf.py:42, 63-89)."""
    dtype = np.dtype(dtype)
    alpha = dtype.type(alpha)
    lamb = alpha**2 * dtype.type(n + kappa) - dtype.type(n)
    factor = dtype.type(1) / (dtype.type(2) * (dtype.type(n) + lamb))
    w_mean = np.concatenate([[lamb / (n + lamb)], np.ones(2 * n, dtype) * factor]).astype(dtype)
    w_cov = np.concatenate([[lamb / (n + lamb) + (1 - alpha**2 + beta)], np.ones(2 * n, dtype) * factor]).astype(dtype)
    I_w = np.eye(2 * n + 1, dtype=dtype) - w_mean[:, None]
    W = (I_w @ np.diag(w_cov) @ I_w.T).astype(dtype)
    return dtype.type(lamb), w_mean, w_cov, W


def ukf_sigmas(m, P, lamb):
    """_compute_sigmas (inference_ukf.py:45-60): m, m + c*chol(P)[:,i], m - c*chol(P)[:,i]; returns [N,2n+1,n]."""
    n = m.shape[-1]
    dist = np.sqrt(m.dtype.type(n) + lamb) * cholesky_lower(P)
    cols = np.swapaxes(dist, -1, -2)  # cols[:, i, :] = dist[:, :, i]
    return np.concatenate([m[:, None, :], m[:, None, :] + cols, m[:, None, :] - cols], axis=1)


def ukf_filter(mdl: Model, t, y, alpha=math.sqrt(3), beta=2, kappa=1, dt0=0.01, dt_final=1e-10, max_steps=100000,
               dtype=np.float64, inputs=None):
    """unscented_kalman_filter (inference_ukf.py:206-308), batched."""
    dtype = np.dtype(dtype)
    mdl = mdl.cast(dtype)
    y = np.asarray(y, dtype=dtype)
    N, T, _ = y.shape
    u = _inputs(inputs, N, T, dtype)
    d = mdl.d
    lamb, w_mean, w_cov, W = ukf_weights(d, alpha, beta, kappa, dtype)
    LQL = _LQL(mdl)
    drift = mdl.drift
    t0s, t1s = _t0_t1(t, dt_final, dtype)
    ll = np.zeros(N, dtype=dtype)
    pm = np.broadcast_to(mdl.m0, (N, d)).copy()
    pP = np.broadcast_to(mdl.P0, (N, d, d)).copy()
    out = {
        "filtered_means": np.zeros((N, T, d), dtype),
        "filtered_covariances": np.zeros((N, T, d, d), dtype),
        "predicted_means": np.zeros((N, T, d), dtype),
        "predicted_covariances": np.zeros((N, T, d, d), dtype),
    }

    def rhs(yv):
        m_t, P_t = yv
        X = ukf_sigmas(m_t, P_t, lamb)  # [N,2n+1,n]
        fX = drift.f(X.reshape(-1, d)).reshape(X.shape)
        dm = np.einsum("nsi,s->ni", fX, w_mean)
        foo = np.einsum("nsi,st,ntj->nij", fX, W, X)
        return dm, foo + np.swapaxes(foo, -1, -2) + LQL

    for k in range(T):
        yk = y[:, k]
        _set_step(u, t0s, k)
        # _condition_on (inference_ukf.py:162-203)
        X = ukf_sigmas(pm, pP, lamb)
        Y = mdl.h(X.reshape(-1, d)).reshape(X.shape[:2] + (mdl.m,))
        ymean = np.einsum("s,nsj->nj", w_mean, Y)
        dY = Y - ymean[:, None, :]
        dX = X - pm[:, None, :]
        S = np.einsum("s,nsi,nsj->nij", w_cov, dY, dY) + mdl.R
        C = np.einsum("s,nsi,nsj->nij", w_cov, dX, dY)
        ll = ll + mvn_logpdf(yk, ymean, S)
        K = np.swapaxes(psd_solve(S, np.swapaxes(C, -1, -2)), -1, -2)
        fm = pm + np.einsum("nij,nj->ni", K, yk - ymean)
        fP = pP - K @ S @ np.swapaxes(K, -1, -2)  # NB: no symmetrize in the UKF
        pm, pP = diffeqsolve(rhs, t0s[:, k], t1s[:, k], (fm, fP), dt0, max_steps)
        out["filtered_means"][:, k] = fm
        out["filtered_covariances"][:, k] = fP
        out["predicted_means"][:, k] = pm
        out["predicted_covariances"][:, k] = pP
    out["marginal_loglik"] = ll
    return out


def forecast(mdl: Model, m_init, P_init, t_init, t_forecast, method="ekf", state_order="second", alpha=math.sqrt(3), beta=2,
             kappa=1, dt0=0.01, max_steps=100000, dtype=np.float64):
    """forecast_extended_kalman_filter (inference_ekf.py:679-766) / forecast_unscented_kalman_filter
    (inference_ukf.py:409-505): repeated _predict over t0 = [t_init, t_forecast[:-1]], t1 = t_forecast, no updates.
    t_init [N], t_forecast [N,n]; returns (means [N,n,d], covs [N,n,d,d])."""
    dtype = np.dtype(dtype)
    mdl = mdl.cast(dtype)
    tf = np.asarray(t_forecast, dtype=dtype)
    N, n = tf.shape
    d = mdl.d
    m = np.broadcast_to(np.asarray(m_init, dtype), (N, d)).copy()
    P = np.broadcast_to(np.asarray(P_init, dtype), (N, d, d)).copy()
    t0 = np.asarray(t_init, dtype=dtype).reshape(N).copy()
    means, covs = np.zeros((N, n, d), dtype), np.zeros((N, n, d, d), dtype)
    if method == "ukf":
        lamb, w_mean, w_cov, W = ukf_weights(d, alpha, beta, kappa, dtype)
        LQL = _LQL(mdl)

        def rhs(yv):
            X = ukf_sigmas(yv[0], yv[1], lamb)
            fX = mdl.drift.f(X.reshape(-1, d)).reshape(X.shape)
            foo = np.einsum("nsi,st,ntj->nij", fX, W, X)
            return np.einsum("nsi,s->ni", fX, w_mean), foo + np.swapaxes(foo, -1, -2) + LQL
    for k in range(n):
        if method == "ukf":
            m, P = diffeqsolve(rhs, t0, tf[:, k], (m, P), dt0, max_steps)
        else:
            m, P = ekf_predict(mdl, m, P, t0, tf[:, k], state_order, dt0, max_steps)
        means[:, k], covs[:, k] = m, P
        t0 = tf[:, k]
    return means, covs


def emission_moments(mdl: Model, state_means, state_covs=None, method="ekf", alpha=math.sqrt(3), beta=2, kappa=1, t=None, inputs=None):
    """emissions_extended_kalman_filter (inference_ekf.py:768-855): (h(m, u, t), H P H^T + R) with H = jacfwd(h)(m, u, t) per row;
    emissions_unscented_kalman_filter (inference_ukf.py:507-612): the sigma points of (m, P) (_compute_sigmas, :45-60) through h, weighted
    mean (w_mean) and covariance (w_cov) + R.  state_means [rows, d], state_covs [rows, d, d] or None (point estimates: h(m) alone, second
    element None); t [rows], inputs [rows, d_u]: this row's time and inputs for an emission that reads them (Model(emission_ut=True))."""
    m = np.asarray(state_means, np.float64)
    rows, d = m.shape
    if getattr(mdl, "emission_ut", False):
        _CTX["u"] = None if inputs is None else np.asarray(inputs, np.float64)
        _CTX["t"] = np.zeros(rows) if t is None else np.asarray(t, np.float64).reshape(rows)
    if state_covs is None:
        return mdl.h(m), None
    P = np.asarray(state_covs, np.float64)
    if method == "ekf":
        H = mdl.Hjac(m)
        return mdl.h(m), H @ P @ np.swapaxes(H, -1, -2) + mdl.R
    lamb, w_mean, w_cov, _ = ukf_weights(d, alpha, beta, kappa, np.float64)
    X = ukf_sigmas(m, P, lamb)                                   # [rows, 2 d + 1, d]  (every sigma point of a row shares the row's inputs and time: _ctx_rows)
    Y = mdl.h(X.reshape(-1, d)).reshape(rows, X.shape[1], -1)
    ym = np.einsum("s,nsk->nk", w_mean, Y)
    dY = Y - ym[:, None]
    return ym, np.einsum("s,nsp,nsq->npq", w_cov, dY, dY) + mdl.R


# --------------------------------------------------------------------------------------
# linear model: smoother type 1 (discrete RTS on the pushed-forward (A, Q))
# --------------------------------------------------------------------------------------
def kf_pushforward(mdl: Model, t0, t1, dt0=0.01, max_steps=100000, dtype=np.float64):
    """compute_pushforward (continuous_discrete_linear_gaussian_ssm/inference.py:105-143): A' = F A, Q' = F Q + Q F^T +
    L Qc L^T from (I, 0) over [t0, t1] with the same Dopri5 loop.  t0, t1: [B]; returns A, Q: [B, d, d]."""
    dtype = np.dtype(dtype)
    mdl = mdl.cast(dtype)
    F = mdl.drift.W
    LQL = _LQL(mdl)
    B = np.asarray(t0).shape[0]
    d = mdl.d
    A0 = np.broadcast_to(np.eye(d, dtype=dtype), (B, d, d)).copy()
    Q0 = np.zeros((B, d, d), dtype)
    rhs = lambda yv: (F @ yv[0], F @ yv[1] + yv[1] @ F.T + LQL)
    return diffeqsolve(rhs, np.asarray(t0, dtype), np.asarray(t1, dtype), (A0, Q0), dt0, max_steps)


def kf_filter_inputs(mdl: Model, t, y, dyn_bias=None, B=None, D=None, inputs=None, dt0=0.01, dt_final=1e-10, max_steps=100000):
    """cdlgssm_filter with a dynamics bias and inputs, as the reference runs it (continuous_discrete_linear_gaussian_ssm/
    inference.py:555-632): per step k, with u = inputs[k]:  ll += N(y_k; H m + D u + d, H P H^T + R);  condition (psd_solve);
    (A, Q) = compute_pushforward(t_k, t_k+1);  m <- A m + B u + b (added WITHOUT being integrated, _predict :185-205),
    P <- A P A^T + Q.  The drift's own bias must be zero (it would be integrated).  float64; t [N,T] or [T], y [N,T,m],
    inputs [N,T,nu] or [T,nu]."""
    mdl = mdl.cast(np.float64)
    if mdl.drift.kind != "linear" or np.any(mdl.drift.b != 0):
        raise NotImplementedError("kf_filter_inputs: linear drift with zero (integrated) bias")
    y = np.asarray(y, np.float64)
    N, T, mm = y.shape
    d = mdl.d
    tt = np.broadcast_to(np.asarray(t, np.float64), (N, T)) if np.ndim(t) == 1 else np.asarray(t, np.float64)
    b = np.zeros(d) if dyn_bias is None else np.asarray(dyn_bias, np.float64)
    u = np.zeros((N, T, 0)) if inputs is None else np.broadcast_to(np.asarray(inputs, np.float64), (N, T, np.shape(inputs)[-1]))
    Bm = np.zeros((d, u.shape[-1])) if B is None else np.asarray(B, np.float64)
    Dm = np.zeros((mm, u.shape[-1])) if D is None else np.asarray(D, np.float64)
    H, R, hb = mdl.H, mdl.R, mdl.bias
    m = np.broadcast_to(mdl.m0, (N, d)).copy()
    P = np.broadcast_to(mdl.P0, (N, d, d)).copy()
    ll = np.zeros(N)
    out = {k: [] for k in ("filtered_means", "filtered_covariances", "predicted_means", "predicted_covariances")}
    eye = np.eye(mm)
    for k in range(T):
        S = H @ P @ H.T + R
        v = y[:, k] - (m @ H.T + u[:, k] @ Dm.T + hb)
        Lc = cholesky_lower(S)
        z = solve_lower(Lc, v[..., None])[..., 0]
        ll += -0.5 * np.einsum("ni,ni->n", z, z) - np.sum(np.log(np.diagonal(Lc, axis1=-2, axis2=-1)), -1) - 0.5 * mm * math.log(2 * math.pi)
        X = psd_solve(S, H @ P)                                          # K^T
        m = m + np.einsum("nri,nr->ni", X, v)
        P = symmetrize(P - np.swapaxes(X, -1, -2) @ S @ X)
        out["filtered_means"].append(m)
        out["filtered_covariances"].append(P)
        t1 = tt[:, k + 1] if k + 1 < T else tt[:, k] + dt_final
        A, Q = kf_pushforward(mdl, tt[:, k], t1, dt0, max_steps)
        m = np.einsum("nij,nj->ni", A, m) + u[:, k] @ Bm.T + b
        P = A @ P @ np.swapaxes(A, -1, -2) + Q
        out["predicted_means"].append(m)
        out["predicted_covariances"].append(P)
    res = {k: np.stack(v_, axis=1) for k, v_ in out.items()}
    res["marginal_loglik"] = ll
    return res


def kf_smoother_type1(mdl: Model, t, y, dt0=0.01, dt_final=1e-10, max_steps=100000, dtype=np.float64,
                      filtered: Optional[dict] = None):
    """cdlgssm_smoother(..., smoother_type='cd_smoother_1') (inference.py:694-823, _step_1 :746-773; Sarkka Alg. 3.17):
    C = psd_solve(Q + A P_f A^T, A P_f)^T;  m_s = m_f + C (m_s' - A m_f);  P_s = P_f + C (P_s' - A P_f A^T - Q) C^T;
    cross = C P_s' + m_s m_s'^T.  Linear drift with zero bias, no inputs.  The forward pass is this build's filter (moments
    integrated directly)."""
    dtype = np.dtype(dtype)
    if mdl.drift.kind != "linear" or np.any(mdl.drift.b != 0):
        raise NotImplementedError("type-1 smoother: linear drift with zero bias")
    if filtered is None:
        filtered = ekf_filter(mdl, t, y, "first", 1, dt0, dt_final, max_steps, dtype=dtype)
    mdlc = mdl.cast(dtype)
    t = np.asarray(t, dtype=dtype)
    fm, fP = filtered["filtered_means"], filtered["filtered_covariances"]
    N, T, d = fm.shape
    tt = np.broadcast_to(t, (N, T)) if t.ndim == 1 else t
    sm, sP = fm.copy(), fP.copy()
    cross = np.zeros((N, max(T - 1, 0), d, d), dtype)
    eye = np.eye(d, dtype=dtype)
    for k in range(T - 2, -1, -1):
        A, Q = kf_pushforward(mdlc, tt[:, k], tt[:, k + 1], dt0, max_steps, dtype)
        AP = A @ fP[:, k]
        Ppred = AP @ np.swapaxes(A, -1, -2) + Q
        C = np.swapaxes(psd_solve(Ppred, AP), -1, -2)
        sm[:, k] = fm[:, k] + np.einsum("nij,nj->ni", C, sm[:, k + 1] - np.einsum("nij,nj->ni", A, fm[:, k]))
        sP[:, k] = fP[:, k] + C @ (sP[:, k + 1] - Ppred) @ np.swapaxes(C, -1, -2)
        cross[:, k] = C @ sP[:, k + 1] + sm[:, k][:, :, None] * sm[:, k + 1][:, None, :]
    out = dict(filtered)
    out.update(smoothed_means=sm, smoothed_covariances=sP, smoothed_cross_covariances=cross)
    return out


# --------------------------------------------------------------------------------------
# d(marginal log-likelihood)/d(drift parameters): forward-mode sensitivities of the EKF recursion
# --------------------------------------------------------------------------------------
# The reference obtains this gradient by JAX reverse-mode AD through the same computation
# (ssm_temissions.py:550-568 `value_and_grad(_loss_fn)`); the exact derivative of the discretised
# algorithm is unique, so differentiating the recursion forward gives the same numbers.  Restated
# here as the checker for cdkf_ekf_loglik_grad_*; itself checked against central finite differences
# of ekf_filter's log-likelihood (tests/test_oracle.py).
def _drift_param_derivs(drift, x):
    """Returns (dfdth [N,P,d], dFdth [N,P,d,d], dFdx [N,d(i),d,d]) for the drift at x [N,d]."""
    N, d = x.shape
    if drift.kind == "lorenz63":
        dfdth = np.zeros((N, 3, 3), x.dtype)
        dfdth[:, 0, 0] = x[:, 1] - x[:, 0]
        dfdth[:, 1, 1] = x[:, 0]
        dfdth[:, 2, 2] = -x[:, 2]
        dFdth = np.zeros((N, 3, 3, 3), x.dtype)
        dFdth[:, 0, 0, 0], dFdth[:, 0, 0, 1] = -1, 1
        dFdth[:, 1, 1, 0] = 1
        dFdth[:, 2, 2, 2] = -1
        dFdx = np.zeros((N, 3, 3, 3), x.dtype)
        dFdx[:, 0, 1, 2], dFdx[:, 0, 2, 1] = -1, 1   # d/dx
        dFdx[:, 1, 2, 0] = 1                          # d/dy
        dFdx[:, 2, 1, 0] = -1                         # d/dz
        return dfdth, dFdth, dFdx
    if drift.kind == "linear":
        P = d * d + d
        dfdth = np.zeros((N, P, d), x.dtype)
        dFdth = np.zeros((N, P, d, d), x.dtype)
        for i in range(d):
            for j in range(d):
                dfdth[:, i * d + j, i] = x[:, j]
                dFdth[:, i * d + j, i, j] = 1
            dfdth[:, d * d + i, i] = 1
        return dfdth, dFdth, np.zeros((N, d, d, d), x.dtype)
    raise NotImplementedError(drift.kind)


def ekf_loglik_grad(mdl: Model, t, y, dt0=0.01, dt_final=1e-10, max_steps=100000, dtype=np.float64):
    """Returns (ll [N], grad [N, n_theta]) for the EKF with state_order first/second (identical for the drifts
    supported here), num_iter = 1.  theta ordering = drift.theta()."""
    dtype = np.dtype(dtype)
    mdl = mdl.cast(dtype)
    y = np.asarray(y, dtype=dtype)
    N, T, _ = y.shape
    d, mm = mdl.d, mdl.m
    drift = mdl.drift
    npar = drift.theta().size
    H, R, bias = mdl.H, mdl.R, mdl.bias
    LQL = _LQL(mdl)
    t0s, t1s = _t0_t1(t, dt_final, dtype)
    m = np.broadcast_to(mdl.m0, (N, d)).copy()
    P = np.broadcast_to(mdl.P0, (N, d, d)).copy()
    dm = np.zeros((N, npar, d), dtype)
    dP = np.zeros((N, npar, d, d), dtype)
    ll = np.zeros(N, dtype)
    g = np.zeros((N, npar), dtype)
    eye = np.eye(mm, dtype=dtype)
    T_ = lambda A: np.swapaxes(A, -1, -2)
    for k in range(T):
        # ---- update with sensitivities ----
        HP = H @ P
        S = HP @ H.T + R
        v = y[:, k] - (m @ H.T + bias)
        dHP = H @ dP                                    # [N,P,m,d]
        dS = dHP @ H.T                                  # [N,P,m,m]
        dv = -dm @ H.T                                  # [N,P,m]
        Lc = cholesky_lower(S)
        Sinv = solve_upper_from_lower(Lc, solve_lower(Lc, np.broadcast_to(eye, S.shape).copy()))
        w = np.einsum("nij,nj->ni", Sinv, v)
        with np.errstate(invalid="ignore", divide="ignore"):
            logdet_half = np.sum(np.log(np.diagonal(Lc, axis1=-2, axis2=-1)), axis=-1)
        ll = ll + (-0.5 * np.einsum("ni,ni->n", v, w) - logdet_half - 0.5 * mm * math.log(2 * math.pi))
        g = g + (-np.einsum("ni,npi->np", w, dv) + 0.5 * np.einsum("ni,npij,nj->np", w, dS, w)
                 - 0.5 * np.einsum("nij,npji->np", Sinv, dS))
        Sb = symmetrize(S) + dtype.type(1e-9) * eye
        Lb = cholesky_lower(Sb)
        X = solve_upper_from_lower(Lb, solve_lower(Lb, HP))          # [N,m,d]
        dSb = symmetrize(dS)
        rhs = dHP - dSb @ X[:, None]
        Lbp = np.broadcast_to(Lb[:, None], dSb.shape).copy().reshape(-1, mm, mm)
        dX = solve_upper_from_lower(Lbp, solve_lower(Lbp, rhs.reshape(-1, mm, d))).reshape(rhs.shape)
        m_new = m + np.einsum("nri,nr->ni", X, v)
        dm_new = dm + np.einsum("npri,nr->npi", dX, v) + np.einsum("nri,npr->npi", X, dv)
        SX = S @ X
        Tm = T_(X) @ SX
        dT = T_(dX) @ SX[:, None] + T_(X)[:, None] @ (dS @ X[:, None]) + T_(X)[:, None] @ (S[:, None] @ dX)
        P = symmetrize(P - Tm)
        dP = symmetrize(dP - dT)
        m, dm = m_new, dm_new
        # ---- predict with sensitivities (same Dormand-Prince steps as the primal) ----

        def rhs_all(yv):
            mm_, PP, dmm, dPP = yv
            F = drift.jac(mm_)
            dfdth, dFdth, dFdx = _drift_param_derivs(drift, mm_)
            dmdt = drift.f(mm_)
            dPdt = F @ PP + PP @ T_(F) + LQL
            ddm = np.einsum("nij,npj->npi", F, dmm) + dfdth
            dF = np.einsum("nkij,npk->npij", dFdx, dmm) + dFdth
            B = dF @ PP[:, None] + F[:, None] @ dPP
            return dmdt, dPdt, ddm, B + T_(B)

        m, P, dm, dP = diffeqsolve(rhs_all, t0s[:, k], t1s[:, k], (m, P, dm, dP), dt0, max_steps, err_components=2)
    return ll, g


def ukf_loglik_grad(mdl: Model, t, y, alpha=math.sqrt(3), beta=2, kappa=1, dt0=0.01, dt_final=1e-10, max_steps=100000,
                    dtype=np.float64):
    """(ll [N], grad [N, n_theta]) of the UNSCENTED filter's marginal log-likelihood w.r.t. the drift parameters -- what
    jax.value_and_grad of the fit_sgd loss yields with filter_hyperparams=UKFHyperParams() (ssm_temissions.py:500, 555-568 through
    models.py:393-408, inference_ukf.py:206-308) -- for the drifts whose sigma-point sums collapse exactly (Lorenz-63: quadratic;
    linear) and a linear emission: with P_s = sym(P) = chol(P) chol(P)^T the weighted sums of _predict / _condition_on are, for every
    (alpha, beta, kappa),
        dm/dt = f(m) + (0, -P_s02, P_s01),   dP/dt = F(m) P_s + P_s F(m)^T + L Qc L^T,   S = H P_s H^T + R,   C = P_s H^T,
    and this routine carries forward sensitivities through exactly these (the derivative of a function does not depend on how the
    function is written down).  Pinned in tests/test_oracle.py by central finite differences of ukf_filter -- the routine that forms
    the sigma points literally."""
    dtype = np.dtype(dtype)
    mdl = mdl.cast(dtype)
    y = np.asarray(y, dtype=dtype)
    N, T, _ = y.shape
    d, mm = mdl.d, mdl.m
    drift = mdl.drift
    if drift.kind not in ("lorenz63", "linear"):
        raise NotImplementedError("ukf_loglik_grad: the closed form of the sigma-point sums is written for Lorenz-63 and linear drifts")
    curved = drift.kind == "lorenz63"
    npar = drift.theta().size
    H, R, bias = mdl.H, mdl.R, mdl.bias
    LQL = _LQL(mdl)
    t0s, t1s = _t0_t1(t, dt_final, dtype)
    m = np.broadcast_to(mdl.m0, (N, d)).copy()
    P = np.broadcast_to(mdl.P0, (N, d, d)).copy()
    dm = np.zeros((N, npar, d), dtype)
    dP = np.zeros((N, npar, d, d), dtype)
    ll = np.zeros(N, dtype)
    g = np.zeros((N, npar), dtype)
    eye = np.eye(mm, dtype=dtype)
    T_ = lambda A: np.swapaxes(A, -1, -2)

    def curv(Ps):  # b(P_s) = (0, -P_s02, P_s01) on the last axis of [..., 3, 3]
        out = np.zeros(Ps.shape[:-1], Ps.dtype)
        if curved:
            out[..., 1] = -Ps[..., 0, 2]
            out[..., 2] = Ps[..., 0, 1]
        return out

    for k in range(T):
        Ps, dPs = symmetrize(P), symmetrize(dP)
        HP = H @ Ps
        S = HP @ H.T + R
        v = y[:, k] - (m @ H.T + bias)
        dHP = H @ dPs
        dS = dHP @ H.T
        dv = -dm @ H.T
        Lc = cholesky_lower(S)
        Sinv = solve_upper_from_lower(Lc, solve_lower(Lc, np.broadcast_to(eye, S.shape).copy()))
        w = np.einsum("nij,nj->ni", Sinv, v)
        with np.errstate(invalid="ignore", divide="ignore"):
            logdet_half = np.sum(np.log(np.diagonal(Lc, axis1=-2, axis2=-1)), axis=-1)
        ll = ll + (-0.5 * np.einsum("ni,ni->n", v, w) - logdet_half - 0.5 * mm * math.log(2 * math.pi))
        g = g + (-np.einsum("ni,npi->np", w, dv) + 0.5 * np.einsum("ni,npij,nj->np", w, dS, w)
                 - 0.5 * np.einsum("nij,npji->np", Sinv, dS))
        Sb = symmetrize(S) + dtype.type(1e-9) * eye
        Lb = cholesky_lower(Sb)
        X = solve_upper_from_lower(Lb, solve_lower(Lb, HP))          # K^T
        dSb = symmetrize(dS)
        rhs = dHP - dSb @ X[:, None]
        Lbp = np.broadcast_to(Lb[:, None], dSb.shape).copy().reshape(-1, mm, mm)
        dX = solve_upper_from_lower(Lbp, solve_lower(Lbp, rhs.reshape(-1, mm, d))).reshape(rhs.shape)
        m_new = m + np.einsum("nri,nr->ni", X, v)
        dm_new = dm + np.einsum("npri,nr->npi", dX, v) + np.einsum("nri,npr->npi", X, dv)
        SX = S @ X
        P = P - T_(X) @ SX                                            # (no symmetrize in the unscented update)
        dP = dP - (T_(dX) @ SX[:, None] + T_(X)[:, None] @ (dS @ X[:, None]) + T_(X)[:, None] @ (S[:, None] @ dX))
        m, dm = m_new, dm_new

        def rhs_all(yv):
            mm_, PP, dmm, dPP = yv
            PPs, dPPs = symmetrize(PP), symmetrize(dPP)
            F = drift.jac(mm_)
            dfdth, dFdth, dFdx = _drift_param_derivs(drift, mm_)
            dmdt = drift.f(mm_) + curv(PPs)
            dPdt = F @ PPs + PPs @ T_(F) + LQL
            ddm = np.einsum("nij,npj->npi", F, dmm) + dfdth + curv(dPPs)
            dF = np.einsum("nkij,npk->npij", dFdx, dmm) + dFdth
            B = dF @ PPs[:, None] + F[:, None] @ dPPs
            return dmdt, dPdt, ddm, B + T_(B)

        m, P, dm, dP = diffeqsolve(rhs_all, t0s[:, k], t1s[:, k], (m, P, dm, dP), dt0, max_steps, err_components=2)
    return ll, g


# --------------------------------------------------------------------------------------
# the same gradient by the discrete adjoint (reverse mode) -- all drift parameters, any registry drift
# --------------------------------------------------------------------------------------
# What jax.value_and_grad does in the reference (reverse mode through update + Dormand-Prince steps,
# diffrax_utils.py:49 RecursiveCheckpointAdjoint), written out: checker for the reverse-sweep HIP kernel
# (cdkf_ekf_loglik_grad_* with the MLP drift).  state_order 'first' or 'second' (the mean term 0.5 P grad(div f) of a drift
# with a non-zero grad(div f) -- the MLP -- is reversed by divgrad_vjp: third derivatives); one trajectory at a time, plain loops.
_DP_A = [[], [1 / 5], [3 / 40, 9 / 40], [44 / 45, -56 / 15, 32 / 9], [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
         [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656]]
_DP_B = [35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84]


def drift_vjp(drift, x, lam, G):
    """Gradient of  s(x, theta) = lam . f(x) + <G, F(x)>  w.r.t. (x, theta);  x [d], lam [d], G [d,d]."""
    d = x.shape[0]
    if drift.kind == "mlp":
        W1, W2, W3 = drift.W1, drift.W2, drift.W3
        z1 = W1 @ x + drift.b1
        a1 = np.tanh(z1)
        d1 = 1 - a1 * a1
        U = d1[:, None] * W1                      # tangent of a1 for the d unit directions  [h1,d]
        Tz = W2 @ U                               # tangent of z2                           [h2,d]
        a2 = np.tanh(W2 @ a1 + drift.b2)
        d2 = 1 - a2 * a2
        V = d2[:, None] * Tz                      # tangent of a2                           [h2,d]
        # reverse
        c2 = W3.T @ G                             # cotangent of V                          [h2,d]
        gW3 = np.outer(lam, a2) + G @ V.T
        gb3 = lam.copy()
        a2b = W3.T @ lam + np.sum(-2 * a2[:, None] * Tz * c2, axis=1)
        zt2 = d2[:, None] * c2                    # cotangent of Tz
        z2b = d2 * a2b
        gW2 = np.outer(z2b, a1) + zt2 @ U.T
        gb2 = z2b
        c1 = W2.T @ zt2                           # cotangent of U                          [h1,d]
        a1b = W2.T @ z2b + np.sum(-2 * a1[:, None] * W1 * c1, axis=1)
        z1b = d1 * a1b
        gW1 = np.outer(z1b, x) + d1[:, None] * c1
        gb1 = z1b
        xb = W1.T @ z1b
        return xb, np.concatenate([g.ravel() for g in (gW1, gb1, gW2, gb2, gW3, gb3)])
    if drift.kind == "custom":
        if drift._vjp is None:
            raise NotImplementedError("CallableDrift without vjp")
        xb, tb = drift._vjp(x, lam, G, drift.th, *(_ctx_rows(1) if drift.ut else ()))  # (ut: the single row's u [1, d_u] and t [1])
        return np.asarray(xb, np.float64), np.asarray(tb, np.float64)
    F = drift.jac(x[None])[0]
    xb = F.T @ lam
    if drift.kind == "lorenz63":
        th = np.array([lam[0] * (x[1] - x[0]) - G[0, 0] + G[0, 1], lam[1] * x[0] + G[1, 0], -lam[2] * x[2] - G[2, 2]])
        xb = xb + np.array([-G[1, 2] + G[2, 1], G[2, 0], -G[1, 0]])
        return xb, th
    if drift.kind == "linear":
        return xb, np.concatenate([(np.outer(lam, x) + G).ravel(), lam])
    if drift.kind == "lorenz96":
        for i in range(d):
            ip1, im1, im2 = (i + 1) % d, (i - 1) % d, (i - 2) % d
            xb[im1] += G[i, ip1] - G[i, im2]
            xb[ip1] += G[i, im1]
            xb[im2] -= G[i, im1]
        return xb, np.array([lam.sum()])
    raise NotImplementedError(drift.kind)


def divgrad_vjp(drift, x, u):
    """Gradient of  s(x, theta) = u . g(x),  g = grad(div f) = drift.divgrad,  w.r.t. (x, theta);  x [d], u [d].

    The reverse pass of the EKF's state_order='second' mean term 0.5 P g(m) (inference_ekf.py:108-116 as SURVEY.md
    section 0.5 reads it): with lam the cotangent of the mean's slope, u = 0.5 P lam.  Only the MLP has g != 0 among the
    registry drifts.  Reverse mode through MLPDrift.divgrad, line by line (same intermediate names):
        M = (W1 W3)^T, G = M * W2, td = d2 G, s = G d1, s2 = dd2 s, tc = s2 W2, tq = dd1 td + d1 tc, g = W1^T tq."""
    if drift.kind == "custom":
        if drift._gvjp is None:
            raise NotImplementedError("CallableDrift without gvjp")
        xb, tb = drift._gvjp(x, u, drift.th, *(_ctx_rows(1) if drift.ut else ()))
        return np.asarray(xb, np.float64), np.asarray(tb, np.float64)
    if drift.kind != "mlp":
        return np.zeros_like(x), np.zeros(drift.theta().size)
    W1, W2, W3 = drift.W1, drift.W2, drift.W3
    a1 = np.tanh(W1 @ x + drift.b1)
    d1 = 1 - a1 * a1
    a2 = np.tanh(W2 @ a1 + drift.b2)
    d2 = 1 - a2 * a2
    M = (W1 @ W3).T                                # [h2,h1]
    G = M * W2
    td = d2 @ G                                    # [h1]
    s = G @ d1                                     # [h2]
    s2 = -2 * a2 * d2 * s
    tc = s2 @ W2                                   # [h1]
    tq = -2 * a1 * d1 * td + d1 * tc
    # reverse
    r = W1 @ u                                     # cotangent of tq
    gW1 = np.outer(tq, u)
    td_b = -2 * a1 * d1 * r
    tc_b = d1 * r
    z1_b = r * (-2 * a1 * tq - 2 * d1 * d1 * td)   # through a1, d1 inside tq
    d2_b = G @ td_b
    G_b = np.outer(d2, td_b)
    s2_b = W2 @ tc_b
    gW2 = np.outer(s2, tc_b)
    s_b = -2 * a2 * d2 * s2_b
    z2_b = s2_b * (-2 * s) * d2 * (1 - 3 * a2 * a2) + d2_b * (-2 * a2 * d2)
    G_b = G_b + np.outer(s_b, d1)
    z1_b = z1_b + (s_b @ G) * (-2 * a1 * d1)
    gW2 = gW2 + G_b * M
    M_b = G_b * W2                                 # M_pq = sum_i W3_ip W1_qi
    gW1 = gW1 + M_b.T @ W3.T
    gW3 = W1.T @ M_b.T
    gW2 = gW2 + np.outer(z2_b, a1)
    z1_b = z1_b + (W2.T @ z2_b) * d1
    gW1 = gW1 + np.outer(z1_b, x)
    xb = W1.T @ z1_b
    return xb, np.concatenate([g_.ravel() for g_ in (gW1, z1_b, gW2, z2_b, gW3, np.zeros_like(drift.b3))])


def _step_sizes(t0, t1, dt0, tol, max_steps):
    """The dt sequence of the diffeqsolve loop above for one interval."""
    out = []
    tprev, tnext = t0, min(t0 + dt0, t1)
    while tprev < t1 and len(out) < max_steps:
        out.append(tnext - tprev)
        tprev = min(tnext, t1)
        tn = tnext + dt0
        tnext = t1 if tn > t1 - tol else tn
    return out


def ukf_curvature(drift, P):
    """The term the unscented filter's mean equation has beyond f(m) for a QUADRATIC drift (every alpha, beta, kappa): the sigma points
    are symmetric about the mean, f(m + o) + f(m - o) = 2 f(m) + 2 b(o, o) with b the drift's bilinear part, and sum_i o_i o_i^T = c^2 P,
    2 w_i c^2 = 1, so the weighted sum of inference_ukf.py:124-143 is f(m) + 0.5 sum_jk (d^2 f / dx_j dx_k) P_jk -- (0, -P_02, P_01) for
    Lorenz-63, P_{i+1,i-1} - P_{i-2,i-1} for Lorenz-96, nothing for a linear drift; the covariance equation is the extended filter's
    (f(m + o) - f(m - o) = 2 F(m) o exactly).  P [d,d] symmetric."""
    d = P.shape[0]
    if drift.kind == "linear":
        return np.zeros(d)
    if drift.kind == "lorenz63":
        return np.array([0.0, -P[0, 2], P[0, 1]])
    if drift.kind == "lorenz96":
        i = np.arange(d)
        return P[(i + 1) % d, (i - 1) % d] - P[(i - 2) % d, (i - 1) % d]
    raise NotImplementedError(f"ukf_curvature: drift {drift.kind} is not quadratic (its sigma-point sums have no closed form)")


def ukf_curvature_vjp(drift, lam):
    """Cotangent of P through lam . ukf_curvature(drift, P), as a symmetric matrix."""
    d = lam.shape[0]
    Pb = np.zeros((d, d))
    if drift.kind == "lorenz63":
        Pb[0, 2] -= lam[1]
        Pb[0, 1] += lam[2]
    elif drift.kind == "lorenz96":
        i = np.arange(d)
        np.add.at(Pb, ((i + 1) % d, (i - 1) % d), lam)
        np.add.at(Pb, ((i - 2) % d, (i - 1) % d), -lam)
    elif drift.kind != "linear":
        raise NotImplementedError(drift.kind)
    return 0.5 * (Pb + Pb.T)


def ukf_loglik_grad_all(mdl: Model, t, y, dt0=0.01, dt_final=1e-10, max_steps=100000):
    """(ll [N], grad [N, n_theta], dict of the other leaves' gradients) of the UNSCENTED filter's marginal log-likelihood -- what
    jax.value_and_grad yields in the reference with filter_hyperparams=UKFHyperParams() (ssm_temissions.py:500, 555-568 -> models.py:
    393-408, 708 -> inference_ukf.py:206-308) -- for the drifts whose sigma-point sums collapse exactly (Lorenz-63, Lorenz-96: quadratic;
    linear) and a linear emission: the discrete adjoint of the extended filter's recursion with ukf_curvature in the mean equation (the
    derivative of a function does not depend on how it is written down).  Pinned by central finite differences of ukf_filter -- the
    routine that forms the sigma points literally (tests/test_oracle.py)."""
    return ekf_loglik_grad_adjoint(mdl, t, y, dt0, dt_final, max_steps, full=True, state_order="first", ukf=True)


def ekf_loglik_grad_adjoint(mdl: Model, t, y, dt0=0.01, dt_final=1e-10, max_steps=100000, full=False, state_order="first", ukf=False,
                            num_iter=1, inputs=None):
    """Returns (ll [N], grad [N, n_theta]): EKF, state_order 'first' or 'second' (the mean term 0.5 P grad(div f), reversed by
    divgrad_vjp); ``num_iter`` update iterations as inference_ekf.py:153-199 runs them (each from the previous one's posterior, the
    log-likelihood term from the first one's inputs, symmetrize once at the end); float64.

    ``full=True`` adds a dict with the gradients w.r.t. every other parameter of the model, each with a leading [N]:
    m0, P0, L, Qc, H, bias, R and LQL (= the cotangent of L Qc L^T the first two are chained from).  Cotangents of the
    symmetric matrices (P0, LQL, R; Qc through LQL) are symmetric: they pair with symmetric perturbations, which is
    what every symmetric parametrisation (the reference's RealToPSDBijector) produces."""
    mdl = mdl.cast(np.float64)
    y = np.asarray(y, np.float64)
    t = np.asarray(t, np.float64)
    N, T, _ = y.shape
    d, mm = mdl.d, mdl.m
    drift = mdl.drift
    H, R, bias = mdl.H, mdl.R, mdl.bias
    LQL = _LQL(mdl)
    npar = drift.theta().size
    eye_m = np.eye(mm)
    u_all = _inputs(inputs, N, T, np.float64)   # f(x, u, t): the interval's inputs row and the stage times travel in _CTX (module header)
    sym = lambda A: 0.5 * (A + A.T)
    f = lambda x: drift.f(x[None])[0]
    jac = lambda x: drift.jac(x[None])[0]

    _A, _B = TABLEAUS[_ACTIVE[-1][0]]  # the Runge-Kutta method of the enclosing use_solver (fixed steps)
    _DP_A, _DP_B, NST = [list(r) for r in _A], list(_B), len(_B)
    adaptive = _ACTIVE[-1][1] is not None  # the accepted step sizes are constants of the reverse sweep (controller under stop_gradient)
    second = state_order == "second"
    assert state_order in ("first", "second")
    divgrad = lambda x: drift.divgrad(x[None])[0]

    def rhs(x, P):
        F = jac(x)
        A = F @ P
        fm = f(x) + 0.5 * P @ divgrad(x) if second else f(x)
        if ukf:  # (the unscented filter's moment equations in closed form: ukf_curvature)
            fm = fm + ukf_curvature(drift, sym(P))
        return fm, A + A.T + LQL

    def step_sizes(x, P, ta, tb):
        """The dt sequence of the solve over [ta, tb] from (x, P): the fixed-step loop's, or the sizes the controller accepts."""
        if not adaptive:
            return _step_sizes(ta, tb, dt0, 1e-10, max_steps)
        log = []

        def rhs_b(yv):
            km, kp = rhs(yv[0][0], yv[1][0])
            return km[None], kp[None]

        diffeqsolve(rhs_b, np.array([ta]), np.array([tb]), (x[None].copy(), P[None].copy()), dt0, max_steps, dt_log=log)
        return log

    def at_stage(ts, dt, i):
        """context time of stage i of the step that starts at ts: ts + c_i dt"""
        if ts is not None:
            _CTX["t"] = np.array([ts + dt * sum(_DP_A[i])])

    def stages(x, P, dt, ts=None):
        ks = []
        for i in range(NST):
            xs = x + dt * sum((_DP_A[i][j] * ks[j][0] for j in range(i)), np.zeros(d))
            Ps = P + dt * sum((_DP_A[i][j] * ks[j][1] for j in range(i)), np.zeros((d, d)))
            at_stage(ts, dt, i)
            ks.append(rhs(xs, Ps))
        return ks

    def stage_in(x, P, dt, ks, i):
        return (x + dt * sum((_DP_A[i][j] * ks[j][0] for j in range(i)), np.zeros(d)),
                P + dt * sum((_DP_A[i][j] * ks[j][1] for j in range(i)), np.zeros((d, d))))

    ll_out, g_out = np.zeros(N), np.zeros((N, npar))
    extra = {"m0": np.zeros((N, d)), "P0": np.zeros((N, d, d)), "LQL": np.zeros((N, d, d)), "H": np.zeros((N, mm, d)),
             "bias": np.zeros((N, mm)), "R": np.zeros((N, mm, mm))}
    for n in range(N):
        tn = t if t.ndim == 1 else t[n]
        # ---- forward sweep, keeping predicted and filtered moments ----
        mp, Pp, mf, Pf = [mdl.m0.copy()], [sym(mdl.P0)], [], []
        upd_in = []
        dts_fwd = {}
        ll = 0.0
        for k in range(T):
            m_, P_ = mp[k], Pp[k]
            S = H @ P_ @ H.T + R
            v = y[n, k] - (H @ m_ + bias)
            Lc = np.linalg.cholesky(S)
            w = np.linalg.solve(S, v)
            ll += -0.5 * v @ w - np.log(np.diag(Lc)).sum() - 0.5 * mm * math.log(2 * math.pi)
            its = [(m_, P_)]                                   # inputs of every update iteration
            for _it in range(num_iter):
                mi, Pi = its[-1]
                Si = H @ Pi @ H.T + R
                Xi = np.linalg.solve(sym(Si) + 1e-9 * eye_m, H @ Pi)
                its.append((mi + Xi.T @ (y[n, k] - (H @ mi + bias)), Pi - Xi.T @ Si @ Xi))
            upd_in.append(its[:-1])
            mf.append(its[-1][0])
            Pf.append(sym(its[-1][1]))
            if k + 1 < T:
                x, P = mf[k], Pf[k]
                _CTX["u"] = u_all[n:n + 1, k]
                dts_fwd[k] = step_sizes(x, P, tn[k], tn[k + 1])
                ts = tn[k]
                for dt in dts_fwd[k]:
                    ks = stages(x, P, dt, ts)
                    x = x + dt * sum(_DP_B[i] * ks[i][0] for i in range(NST))
                    P = P + dt * sum(_DP_B[i] * ks[i][1] for i in range(NST))
                    ts = ts + dt
                mp.append(x)
                Pp.append(P)
        # ---- backward sweep ----
        thb = np.zeros(npar)
        mb, Pb = np.zeros(d), np.zeros((d, d))          # adjoint of the filtered moments at k
        for k in range(T - 1, -1, -1):
            Pb = sym(Pb)
            for it in range(num_iter - 1, -1, -1):                 # the update iterations reversed; the likelihood term sits on the first
                m_, P_ = upd_in[k][it]
                HP = H @ P_
                S = HP @ H.T + R
                v = y[n, k] - (H @ m_ + bias)
                Sb = sym(S) + 1e-9 * eye_m
                X = np.linalg.solve(Sb, HP)
                vb = X @ mb
                Kb = np.outer(v, mb) - 2 * S @ X @ Pb              # cotangent of K^T  [m,d]
                Sbar = -X @ Pb @ X.T
                if it == 0:
                    Sinv = np.linalg.inv(S)
                    w = Sinv @ v
                    vb = vb - w
                    Sbar = Sbar + 0.5 * np.outer(w, w) - 0.5 * Sinv
                Ub = np.linalg.solve(Sb, Kb)                       # [m,d]
                Sbar = Sbar + sym(-X @ Ub.T)
                extra["R"][n] += Sbar
                extra["H"][n] += 2 * Sbar @ HP - np.outer(vb, m_) + Ub @ P_
                extra["bias"][n] -= vb
                Pb = Pb + sym(Ub.T @ H) + H.T @ Sbar @ H
                mb = mb - H.T @ vb
            if k == 0:
                extra["m0"][n], extra["P0"][n] = mb, sym(Pb)
                break
            # predict k-1 -> k: reverse the Dormand-Prince steps
            dts = dts_fwd[k - 1]
            _CTX["u"] = u_all[n:n + 1, k - 1]
            starts, tstarts = [(mf[k - 1], Pf[k - 1])], [tn[k - 1]]
            for dt in dts[:-1]:
                x, P = starts[-1]
                ks = stages(x, P, dt, tstarts[-1])
                starts.append((x + dt * sum(_DP_B[i] * ks[i][0] for i in range(NST)),
                               P + dt * sum(_DP_B[i] * ks[i][1] for i in range(NST))))
                tstarts.append(tstarts[-1] + dt)
            for (x, P), dt, ts in zip(reversed(starts), reversed(dts), reversed(tstarts)):
                ks = stages(x, P, dt, ts)
                Yb = [None] * NST
                for i in range(NST - 1, -1, -1):
                    lam = dt * (_DP_B[i] * mb + sum((_DP_A[j][i] * Yb[j][0] for j in range(i + 1, NST)), np.zeros(d)))
                    Lam = dt * (_DP_B[i] * Pb + sum((_DP_A[j][i] * Yb[j][1] for j in range(i + 1, NST)), np.zeros((d, d))))
                    Lam = sym(Lam)
                    xs, Ps = stage_in(x, P, dt, ks, i)
                    at_stage(ts, dt, i)
                    F = jac(xs)
                    xb, tb = drift_vjp(drift, xs, lam, 2 * Lam @ Ps)
                    Pbar = F.T @ Lam + Lam @ F
                    if second:  # dm/dt also carries 0.5 Ps g(xs)
                        xb2, tb2 = divgrad_vjp(drift, xs, 0.5 * Ps.T @ lam)
                        xb, tb = xb + xb2, tb + tb2
                        Pbar = Pbar + sym(0.5 * np.outer(lam, divgrad(xs)))
                    if ukf:
                        Pbar = Pbar + ukf_curvature_vjp(drift, lam)
                    thb += tb
                    extra["LQL"][n] += Lam
                    Yb[i] = (xb, Pbar)
                mb = mb + sum(Yb[i][0] for i in range(NST))
                Pb = sym(Pb + sum(Yb[i][1] for i in range(NST)))
        ll_out[n] = ll
        g_out[n] = thb
    if not full:
        return ll_out, g_out
    extra["L"] = extra["LQL"] @ mdl.L @ mdl.Qc.T + np.swapaxes(extra["LQL"], -1, -2) @ mdl.L @ mdl.Qc
    extra["Qc"] = mdl.L.T @ extra["LQL"] @ mdl.L
    return ll_out, g_out, extra


# --------------------------------------------------------------------------------------
# the unscented filter's gradient for ANY drift / emission: forward mode through the literal sigma-point recursion
# --------------------------------------------------------------------------------------
def _drift_dtheta(drift, x):
    """d f / d theta at x [B, d] -> [B, P, d] (the forward twin of drift_vjp's theta part)."""
    B, d = x.shape
    if drift.kind in ("lorenz63", "linear"):
        return _drift_param_derivs(drift, x)[0]
    if drift.kind == "lorenz96":
        return np.ones((B, 1, d), x.dtype)
    if drift.kind == "mlp":
        W1, W2, W3 = drift.W1, drift.W2, drift.W3
        a1, a2 = drift._fwd(x)
        d1, d2 = 1 - a1 * a1, 1 - a2 * a2
        A2 = W3[None] * d2[:, None, :]                                 # d f / d z2  [B, d, h2]
        A1 = np.einsum("bip,pq->biq", A2, W2) * d1[:, None, :]         # d f / d z1  [B, d, h1]
        eye = np.eye(d, dtype=x.dtype)
        blocks = [np.einsum("biq,bk->bqki", A1, x).reshape(B, -1, d),                  # W1 [h1, d]
                  np.swapaxes(A1, 1, 2),                                                   # b1
                  np.einsum("bip,bq->bpqi", A2, a1).reshape(B, -1, d),                   # W2 [h2, h1]
                  np.swapaxes(A2, 1, 2),                                                   # b2
                  np.einsum("ij,bp->bjpi", eye, a2).reshape(B, -1, d),                   # W3 [d, h2]
                  np.broadcast_to(eye, (B, d, d))]                                         # b3
        return np.concatenate(blocks, axis=1)
    if getattr(drift, "_dth", None) is not None:
        return np.asarray(drift._dth(x, drift.th, *drift._extra(x)), dtype=x.dtype)
    raise NotImplementedError("d f / d theta of a %s drift: give CallableDrift a dtheta callable" % drift.kind)


def ukf_loglik_grad_all_literal(mdl: Model, t, y, alpha=math.sqrt(3), beta=2, kappa=1, dt0=0.01, dt_final=1e-10, max_steps=100000,
                                inputs=None, h_eta=None):
    """(ll [N], grad [N, n_theta], dict of the other leaves' gradients) of the UNSCENTED filter's marginal log-likelihood for ANY drift
    and emission -- jax.value_and_grad through unscented_kalman_filter (ssm_temissions.py:500, 555-568 -> models.py:393-408, 708 ->
    inference_ukf.py:93-203) -- by forward-mode tangents carried through the literal sigma-point recursion of ukf_filter above: the
    Cholesky factor's tangent L' = L Phi(L^-1 sym(P') L^-T) (Phi: lower triangle, halved diagonal; jnp.linalg.cholesky symmetrises its
    input), the sigma points' x' = m' +- c L'_i, the drift's f'(x) = F(x) x' + d f / d theta_p, the Dormand-Prince combination (linear),
    the update's weighted sums, solves and log-density differentiated term by term.  One tangent per leaf ENTRY; symmetric leaves (P0,
    L Qc L^T, R) are perturbed symmetrically and their cotangents returned symmetric, as ekf_loglik_grad_adjoint(full=True) does.
    A non-linear emission (mdl.emission given as callables) uses its Jacobian mdl.Hjac(x) [B, m, d]; its parameters eta = [H | bias]
    enter through h_eta(x) -> [B, m, m d + m] (d h / d eta) when given, else H and bias are the linear emission's.
    Pinned by finite differences of ukf_filter (tests/test_oracle.py).  float64, fixed-step solvers."""
    dtype = np.dtype(np.float64)
    mdl = mdl.cast(dtype)
    y = np.asarray(y, dtype)
    N, T, _ = y.shape
    d, mm = mdl.d, mdl.m
    drift = mdl.drift
    u = _inputs(inputs, N, T, dtype)
    lamb, w_mean, w_cov, _W = ukf_weights(d, alpha, beta, kappa, dtype)
    c = np.sqrt(dtype.type(d) + lamb)
    S_ = 2 * d + 1
    nth = drift.theta().size
    iu_d, iu_m = np.triu_indices(d), np.triu_indices(mm)
    npd, npm = len(iu_d[0]), len(iu_m[0])
    linear_h = getattr(mdl, "emission", None) is None
    sizes = [nth, d, npd, npd, mm * d, mm, npm]
    off = np.concatenate([[0], np.cumsum(sizes)])
    Q = int(off[-1])
    T_ = lambda A: np.swapaxes(A, -1, -2)
    sym = lambda A: 0.5 * (A + T_(A))

    def sym_seed(n, iu):
        out = np.zeros((len(iu[0]), n, n), dtype)
        for e, (i, j) in enumerate(zip(*iu)):
            out[e, i, j] = out[e, j, i] = 1.0
        return out

    # tangents of the leaves: [Q, ...]
    dm0 = np.zeros((Q, d), dtype); dm0[off[1]:off[2]] = np.eye(d)
    dP0 = np.zeros((Q, d, d), dtype); dP0[off[2]:off[3]] = sym_seed(d, iu_d)
    dLQL = np.zeros((Q, d, d), dtype); dLQL[off[3]:off[4]] = sym_seed(d, iu_d)
    dH = np.zeros((Q, mm, d), dtype); dH[off[4]:off[5]] = np.eye(mm * d).reshape(mm * d, mm, d)
    db = np.zeros((Q, mm), dtype); db[off[5]:off[6]] = np.eye(mm)
    dR = np.zeros((Q, mm, mm), dtype); dR[off[6]:off[7]] = sym_seed(mm, iu_m)
    LQL = _LQL(mdl)
    H, bias, R = mdl.H, mdl.bias, mdl.R

    def chol_t(P, dP):
        """L [N,d,d], L' [N,Q,d,d] of sym(P)."""
        L = cholesky_lower(sym(P))
        Li = solve_lower(L, np.broadcast_to(np.eye(d, dtype=dtype), L.shape).copy())          # L^-1
        M = Li[:, None] @ sym(dP) @ T_(Li)[:, None]
        Phi = np.tril(M, -1) + 0.5 * M * np.eye(d)
        return L, L[:, None] @ Phi

    def sigmas_t(m, P, dm, dP):
        L, dL = chol_t(P, dP)
        cols, dcols = c * T_(L), c * T_(dL)                                # cols[n, i, :] = c L[:, i]
        X = np.concatenate([m[:, None], m[:, None] + cols, m[:, None] - cols], axis=1)            # [N, S, d]
        dX = np.concatenate([dm[:, :, None], dm[:, :, None] + dcols, dm[:, :, None] - dcols], axis=2)   # [N, Q, S, d]
        return X, dX

    def f_t(X, dX):
        """f(X) [N,S,d] and its tangent [N,Q,S,d]."""
        flat = X.reshape(-1, d)
        fX = drift.f(flat).reshape(X.shape)
        F = drift.jac(flat).reshape(X.shape + (d,))
        dfX = np.einsum("nsij,nqsj->nqsi", F, dX)
        if nth:
            Jth = _drift_dtheta(drift, flat).reshape(N, S_, nth, d)       # [N, S, P, d]
            dfX[:, :nth] += np.swapaxes(Jth, 1, 2)
        return fX, dfX

    def h_t(X, dX):
        flat = X.reshape(-1, d)
        if linear_h:
            Y = X @ H.T + bias
            dY = dX @ H.T + np.einsum("qij,nsj->nqsi", dH, X) + db[None, :, None, :]
            return Y, dY
        Y = mdl.h(flat).reshape(N, S_, mm)
        Hx = np.asarray(mdl.Hjac(flat), dtype).reshape(N, S_, mm, d)
        dY = np.einsum("nsij,nqsj->nqsi", Hx, dX)
        if h_eta is not None:
            He = np.asarray(h_eta(flat), dtype).reshape(N, S_, mm, mm * d + mm)
            dY[:, off[4]:off[6]] += np.moveaxis(He, -1, 1)
        return Y, dY

    def rhs(yv):
        m_t, P_t, dm_t, dP_t = yv
        X, dX = sigmas_t(m_t, P_t, dm_t, dP_t)
        fX, dfX = f_t(X, dX)
        dm = np.einsum("nsi,s->ni", fX, w_mean)
        ddm = np.einsum("nqsi,s->nqi", dfX, w_mean)
        Xc, fc = X - m_t[:, None], fX - dm[:, None]
        dXc, dfc = dX - dm_t[:, :, None], dfX - ddm[:, :, None]
        foo = np.einsum("s,nsi,nsj->nij", w_cov, fc, Xc)
        dfoo = np.einsum("s,nqsi,nsj->nqij", w_cov, dfc, Xc) + np.einsum("s,nsi,nqsj->nqij", w_cov, fc, dXc)
        return dm, foo + T_(foo) + LQL, ddm, dfoo + T_(dfoo) + dLQL[None]

    t0s, t1s = _t0_t1(t, dt_final, dtype)
    m = np.broadcast_to(mdl.m0, (N, d)).copy()
    P = np.broadcast_to(mdl.P0, (N, d, d)).copy()
    dm = np.broadcast_to(dm0, (N, Q, d)).copy()
    dP = np.broadcast_to(dP0, (N, Q, d, d)).copy()
    ll = np.zeros(N, dtype)
    g = np.zeros((N, Q), dtype)
    eye_m = np.eye(mm, dtype=dtype)
    for k in range(T):
        _set_step(u, t0s, k)
        X, dX = sigmas_t(m, P, dm, dP)
        Y, dY = h_t(X, dX)
        ym = np.einsum("s,nsj->nj", w_mean, Y)
        dym = np.einsum("s,nqsj->nqj", w_mean, dY)
        Yc, dYc = Y - ym[:, None], dY - dym[:, :, None]
        Xc, dXc = X - m[:, None], dX - dm[:, :, None]
        S = np.einsum("s,nsi,nsj->nij", w_cov, Yc, Yc) + R
        dS = np.einsum("s,nqsi,nsj->nqij", w_cov, dYc, Yc)
        dS = dS + T_(dS) + dR[None]
        C = np.einsum("s,nsi,nsj->nij", w_cov, Xc, Yc)                                   # [N, d, m]
        dC = np.einsum("s,nqsi,nsj->nqij", w_cov, dXc, Yc) + np.einsum("s,nsi,nqsj->nqij", w_cov, Xc, dYc)
        v, dv = y[:, k] - ym, -dym
        # log-density (S as given; its Cholesky factor symmetrises the input) and its tangent
        Lc = cholesky_lower(sym(S))
        Sinv = solve_upper_from_lower(Lc, solve_lower(Lc, np.broadcast_to(eye_m, S.shape).copy()))
        w = np.einsum("nij,nj->ni", Sinv, v)
        with np.errstate(invalid="ignore", divide="ignore"):
            ll = ll + (-0.5 * np.einsum("ni,ni->n", v, w) - np.sum(np.log(np.diagonal(Lc, axis1=-2, axis2=-1)), axis=-1)
                       - 0.5 * mm * math.log(2 * math.pi))
        dSs = sym(dS)
        g = g + (-np.einsum("ni,nqi->nq", w, dv) + 0.5 * np.einsum("ni,nqij,nj->nq", w, dSs, w) - 0.5 * np.einsum("nij,nqji->nq", Sinv, dSs))
        # gain K = psd_solve(S, C^T)^T: Sb = sym(S) + 1e-9 I;  X_ = Sb^-1 C^T  [N, m, d]
        Sb = sym(S) + dtype.type(1e-9) * eye_m
        Lb = cholesky_lower(Sb)
        Xg = solve_upper_from_lower(Lb, solve_lower(Lb, T_(C)))
        rhs_ = T_(dC) - dSs @ Xg[:, None]
        Lbq = np.broadcast_to(Lb[:, None], (N, Q, mm, mm)).reshape(-1, mm, mm)
        dXg = solve_upper_from_lower(Lbq, solve_lower(Lbq, rhs_.reshape(-1, mm, d))).reshape(N, Q, mm, d)
        K, dK = T_(Xg), T_(dXg)
        m_new = m + np.einsum("nij,nj->ni", K, v)
        dm = dm + np.einsum("nqij,nj->nqi", dK, v) + np.einsum("nij,nqj->nqi", K, dv)
        KS = K @ S
        dP = dP - (dK @ T_(KS)[:, None] + K[:, None] @ dS @ T_(K)[:, None] + KS[:, None] @ T_(dK))
        P = P - KS @ T_(K)
        m = m_new
        if k + 1 < T:   # (the last predict does not enter the log-likelihood)
            m, P, dm, dP = diffeqsolve(rhs, t0s[:, k], t1s[:, k], (m, P, dm, dP), dt0, max_steps, err_components=2)

    def unpack_sym(col, n, iu):
        out = np.zeros((N, n, n), dtype)
        for e, (i, j) in enumerate(zip(*iu)):
            out[:, i, j] = out[:, j, i] = col[:, e] if i == j else 0.5 * col[:, e]
        return out
    extra = {"m0": g[:, off[1]:off[2]], "P0": unpack_sym(g[:, off[2]:off[3]], d, iu_d), "LQL": unpack_sym(g[:, off[3]:off[4]], d, iu_d),
             "H": g[:, off[4]:off[5]].reshape(N, mm, d), "bias": g[:, off[5]:off[6]], "R": unpack_sym(g[:, off[6]:off[7]], mm, iu_m)}
    extra["L"] = extra["LQL"] @ mdl.L @ mdl.Qc.T + T_(extra["LQL"]) @ mdl.L @ mdl.Qc
    extra["Qc"] = mdl.L.T @ extra["LQL"] @ mdl.L
    return ll, g[:, :nth], extra


# --------------------------------------------------------------------------------------
# synthetic data (time-grid recipe of simulation_utils.py:46-49)
# --------------------------------------------------------------------------------------
def irregular_times(rng, N, T, T_total):
    u = rng.uniform(0.0, 1.0, size=(N, T))
    s = np.cumsum(u, axis=1)
    return s / s[:, -1:] * T_total


def simulate(mdl: Model, t, rng, h=1e-3):
    """Euler-Maruyama states at step h from x0 ~ N(m0,P0); y = H x + bias + N(0,R).  Data realism only."""
    N, T = t.shape
    d, m = mdl.d, mdl.m
    Lc = mdl.L @ np.linalg.cholesky(mdl.Qc)
    x = mdl.m0 + rng.standard_normal((N, d)) @ np.linalg.cholesky(mdl.P0).T
    cur = t[:, 0].copy()
    ys = np.zeros((N, T, m))
    Rc = np.linalg.cholesky(mdl.R)
    for k in range(T):
        gap = t[:, k] - cur
        nsub = int(max(1, math.ceil(gap.max() / h))) if gap.max() > 0 else 0
        if nsub:
            hh = (gap / nsub)[:, None]
            for _ in range(nsub):
                x = x + hh * mdl.drift.f(x) + np.sqrt(hh) * (rng.standard_normal((N, d)) @ Lc.T)
        cur = t[:, k].copy()
        ys[:, k] = x @ mdl.H.T + mdl.bias + rng.standard_normal((N, m)) @ Rc.T
    return ys


def lorenz63_model(m_obs=3, P0_scale=5.0):
    """C2/C3 of SURVEY.md section 8d: sigma=10, rho=28, beta=8/3, L=Qc=I, H=I3 (or first rows), R=I, m0=0, P0=5I."""
    d = 3
    return Model(Lorenz63Drift(), np.eye(d), np.eye(d), np.eye(d)[:m_obs], np.zeros(m_obs), np.eye(m_obs), np.zeros(d),
                 P0_scale * np.eye(d))
