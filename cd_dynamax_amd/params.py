"""Parameter / hyper-parameter / posterior containers of the CDNLGSSM hot path.

Same names, fields and ``.f(x, u, t)`` accessors as the reference, so that code written against
cd_dynamax builds these objects unchanged (NumPy arrays instead of jax arrays):

* ``LearnableVector / LearnableMatrix / LearnableLinear / LearnableLorenz63`` and
  ``ParamsCDNLGSSM{,Dynamics,Emissions}``:
  /root/reference/src/continuous_discrete_nonlinear_gaussian_ssm/cdnlgssm_utils.py:38-209
* ``ParamsLGSSMInitial, PosteriorGSSMFiltered, PosteriorGSSMSmoothed``:
  /root/reference/dynamax/linear_gaussian_ssm/inference.py:19-33, 112-143
* ``EKFHyperParams`` inference_ekf.py:34-44, ``UKFHyperParams`` inference_ukf.py:25-33
* ``ParameterProperties``: /root/reference/dynamax/parameters.py:25 (metadata only here)

``LearnableLorenz96`` and ``LearnableMLP`` are build-defined drift families (BASELINE.json configs 4, 5;
the reference's NeuralNetDrift notebooks are absent from the mount).
"""
from __future__ import annotations

import math
from typing import Any, NamedTuple, Optional

import numpy as np


class ParameterProperties(NamedTuple):
    trainable: bool = True
    constrainer: Any = None


class LearnableVector(NamedTuple):
    params: Any

    def f(self, x=None, u=None, t=None):
        return self.params


class LearnableMatrix(NamedTuple):
    params: Any

    def f(self, x=None, u=None, t=None):
        return self.params


class LearnableLinear(NamedTuple):
    """f(x) = weights @ x + bias"""
    weights: Any
    bias: Any

    def f(self, x, u=None, t=None):
        return np.asarray(self.weights) @ np.asarray(x) + np.asarray(self.bias)


class LearnableLorenz63(NamedTuple):
    sigma: Any
    rho: Any
    beta: Any

    def f(self, x, u=None, t=None):
        return np.array([
            self.sigma * (x[1] - x[0]),
            x[0] * (self.rho - x[2]) - x[1],
            x[0] * x[1] - self.beta * x[2],
        ])


class LearnableLorenz96(NamedTuple):
    """f_i = (x_{i+1} - x_{i-2}) x_{i-1} - x_i + forcing (cyclic)."""
    forcing: Any

    def f(self, x, u=None, t=None):
        x = np.asarray(x)
        return (np.roll(x, -1) - np.roll(x, 2)) * np.roll(x, 1) - x + self.forcing


class LearnableMLP(NamedTuple):
    """f(x) = W3 tanh(W2 tanh(W1 x + b1) + b2) + b3."""
    W1: Any
    b1: Any
    W2: Any
    b2: Any
    W3: Any
    b3: Any

    def f(self, x, u=None, t=None):
        a1 = np.tanh(np.asarray(self.W1) @ np.asarray(x) + np.asarray(self.b1))
        a2 = np.tanh(np.asarray(self.W2) @ a1 + np.asarray(self.b2))
        return np.asarray(self.W3) @ a2 + np.asarray(self.b3)


class ConstantStepSize(NamedTuple):
    """diffrax.ConstantStepSize(): fixed steps of dt0 -- the default of the reference (src/utils/diffrax_utils.py:47)."""


class PIDController(NamedTuple):
    """diffrax.PIDController(rtol, atol, pcoeff, icoeff, dcoeff, dtmin, dtmax, safety, factormin, factormax) for
    ``diffeqsolve_settings['stepsize_controller']``; the remaining diffrax options are at their defaults (RMS norm, force_dtmin=True, no
    step_ts / jump_ts).  A diffrax controller object with the same attributes is accepted as well."""
    rtol: float
    atol: float
    pcoeff: float = 0.0
    icoeff: float = 1.0
    dcoeff: float = 0.0
    dtmin: Optional[float] = None
    dtmax: Optional[float] = None
    safety: float = 0.9
    factormin: float = 0.2
    factormax: float = 10.0


class LearnableCustomDrift(NamedTuple):
    """A user-defined drift for the HIP path.  The reference takes any callable (cdnlgssm_utils.py:38-61); here the drift
    is C source compiled at run time into the register-resident sweep kernels (include/cdkf.h,
    cdkf_custom_drift_register):

      f_src        statements computing ``fx[i]`` from ``x[j]`` and ``theta[k]``         e.g. "fx[0] = x[1]; fx[1] = -theta[0]*sin(x[0]);"
      jac_src      statements assigning the non-zero ``F[i][j]`` = d f_i / d x_j        e.g. "F[0][1] = R(1); F[1][0] = -theta[0]*cos(x[0]);"
                   or None: the Jacobian is derived from ``f_src`` by dual numbers (what ``jacfwd`` does in the reference)
      divgrad_src  statements assigning ``g[i]`` = d/dx_i sum_j d f_j / d x_j; "auto": derived from ``f_src`` (second derivatives by
                   nested dual numbers); "" : identically zero; None: EKF ``state_order='second'`` is refused

    ``cdnlgssm_loglik_and_grad`` / ``fit_sgd`` differentiate the log-likelihood w.r.t. ``theta`` (dual numbers again: ``state_order``
    'first', or 'second' with ``divgrad_src=""`` or ``"auto"`` -- the latter on the reverse sweep, third derivatives of f).  Wherever derivatives are derived, ``f_src`` is also compiled with a dual-number
    scalar type ``T`` in place of ``R``: declare temporaries ``auto`` or ``T`` there, not ``R``.

    ``R`` is the compute type (float or double).  state_dim, emission_dim <= 6: the register-resident kernels.  Beyond that
    (state_dim <= 64, as far as the workgroup kernels' LDS holds the shape: d = m = 40 in float64, 60 in float32) the same ``f_src`` is
    compiled into the workgroup-per-trajectory kernels -- ``jac_src`` must then be None and ``divgrad_src`` None, "" or "auto" (a thread
    per direction of the Jacobian, all by dual numbers), the emission linear; filters, smoother, forecast and -- on the shape-generic
    reverse sweep, up to 43 dimensions in float64 -- the gradients (``cdnlgssm_loglik_and_grad_all`` at any state_dim); 10 - 20 s of
    compilation per variant on first use.  ``py_f`` (optional) is the same function as a Python
    callable ``f(x, u, t)`` for host-side use; it is never called by the filter."""
    theta: Any
    f_src: str
    jac_src: Optional[str] = None
    divgrad_src: Optional[str] = None
    py_f: Optional[Any] = None

    def f(self, x, u=None, t=None):
        if self.py_f is None:
            raise NotImplementedError("LearnableCustomDrift: no Python callable was given (py_f)")
        return self.py_f(x, u, t)


class LearnableCustomEmission(NamedTuple):
    """A user-defined (non-linear) emission function for the HIP path -- the counterpart of passing any callable as
    ``ParamsCDNLGSSMEmissions.emission_function`` in the reference (linearised by the EKF with jacfwd, inference_ekf.py:258-259;
    evaluated at the sigma points by the UKF, inference_ukf.py:162-203):

      h_src     statements computing ``hx[r]`` from ``x[k]`` and ``eta[j]``            e.g. "hx[0] = eta[0] * sin(x[0]);"
      hjac_src  statements assigning the non-zero ``H[r][k]`` = d h_r / d x_k          e.g. "H[0][0] = eta[0] * cos(x[0]);"
                or None: derived from ``h_src`` by dual numbers (the reference's jacfwd; temporaries in ``h_src`` then ``auto`` / ``T``)

    ``eta``: the emission's parameter vector, at most emission_dim * (state_dim + 1) entries (it travels in the H / bias
    block of the C model).  state_dim, emission_dim <= 6: every entry point.  Up to 16: the filters and the log-likelihood gradients
    (the literal recursions on dual numbers, ``csrc/cdkf_ukf_tangent_kernels.h``; ``h_src`` is then compiled over dual numbers whether
    or not ``hjac_src`` is given: temporaries ``auto`` / ``T``) and, for the smoother, the workgroup kernels' backward sweep.  ``py_h`` (optional): the same function as a Python callable."""
    eta: Any
    h_src: str
    hjac_src: Optional[str] = None
    py_h: Optional[Any] = None

    def f(self, x, u=None, t=None):
        if self.py_h is None:
            raise NotImplementedError("LearnableCustomEmission: no Python callable was given (py_h)")
        return self.py_h(x, u, t)


class ParamsLGSSMInitial(NamedTuple):
    mean: Any
    cov: Any


class ParamsCDNLGSSMDynamics(NamedTuple):
    drift: Any
    diffusion_coefficient: Any
    diffusion_cov: Any
    approx_order: Any = 2.0


class ParamsCDNLGSSMEmissions(NamedTuple):
    emission_function: Any
    emission_cov: Any


class ParamsCDNLGSSM(NamedTuple):
    initial: ParamsLGSSMInitial
    dynamics: ParamsCDNLGSSMDynamics
    emissions: ParamsCDNLGSSMEmissions


class PosteriorGSSMFiltered(NamedTuple):
    marginal_loglik: Any
    filtered_means: Optional[Any] = None
    filtered_covariances: Optional[Any] = None
    predicted_means: Optional[Any] = None
    predicted_covariances: Optional[Any] = None


class PosteriorGSSMSmoothed(NamedTuple):
    marginal_loglik: Any
    filtered_means: Any
    filtered_covariances: Any
    smoothed_means: Any
    smoothed_covariances: Any
    smoothed_cross_covariances: Optional[Any] = None


class GSSMForecast(NamedTuple):
    """cdnlgssm_utils.py:227-248 (only the Gaussian-forecast fields are produced by the HIP path)."""
    forecasted_state_means: Optional[Any] = None
    forecasted_state_covariances: Optional[Any] = None
    forecasted_emission_means: Optional[Any] = None
    forecasted_emission_covariances: Optional[Any] = None
    forecasted_state_path: Optional[Any] = None
    forecasted_emission_path: Optional[Any] = None


class EKFHyperParams(NamedTuple):
    dt_final: float = 1e-10
    state_order: str = "second"
    emission_order: str = "first"
    smooth_order: str = "first"
    cov_rescaling: float = 1.0
    diffeqsolve_settings: dict = {}


class UKFHyperParams(NamedTuple):
    """inference_ukf.py:25-33.  ``sigma_points`` is not in the reference: True makes every drift take the kernels that form the
    2 d + 1 sigma points and factorise the covariance in every Runge-Kutta stage (``CDKF_FLAG_UKF_SIGMA_POINTS``) -- the
    reference's literal arithmetic, including its NaN when a STAGE covariance loses positive definiteness
    (inference_ukf.py:57, :138); by default Lorenz-63 / linear drifts use the exact closed form of the weighted sums."""
    dt_final: float = 1e-10
    alpha: float = math.sqrt(3)
    beta: int = 2
    kappa: int = 1
    diffeqsolve_settings: dict = {}
    sigma_points: bool = False


class EnKFHyperParams(NamedTuple):
    """Present only so that ``isinstance`` dispatch can reject it explicitly: the ensemble filter is
    stochastic (JAX PRNG / Brownian tree) and out of scope."""
    dt_final: float = 1e-10
    N_particles: int = 100
