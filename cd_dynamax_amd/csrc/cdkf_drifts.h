// cdkf_drifts.h -- drift registry for the register-resident kernels: f(x), Jacobian F = df/dx with a
// compile-time sparsity mask, and g = grad(div f) for the reference's "second order" mean term
// (0.5 * jnp.trace(H_t @ P) == 0.5 * P g, inference_ekf.py:111-114).
//
// Reference classes: LearnableLinear (cdnlgssm_utils.py:50-61), LearnableLorenz63 (:63-83).
// Lorenz-96 is build-defined (BASELINE.json config 4).
#pragma once
#include "cdkf_math.h"

namespace cdkf {

// f(x) = W x + b
template <typename R, int D>
struct DriftLinear {
  static constexpr int NTHETA = D * D + D;
  static constexpr bool HAS_G = false;
  static constexpr bool CONST_JAC = true;
  R W[D][D];
  R b[D];
  __host__ void load(const double* th) {
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < D; ++j) W[i][j] = R(th[i * D + j]);
    for (int i = 0; i < D; ++i) b[i] = R(th[D * D + i]);
  }
  static constexpr bool nz(int, int) { return true; }
  CDKF_DEV void f(const R* x, R (&fx)[D]) const {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      R s = W[i][0] * x[0];
#pragma unroll
      for (int j = 1; j < D; ++j) s = rfma(W[i][j], x[j], s);
      fx[i] = s + b[i];
    }
  }
  CDKF_DEV void jac(const R*, R (&F)[D][D]) const {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) F[i][j] = W[i][j];
  }
  CDKF_DEV void divgrad(const R*, R (&g)[D]) const {
#pragma unroll
    for (int i = 0; i < D; ++i) g[i] = R(0);
  }
};

// Lorenz-63: f = [sigma (y - x), x (rho - z) - y, x y - beta z]
template <typename R, int D>
struct DriftLorenz63 {
  static_assert(D == 3, "Lorenz-63 has state_dim 3");
  static constexpr int NTHETA = 3;
  static constexpr bool HAS_G = false;  // dF_ii/dx is constant
  static constexpr bool CONST_JAC = false;
  R sigma, rho, beta;
  __host__ void load(const double* th) {
    sigma = R(th[0]);
    rho = R(th[1]);
    beta = R(th[2]);
  }
  static constexpr bool nz(int i, int k) { return !((i == 0) && (k == 2)); }
  CDKF_DEV void f(const R* x, R (&fx)[3]) const {
    fx[0] = sigma * (x[1] - x[0]);
    fx[1] = rfma(x[0], rho - x[2], -x[1]);
    fx[2] = rfma(x[0], x[1], -beta * x[2]);
  }
  CDKF_DEV void jac(const R* x, R (&F)[3][3]) const {
    F[0][0] = -sigma;
    F[0][1] = sigma;
    F[0][2] = R(0);
    F[1][0] = rho - x[2];
    F[1][1] = R(-1);
    F[1][2] = -x[0];
    F[2][0] = x[1];
    F[2][1] = x[0];
    F[2][2] = -beta;
  }
  CDKF_DEV void divgrad(const R*, R (&g)[3]) const { g[0] = g[1] = g[2] = R(0); }
};

// Lorenz-96: f_i = (x_{i+1} - x_{i-2}) x_{i-1} - x_i + F   (cyclic indices, D >= 4)
template <typename R, int D>
struct DriftLorenz96 {
  static_assert(D >= 4, "Lorenz-96 needs state_dim >= 4");
  static constexpr int NTHETA = 1;
  static constexpr bool HAS_G = false;
  static constexpr bool CONST_JAC = false;
  R forcing;
  __host__ void load(const double* th) { forcing = R(th[0]); }
  static constexpr bool nz(int i, int k) {
    return k == i || k == (i + 1) % D || k == (i + D - 1) % D || k == (i + D - 2) % D;
  }
  CDKF_DEV void f(const R* x, R (&fx)[D]) const {
#pragma unroll
    for (int i = 0; i < D; ++i)
      fx[i] = rfma(x[(i + 1) % D] - x[(i + D - 2) % D], x[(i + D - 1) % D], forcing - x[i]);
  }
  CDKF_DEV void jac(const R* x, R (&F)[D][D]) const {
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
      for (int j = 0; j < D; ++j) F[i][j] = R(0);
      F[i][(i + 1) % D] = x[(i + D - 1) % D];
      F[i][(i + D - 2) % D] = -x[(i + D - 1) % D];
      F[i][(i + D - 1) % D] = x[(i + 1) % D] - x[(i + D - 2) % D];
      F[i][i] = R(-1);
    }
  }
  CDKF_DEV void divgrad(const R*, R (&g)[D]) const {
#pragma unroll
    for (int i = 0; i < D; ++i) g[i] = R(0);
  }
};

}  // namespace cdkf
