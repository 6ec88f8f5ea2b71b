/*
 * cdkf_oracle.c -- plain-C restatement of the reference's continuous-discrete EKF filter, used as
 * (a) a second, independently written checker beside oracle/cdkf_oracle.py and (b) the CPU baseline
 * ("port") that bench.py times on the host cores next to the MI355X numbers.
 *
 * TEST INFRASTRUCTURE ONLY: nothing under cd_dynamax_amd/ links, loads or calls this file.
 *
 * Follows (paths relative to /root/reference):
 *   extended_kalman_filter   src/continuous_discrete_nonlinear_gaussian_ssm/inference_ekf.py:202-326
 *   _predict                 .../inference_ekf.py:46-148     _condition_on  .../inference_ekf.py:153-199
 *   diffeqsolve              src/utils/diffrax_utils.py:40-165 (Dopri5, ConstantStepSize, dt0 = 0.01; the
 *                            stepping loop of diffrax 0.4.0 is restated as in cdkf_oracle.py)
 *   psd_solve / symmetrize   dynamax/utils/utils.py:202-211
 *   MVN log_prob             TFP 0.20.1 MultivariateNormalFullCovariance (inference_ekf.py:286)
 *   drifts                   LearnableLinear / LearnableLorenz63, cdnlgssm_utils.py:50-83; Lorenz-96 build-defined
 *
 * Parity pinning: validated against oracle/cdkf_oracle.py (tests/test_oracle_c.py), which itself is pinned
 * to the reference's known-answer constants and test equalities (see its header).  Dense arithmetic
 * exactly as the reference writes it (full F@P + P@F^T, Jacobian every stage); OpenMP over trajectories
 * mirrors jax.vmap.  REAL is double or float (compiled twice).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef REAL
#define REAL double
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#ifndef SUFFIX
#define SUFFIX _f64
#endif
#define FN(name) CAT(name, SUFFIX)

#define DMAX 48 /* largest state_dim / emission_dim handled */

typedef struct {
  int kind, d, m, order, num_iter;
  const double* theta;
  REAL W[DMAX * DMAX], b[DMAX];       /* linear drift */
  REAL LQL[DMAX * DMAX], LQLz[DMAX * DMAX];
  REAL H[DMAX * DMAX], hb[DMAX], R[DMAX * DMAX];
  REAL th[4];
  REAL dt0, tol;
  long max_steps;
} ctx_t;

static void drift_f(const ctx_t* c, const REAL* x, REAL* f) {
  const int d = c->d;
  if (c->kind == 0) {
    for (int i = 0; i < d; ++i) {
      REAL s = 0;
      for (int j = 0; j < d; ++j) s += c->W[i * d + j] * x[j];
      f[i] = s + c->b[i];
    }
  } else if (c->kind == 1) {
    f[0] = c->th[0] * (x[1] - x[0]);
    f[1] = x[0] * (c->th[1] - x[2]) - x[1];
    f[2] = x[0] * x[1] - c->th[2] * x[2];
  } else {
    for (int i = 0; i < d; ++i)
      f[i] = (x[(i + 1) % d] - x[(i + d - 2) % d]) * x[(i + d - 1) % d] - x[i] + c->th[0];
  }
}

static void drift_jac(const ctx_t* c, const REAL* x, REAL* F) {
  const int d = c->d;
  if (c->kind == 0) {
    memcpy(F, c->W, sizeof(REAL) * d * d);
  } else if (c->kind == 1) {
    F[0] = -c->th[0]; F[1] = c->th[0]; F[2] = 0;
    F[3] = c->th[1] - x[2]; F[4] = -1; F[5] = -x[0];
    F[6] = x[1]; F[7] = x[0]; F[8] = -c->th[2];
  } else {
    memset(F, 0, sizeof(REAL) * d * d);
    for (int i = 0; i < d; ++i) {
      F[i * d + (i + 1) % d] += x[(i + d - 1) % d];
      F[i * d + (i + d - 2) % d] -= x[(i + d - 1) % d];
      F[i * d + (i + d - 1) % d] += x[(i + 1) % d] - x[(i + d - 2) % d];
      F[i * d + i] -= 1;
    }
  }
}

/* rhs of the moment ODEs, state y = [m (d), P (d*d)] (inference_ekf.py:76-123); all registry drifts have
 * grad(div f) = 0, so 'second' == 'first' here */
static void rhs(const ctx_t* c, const REAL* y, REAL* dy) {
  const int d = c->d;
  REAL F[DMAX * DMAX];
  drift_f(c, y, dy);
  if (c->order == 0) return;
  drift_jac(c, y, F);
  const REAL* P = y + d;
  REAL* dP = dy + d;
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < d; ++j) {
      REAL a = 0, bb = 0;
      for (int k = 0; k < d; ++k) {
        a += F[i * d + k] * P[k * d + j];
        bb += P[i * d + k] * F[j * d + k];
      }
      dP[i * d + j] = a + bb + c->LQL[i * d + j];
    }
}

static const double A_[6][5] = {{0},
                                {1.0 / 5},
                                {3.0 / 40, 9.0 / 40},
                                {44.0 / 45, -56.0 / 15, 32.0 / 9},
                                {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729},
                                {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656}};
static const double B_[6] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84};

/* one Dopri5 step: k_j = dt f(stage_j); stage_i = y0 + sum_j a_ij k_j (increment summed first) */
static void dopri5_step(const ctx_t* c, REAL* y, int ns, REAL dt, REAL* work) {
  REAL* k[6];
  for (int j = 0; j < 6; ++j) k[j] = work + j * ns;
  REAL* ys = work + 6 * ns;
  for (int i = 0; i < 6; ++i) {
    if (i == 0) {
      memcpy(ys, y, sizeof(REAL) * ns);
    } else {
      for (int e = 0; e < ns; ++e) {
        REAL acc = (REAL)A_[i][0] * k[0][e];
        for (int j = 1; j < i; ++j) acc += (REAL)A_[i][j] * k[j][e];
        ys[e] = y[e] + acc;
      }
    }
    rhs(c, ys, k[i]);
    for (int e = 0; e < ns; ++e) k[i][e] *= dt;
  }
  for (int e = 0; e < ns; ++e) {
    REAL acc = (REAL)B_[0] * k[0][e];
    for (int j = 2; j < 6; ++j) acc += (REAL)B_[j] * k[j][e];
    y[e] += acc;
  }
}

static void integrate(const ctx_t* c, REAL* y, int ns, REAL t0, REAL t1, REAL* work) {
  REAL tprev = t0, tnext = t0 + c->dt0;
  if (tnext > t1) tnext = t1;
  long steps = 0;
  while (tprev < t1 && steps < c->max_steps) {
    dopri5_step(c, y, ns, tnext - tprev, work);
    tprev = tnext < t1 ? tnext : t1;
    REAL tn = tnext + c->dt0;
    tnext = (tn > t1 - c->tol) ? t1 : tn;
    ++steps;
  }
}

/* lower Cholesky in place on the lower triangle of A (n x n); NaN on a non-positive pivot */
static void chol(REAL* A, int n) {
  for (int j = 0; j < n; ++j) {
    REAL s = A[j * n + j];
    for (int k = 0; k < j; ++k) s -= A[j * n + k] * A[j * n + k];
    REAL p = (REAL)sqrt((double)s);
    A[j * n + j] = p;
    for (int i = j + 1; i < n; ++i) {
      REAL v = A[i * n + j];
      for (int k = 0; k < j; ++k) v -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = v / p;
    }
  }
}

static void update(const ctx_t* c, REAL* mP, const REAL* yobs, REAL* ll) {
  const int d = c->d, m = c->m;
  REAL* mm = mP;
  REAL* P = mP + d;
  REAL HP[DMAX * DMAX], S[DMAX * DMAX], Lc[DMAX * DMAX], X[DMAX * DMAX], KS[DMAX * DMAX], v[DMAX], z[DMAX];
  for (int it = 0; it < c->num_iter; ++it) {
    for (int r = 0; r < m; ++r)
      for (int j = 0; j < d; ++j) {
        REAL s = 0;
        for (int k = 0; k < d; ++k) s += c->H[r * d + k] * P[k * d + j];
        HP[r * d + j] = s;
      }
    for (int r = 0; r < m; ++r)
      for (int q = 0; q < m; ++q) {
        REAL s = 0;
        for (int k = 0; k < d; ++k) s += HP[r * d + k] * c->H[q * d + k];
        S[r * m + q] = s + c->R[r * m + q];
      }
    for (int r = 0; r < m; ++r) {
      REAL s = 0;
      for (int k = 0; k < d; ++k) s += c->H[r * d + k] * mm[k];
      v[r] = yobs[r] - (s + c->hb[r]);
    }
    if (it == 0) { /* log-likelihood with the predicted moments (inference_ekf.py:285-286) */
      memcpy(Lc, S, sizeof(REAL) * m * m);
      chol(Lc, m);
      REAL q = 0, ld = 0;
      for (int i = 0; i < m; ++i) {
        REAL w = v[i];
        for (int k = 0; k < i; ++k) w -= Lc[i * m + k] * z[k];
        z[i] = w / Lc[i * m + i];
        q += z[i] * z[i];
        ld += (REAL)log((double)Lc[i * m + i]);
      }
      *ll += (REAL)-0.5 * q - ld - (REAL)(0.5 * m * 1.8378770664093454835606594728112);
    }
    for (int r = 0; r < m; ++r) /* psd_solve(S, HP) */
      for (int q = 0; q <= r; ++q) {
        REAL s = (REAL)0.5 * (S[r * m + q] + S[q * m + r]);
        if (r == q) s += (REAL)1e-9;
        Lc[r * m + q] = s;
      }
    chol(Lc, m);
    for (int j = 0; j < d; ++j) {
      for (int i = 0; i < m; ++i) {
        REAL w = HP[i * d + j];
        for (int k = 0; k < i; ++k) w -= Lc[i * m + k] * X[k * d + j];
        X[i * d + j] = w / Lc[i * m + i];
      }
      for (int i = m - 1; i >= 0; --i) {
        REAL w = X[i * d + j];
        for (int k = i + 1; k < m; ++k) w -= Lc[k * m + i] * X[k * d + j];
        X[i * d + j] = w / Lc[i * m + i];
      }
    }
    for (int i = 0; i < d; ++i)
      for (int q = 0; q < m; ++q) {
        REAL s = 0;
        for (int r = 0; r < m; ++r) s += X[r * d + i] * S[r * m + q];
        KS[i * m + q] = s;
      }
    REAL Pn[DMAX * DMAX];
    for (int i = 0; i < d; ++i)
      for (int j = 0; j < d; ++j) {
        REAL s = 0;
        for (int q = 0; q < m; ++q) s += KS[i * m + q] * X[q * d + j];
        Pn[i * d + j] = P[i * d + j] - s;
      }
    for (int i = 0; i < d; ++i) {
      REAL s = 0;
      for (int r = 0; r < m; ++r) s += X[r * d + i] * v[r];
      mm[i] += s;
    }
    memcpy(P, Pn, sizeof(REAL) * d * d);
  }
  for (int i = 0; i < d; ++i) /* symmetrize */
    for (int j = i + 1; j < d; ++j) {
      REAL s = (REAL)0.5 * (P[i * d + j] + P[j * d + i]);
      P[i * d + j] = P[j * d + i] = s;
    }
}

static void lql(const double* L, const double* Qc, int d, double scale, REAL* out) {
  REAL Lr[DMAX * DMAX], LQ[DMAX * DMAX];
  for (int i = 0; i < d * d; ++i) Lr[i] = (REAL)L[i] * (REAL)scale;
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < d; ++j) {
      REAL s = 0;
      for (int k = 0; k < d; ++k) s += Lr[i * d + k] * (REAL)Qc[k * d + j];
      LQ[i * d + j] = s;
    }
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < d; ++j) {
      REAL s = 0;
      for (int k = 0; k < d; ++k) s += LQ[i * d + k] * Lr[j * d + k];
      out[i * d + j] = s;
    }
}

/* t [N,T], y [N,T,m]; outputs [N,T,d] / [N,T,d,d] (any may be NULL); returns 0, or -1 on bad sizes */
int FN(cdkf_oracle_ekf_filter)(int drift_kind, int d, int m, const double* theta, const double* L, const double* Qc,
                               const double* H, const double* hb, const double* R, const double* m0, const double* P0,
                               int state_order, int num_iter, double dt0, double dt_final, long max_steps,
                               double cov_rescaling, long N, long T, const REAL* t, const REAL* y, REAL* ll, REAL* fm,
                               REAL* fP, REAL* pm, REAL* pP, int nthreads) {
  if (d < 1 || m < 1 || d > DMAX || m > DMAX || drift_kind < 0 || drift_kind > 2) return -1;
  ctx_t c;
  memset(&c, 0, sizeof(c));
  c.kind = drift_kind; c.d = d; c.m = m; c.order = state_order; c.num_iter = num_iter;
  c.dt0 = (REAL)dt0; c.max_steps = max_steps;
  c.tol = sizeof(REAL) == 8 ? (REAL)1e-10 : (REAL)1e-6;
  if (drift_kind == 0) {
    for (int i = 0; i < d * d; ++i) c.W[i] = (REAL)theta[i];
    for (int i = 0; i < d; ++i) c.b[i] = (REAL)theta[d * d + i];
  } else {
    for (int i = 0; i < (drift_kind == 1 ? 3 : 1); ++i) c.th[i] = (REAL)theta[i];
  }
  lql(L, Qc, d, 1.0, c.LQL);
  lql(L, Qc, d, cov_rescaling, c.LQLz);
  for (int i = 0; i < m * d; ++i) c.H[i] = (REAL)H[i];
  for (int i = 0; i < m; ++i) c.hb[i] = (REAL)hb[i];
  for (int i = 0; i < m * m; ++i) c.R[i] = (REAL)R[i];
  const int ns = d + d * d;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
  {
    REAL* work = (REAL*)malloc(sizeof(REAL) * ns * 8);
    REAL* s = work + 7 * ns;
#pragma omp for schedule(static)
    for (long n = 0; n < N; ++n) {
      for (int i = 0; i < d; ++i) s[i] = (REAL)m0[i];
      for (int i = 0; i < d * d; ++i) s[d + i] = (REAL)P0[i];
      REAL acc = 0;
      for (long k = 0; k < T; ++k) {
        update(&c, s, y + (n * T + k) * m, &acc);
        if (fm) memcpy(fm + (n * T + k) * d, s, sizeof(REAL) * d);
        if (fP) memcpy(fP + (n * T + k) * d * d, s + d, sizeof(REAL) * d * d);
        const REAL t0 = t[n * T + k];
        const REAL t1 = (k + 1 < T) ? t[n * T + k + 1] : t0 + (REAL)dt_final;
        if (state_order == 0) {
          integrate(&c, s, d, t0, t1, work);
          const REAL sq = (REAL)sqrt((double)(t1 - t0));
          for (int i = 0; i < d * d; ++i) s[d + i] += sq * c.LQLz[i];
        } else {
          integrate(&c, s, ns, t0, t1, work);
        }
        if (pm) memcpy(pm + (n * T + k) * d, s, sizeof(REAL) * d);
        if (pP) memcpy(pP + (n * T + k) * d * d, s + d, sizeof(REAL) * d * d);
      }
      ll[n] = acc;
    }
    free(work);
  }
  return 0;
}
