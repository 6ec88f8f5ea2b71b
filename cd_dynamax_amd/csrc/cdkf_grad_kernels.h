// cdkf_grad_kernels.h -- d(marginal log-likelihood)/d(drift parameters) by forward sensitivities.
//
// The reference gets this gradient from jax.value_and_grad through the filter (ssm_temissions.py:550-568,
// the fit_sgd loss of the parameter-estimation tutorials).  The derivative of the discretised recursion is
// unique, so it is computed here in forward mode: alongside (m, P) each lane carries (dm, dP) = d(m, P)/d theta_p
// for ONE drift parameter p, advanced by the same Dormand-Prince steps (the tableau is linear in the state
// and dt does not depend on theta) and the differentiated measurement update.
//
// Mapping: lane <-> (trajectory n, parameter p), p fastest, so the n_theta lanes of one trajectory sit in
// the same wavefront and their observation loads coalesce into one request.  The primal recursion is
// repeated per parameter (n_theta = 3 for Lorenz-63): everything stays in registers, no cross-lane traffic,
// and the cost is n_theta x (primal + one tangent) -- for the small theta of the registry drifts that is
// on par with a reverse sweep that would have to store or recompute every RK stage.
//
// Derivation (per parameter, ' = d/d theta_p):
//   predict  m'.  = F m' + df/dtheta                 P'. = B + B^T,  B = F' P + F P',
//            F'  = sum_i (dF/dm_i) m'_i + dF/dtheta
//   update   S = H P H^T + R, v = y - H m - b;  S' = H P' H^T, v' = -H m'
//            ll' = -w^T v' + 0.5 w^T S' w - 0.5 tr(S^-1 S'),  w = S^-1 v          (S as given: TFP log_prob)
//            Sb = sym(S) + 1e-9 I, X = Sb^-1 H P (K = X^T):   X' = Sb^-1 (H P' - sym(S') X)
//            m+' = m' + X'^T v + X^T v';   P+' = sym(P' - X'^T S X - X^T S' X - X^T S X')
#pragma once
#include "cdkf_reg_kernels.h"

namespace cdkf {

// ---- per-drift parameter derivatives ------------------------------------------------------------------
// init(p), bind(drift): this lane's parameter and the drift whose parameters it differentiates;
// dtheta(x, dfdth, dFdth): d f / d theta_p and d F / d theta_p at x for this lane's p;
// dstate(x, dm, dF): dF += sum_i (dF/dx_i)(x) dm_i.
// (a run-time compiled drift gets all three from dual numbers: launch_custom.hip, cdkf_dual.h)
template <typename R, int D, typename Drift>
struct DriftGrad;

template <typename R>
struct DriftGrad<R, 3, DriftLorenz63<R, 3>> {
  static constexpr int NPAR = 3;
  R e0, e1, e2;
  CDKF_DEV void init(int p) {
    e0 = p == 0 ? R(1) : R(0);
    e1 = p == 1 ? R(1) : R(0);
    e2 = p == 2 ? R(1) : R(0);
  }
  CDKF_DEV void dtheta(const R* x, R (&df)[3], R (&dF)[3][3]) const {
    df[0] = e0 * (x[1] - x[0]);
    df[1] = e1 * x[0];
    df[2] = -e2 * x[2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) dF[i][j] = R(0);
    dF[0][0] = -e0;
    dF[0][1] = e0;
    dF[1][0] = e1;
    dF[2][2] = -e2;
  }
  CDKF_DEV void bind(const DriftLorenz63<R, 3>&) {}
  CDKF_DEV void dstate(const R*, const R* dm, R (&dF)[3][3]) const {
    dF[1][0] -= dm[2];
    dF[1][2] -= dm[0];
    dF[2][0] += dm[1];
    dF[2][1] += dm[0];
  }
  // the unscented filter's mean equation: f(m) + b(P), b(P) = (0, -P_02, P_01) -- the sigma-point sum f_X^T w_mean of
  // inference_ukf.py:124-143 collapsed for this quadratic drift (cdkf_lpe_kernels.h, LpeRhs<R, true>; exact for every alpha, beta, kappa)
  static constexpr bool kCurved = true;
  static CDKF_DEV void curvature(const R* Ppacked, R* dm) {
    dm[1] -= Ppacked[sidx<3>(0, 2)];
    dm[2] += Ppacked[sidx<3>(0, 1)];
  }
};

template <typename R, int D>
struct DriftGrad<R, D, DriftLinear<R, D>> {
  static constexpr int NPAR = D * D + D;
  int p;
  CDKF_DEV void init(int p_) { p = p_; }
  CDKF_DEV void dtheta(const R* x, R (&df)[D], R (&dF)[D][D]) const {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      R s = (p == D * D + i) ? R(1) : R(0);
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const R e = (p == i * D + j) ? R(1) : R(0);
        dF[i][j] = e;
        s = rfma(e, x[j], s);
      }
      df[i] = s;
    }
  }
  CDKF_DEV void bind(const DriftLinear<R, D>&) {}
  CDKF_DEV void dstate(const R*, const R*, R (&)[D][D]) const {}
  static constexpr bool kCurved = false;  // a linear drift has no curvature: its unscented moment equations are the extended filter's
  static CDKF_DEV void curvature(const R*, R*) {}
};

// ---- primal + tangent moment ODE -----------------------------------------------------------------------
// state vector: [m (D), P packed (NP), m' (D), P' packed (NP)]
// UKF: the unscented filter's moment equations in the closed form of its sigma-point sums (mean: + b(P), tangent: + b(P'))
template <typename R, int D, typename Drift, bool UKF = false>
struct EkfSensRhs {
  static constexpr int NS = Dims<D>::NS;
  const Drift& drift;
  const DriftGrad<R, D, Drift>& dg;
  const R* LQL;
  static constexpr bool kTime = DriftTime<Drift>::value;
  CDKF_DEV void set_time(R t) const {
    if constexpr (kTime) drift.set_time(t);
  }
  CDKF_DEV void operator()(const R (&y)[2 * NS], R (&dy)[2 * NS]) const {
    R F[D][D], f[D], dfth[D], dF[D][D];
    drift.f(y, f);
    drift.jac(y, F);
    dg.dtheta(y, dfth, dF);
    dg.dstate(y, y + NS, dF);
    const R* P = y + D;
    const R* dm = y + NS;
    const R* dP = y + NS + D;
    R A[D][D], B[D][D];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) {
        R acc = R(0), accb = R(0);
#pragma unroll
        for (int k = 0; k < D; ++k) {
          acc = rfma(F[i][k], P[sidx<D>(k, j)], acc);
          accb = rfma(dF[i][k], P[sidx<D>(k, j)], accb);
          accb = rfma(F[i][k], dP[sidx<D>(k, j)], accb);
        }
        A[i][j] = acc;
        B[i][j] = accb;
      }
#pragma unroll
    for (int i = 0; i < D; ++i) {
      dy[i] = f[i];
      R s = dfth[i];
#pragma unroll
      for (int k = 0; k < D; ++k) s = rfma(F[i][k], dm[k], s);
      dy[NS + i] = s;
    }
    if constexpr (UKF) {
      DriftGrad<R, D, Drift>::curvature(P, dy);
      DriftGrad<R, D, Drift>::curvature(dP, dy + NS);
    }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = i; j < D; ++j) {
        dy[D + sidx<D>(i, j)] = (A[i][j] + A[j][i]) + LQL[sidx<D>(i, j)];
        dy[NS + D + sidx<D>(i, j)] = B[i][j] + B[j][i];
      }
  }
};

// ---- differentiated measurement update ---------------------------------------------------------------------
template <typename R, int D, int M, typename Args>
CDKF_DEV void ekf_update_sens(const Args& a, R (&ys)[2 * Dims<D>::NS], const R (&yobs)[M], LlAcc& ll, double& g,
                              int& st) {
  constexpr int NS = Dims<D>::NS;
  bool bad = false;
  R* dms = ys + NS;
  R HP[M][D], dHP[M][D], S[M][M], dS[M][M], v[M], dv[M];
#pragma unroll
  for (int r = 0; r < M; ++r) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      R s = R(0), ds = R(0);
#pragma unroll
      for (int k = 0; k < D; ++k) {
        s = rfma(a.H[r][k], ys[D + sidx<D>(k, j)], s);
        ds = rfma(a.H[r][k], dms[D + sidx<D>(k, j)], ds);
      }
      HP[r][j] = s;
      dHP[r][j] = ds;
    }
    R s = R(0), ds = R(0);
#pragma unroll
    for (int k = 0; k < D; ++k) {
      s = rfma(a.H[r][k], ys[k], s);
      ds = rfma(a.H[r][k], dms[k], ds);
    }
    v[r] = yobs[r] - (s + a.hb[r]);
    dv[r] = -ds;
  }
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int c = 0; c < M; ++c) {
      R s = R(0), ds = R(0);
#pragma unroll
      for (int k = 0; k < D; ++k) {
        s = rfma(HP[r][k], a.H[c][k], s);
        ds = rfma(dHP[r][k], a.H[c][k], ds);
      }
      S[r][c] = s + a.Rm[r][c];
      dS[r][c] = ds;
    }
  {  // log-likelihood term and its derivative (S as given)
    R Lc[M][M], inv[M];
    chol_lower<R, M>(S, Lc, inv, bad);
    R q = R(0), pinv = R(1), z[M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
      R w = v[i];
#pragma unroll
      for (int k = 0; k < i; ++k) w = rfma(-Lc[i][k], z[k], w);
      z[i] = w * inv[i];
      q = rfma(z[i], z[i], q);
      pinv *= inv[i];
    }
    ll.add((double)q, (double)pinv, M);
    R Si[M][M + 1];  // [S^-1 | w]
#pragma unroll
    for (int r = 0; r < M; ++r) {
#pragma unroll
      for (int c = 0; c < M; ++c) Si[r][c] = r == c ? R(1) : R(0);
      Si[r][M] = v[r];
    }
    chol_solve<R, M, M + 1>(Lc, inv, Si);
    R acc = R(0);
#pragma unroll
    for (int r = 0; r < M; ++r) {
      acc = rfma(-Si[r][M], dv[r], acc);
      R sw = R(0), tr = R(0);
#pragma unroll
      for (int c = 0; c < M; ++c) {
        sw = rfma(dS[r][c], Si[c][M], sw);
        tr = rfma(Si[r][c], dS[c][r], tr);
      }
      acc = rfma(R(0.5) * Si[r][M], sw, acc);
      acc = rfma(R(-0.5), tr, acc);
    }
    g += (double)acc;
  }
  R Sb[M][M], dSb[M][M];
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int c = 0; c < M; ++c) {
      R s = R(0.5) * (S[r][c] + S[c][r]);
      if (r == c) s += R(1e-9);
      Sb[r][c] = s;
      dSb[r][c] = R(0.5) * (dS[r][c] + dS[c][r]);
    }
  R Lb[M][M], invb[M];
  chol_lower<R, M>(Sb, Lb, invb, bad);
  R X[M][D], dX[M][D];
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int j = 0; j < D; ++j) X[r][j] = HP[r][j];
  chol_solve<R, M, D>(Lb, invb, X);
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      R s = dHP[r][j];
#pragma unroll
      for (int c = 0; c < M; ++c) s = rfma(-dSb[r][c], X[c][j], s);
      dX[r][j] = s;
    }
  chol_solve<R, M, D>(Lb, invb, dX);
  R SX[M][D], dSX[M][D];  // S X  and  S' X + S X'
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      R s = R(0), ds = R(0);
#pragma unroll
      for (int c = 0; c < M; ++c) {
        s = rfma(S[r][c], X[c][j], s);
        ds = rfma(dS[r][c], X[c][j], ds);
        ds = rfma(S[r][c], dX[c][j], ds);
      }
      SX[r][j] = s;
      dSX[r][j] = ds;
    }
  // T = X^T S X,  T' = X'^T (S X) + X^T (S' X + S X');  P+ = sym(P - T), P+' = sym(P' - T')
  R Pn[Dims<D>::NP], dPn[Dims<D>::NP];
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = i; j < D; ++j) {
      R tij = R(0), tji = R(0), dtij = R(0), dtji = R(0);
#pragma unroll
      for (int c = 0; c < M; ++c) {
        tij = rfma(X[c][i], SX[c][j], tij);
        tji = rfma(X[c][j], SX[c][i], tji);
        dtij = rfma(dX[c][i], SX[c][j], dtij);
        dtij = rfma(X[c][i], dSX[c][j], dtij);
        dtji = rfma(dX[c][j], SX[c][i], dtji);
        dtji = rfma(X[c][j], dSX[c][i], dtji);
      }
      const R p = ys[D + sidx<D>(i, j)], dp = dms[D + sidx<D>(i, j)];
      Pn[sidx<D>(i, j)] = (i == j) ? p - tij : R(0.5) * ((p - tij) + (p - tji));
      dPn[sidx<D>(i, j)] = (i == j) ? dp - dtij : R(0.5) * ((dp - dtij) + (dp - dtji));
    }
#pragma unroll
  for (int i = 0; i < D; ++i) {
    R s = ys[i], ds = dms[i];
#pragma unroll
    for (int r = 0; r < M; ++r) {
      s = rfma(X[r][i], v[r], s);
      ds = rfma(dX[r][i], v[r], ds);
      ds = rfma(X[r][i], dv[r], ds);
    }
    ys[i] = s;
    dms[i] = ds;
  }
#pragma unroll
  for (int e = 0; e < Dims<D>::NP; ++e) {
    ys[D + e] = Pn[e];
    dms[D + e] = dPn[e];
  }
  if (bad) st |= kStatusNotPd;
}

template <typename R, int D, int M, typename Drift>
struct GradArgs {
  RegArgs<R, D, M, Drift> a;
  R* grad;  // [N, n_theta]
};

// ---- log-likelihood + gradient sweep -----------------------------------------------------------------------
// GENERIC: run-time Runge-Kutta tableau / adaptive steps (opts.solver, opts.adaptive) instead of the pinned Dormand-Prince
// UKF: the unscented filter's log-likelihood (cdkf_ukf_loglik_grad_*): predict by the closed form of the sigma-point sums, update =
// the extended filter's algebra on the packed symmetric covariance (a linear emission passes the sigma points through exactly, and
// the reference's unscented update has no symmetrisation to lose: inference_ukf.py:162-203); the covariance the update would draw
// its sigma points from must be positive definite (jnp.linalg.cholesky, inference_ukf.py:57): NaN and the NOT_PD flag otherwise.
template <typename R, int D, int M, typename Drift, bool GENERIC = false, bool UKF = false>
CDKF_DEV void ekf_grad_reg_body(const GradArgs<R, D, M, Drift>& ga) {
  constexpr int NS = Dims<D>::NS;
  constexpr int NP = Dims<D>::NP;
  constexpr int NPAR = DriftGrad<R, D, Drift>::NPAR;
  const RegArgs<R, D, M, Drift>& a = ga.a;
  const long total = a.N * NPAR;
  const long gid0 = reg_unit_index(ga.a.lanes, ga.a.xcd_shift);
  if (reg_group_is_surplus(gid0, ga.a.lanes, total)) return;
  const bool live = gid0 < total;
  const long gid = live ? gid0 : total - 1;  // idle lanes shadow the last (trajectory, parameter) pair
  const long n = gid / NPAR;
  const int p = (int)(gid - n * NPAR);

  const R* __restrict__ tp = a.t + n * a.t_sn;
  const R* __restrict__ yp = a.y + n * a.y_sn;

  R ys[2 * NS];
#pragma unroll
  for (int i = 0; i < D; ++i) ys[i] = a.m0[i];
#pragma unroll
  for (int e = 0; e < NP; ++e) ys[D + e] = a.P0[e];
#pragma unroll
  for (int e = 0; e < NS; ++e) ys[NS + e] = R(0);

  LlAcc ll;
  double g = 0.0;
  int st = 0;
  const auto C = TabSel<R, GENERIC>::get(a);
  DriftGrad<R, D, Drift> dg;
  dg.init(p);
  dg.bind(a.drift);
  EkfSensRhs<R, D, Drift, UKF> rhs{a.drift, dg, a.LQL};

  R tcur = tp[0];
  if (a.T > 1) tp += a.t_sk;
  R tnext_obs = tp[0];
  R ycur[M];
#pragma unroll
  for (int r = 0; r < M; ++r) ycur[r] = yp[r * a.y_si];

  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see cdkf_filter_reg_body.inc
  for (long k = 0; k < a.T; ++k) {
    // prefetch of y_{k+1}, t_{k+2} a whole step ahead (cdkf_filter_reg_body.inc)
    if (k + 1 < a.T) yp += a.y_sk;
    if (k + 2 < a.T) tp += a.t_sk;
    R ynext[M];
#pragma unroll
    for (int r = 0; r < M; ++r) ynext[r] = yp[r * a.y_si];
    const R tnn = tp[0];
    if constexpr (DriftInputs<Drift>::value) a.drift.load_inputs(a, n, k);
    if constexpr (UKF && D == 3) {  // leading principal minors of the predicted covariance
      const R p00 = ys[D + sidx<D>(0, 0)], p01 = ys[D + sidx<D>(0, 1)], p02 = ys[D + sidx<D>(0, 2)], p11 = ys[D + sidx<D>(1, 1)],
              p12 = ys[D + sidx<D>(1, 2)], p22 = ys[D + sidx<D>(2, 2)];
      const R m2 = p00 * p11 - p01 * p01;
      const R m3 = p22 * m2 - p12 * (p00 * p12 - p01 * p02) + p02 * (p01 * p12 - p11 * p02);
      if (!(p00 > R(0)) || !(m2 > R(0)) || !(m3 > R(0))) {
        st |= kStatusNotPd;
        ys[0] = R(0) / R(0);
      }
    }
    ekf_update_sens<R, D, M>(a, ys, ycur, ll, g, st);
    if (ys[0] != ys[0]) st |= kStatusNan;
    const R t1 = (k + 1 < a.T) ? tnext_obs : tcur + a.dt_final;
    // the last predict (over dt_final) does not enter the log-likelihood: skip it
    if (k + 1 < a.T) {
      const bool capped = integrate<R, 2 * NS, 0, NS>(ys, tcur, t1, a.dt0, a.max_steps, rhs, C);
      if (capped) st |= kStatusMaxSteps;
    }
    tcur = tnext_obs;
    tnext_obs = tnn;
#pragma unroll
    for (int r = 0; r < M; ++r) ycur[r] = ynext[r];
  }
  ll.flush();
  if (live) {
    ga.grad[gid] = (R)g;
    if (p == 0) {
      a.ll[n] = (R)ll.ll;
      if (a.status) a.status[n] = st;
    }
  }
}
template <typename R, int D, int M, typename Drift, bool GENERIC = false, bool UKF = false>
__global__ __launch_bounds__(64, 1) void ekf_grad_reg_kernel(const GradArgs<R, D, M, Drift> ga) {
  ekf_grad_reg_body<R, D, M, Drift, GENERIC, UKF>(ga);
}

}  // namespace cdkf
