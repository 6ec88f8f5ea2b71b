// cdkf_reg_kernels.h -- lane-per-trajectory ("reg") sweep kernels for small state dimensions.
//
// Mapping: one LANE owns one trajectory for the whole time scan; mean, packed
// symmetric covariance and the six Dormand-Prince slopes live in that lane's VGPRs, model
// parameters arrive as kernel arguments (SGPRs).  At d <= 4 this beats a wave-per-trajectory
// mapping: the symmetric/sparse arithmetic is minimal, there is no cross-lane traffic, and a wave
// issues a 64-lane fp64 instruction in the same 4 cycles whether 9 or 64 lanes carry work.
//
// Reference functions restated (paths relative to /root/reference/src/continuous_discrete_nonlinear_gaussian_ssm):
//   extended_kalman_filter  inference_ekf.py:202-326     _predict :46-148     _condition_on :153-199
//   extended_kalman_smoother inference_ekf.py:450-539    _smooth  :363-448
//   unscented_kalman_filter inference_ukf.py:206-308     _predict :93-159     _condition_on :162-203
#pragma once
#include "cdkf_drifts.h"

namespace cdkf {

constexpr int kStatusNotPd = 1, kStatusNan = 2, kStatusMaxSteps = 4;

template <typename R, int D, int M, typename Drift>
struct RegArgs {
  static constexpr int NP = Dims<D>::NP;
  Drift drift;
  R LQL[NP];   // packed upper triangle of L Qc L^T
  R LQLz[NP];  // zeroth order: (L c) Qc (L c)^T, c = cov_rescaling
  R H[M][D];
  R hb[M];
  R Rm[M][M];
  R m0[D];
  R P0[NP];
  R dt0, dt_final;
  // UKF weights (inference_ukf.py:63-89): sigma scale sqrt(n + lambda), w_mean[0], w_cov[0], w_i
  R ukf_c, ukf_wm0, ukf_wc0, ukf_wi;
  long max_steps;
  int order, num_iter, forecast;
  int solver;   // CDKF_SOLVER_*
  int lanes;      // distinct trajectories per wavefront (power of two <= 64; the other lanes repeat them)
  int xcd_shift;  // log2 of the wavefront groups that share a 128-byte line (reg_unit_index)
  RkTab<R> rk;  // used by the GENERIC instantiations only (solver != CDKF_SOLVER_DOPRI5)
  long N, T;
  // element (n, k, i) of an array lives at  n * sn + k * sk + i.  Reference layout [N,T,w]:
  // (sn, sk) = (T*w, w); time-major layout [T,N,w]: (sn, sk) = (w, N*w); shared t: sn = 0.
  // component stride si: 1 for the array-of-structures layouts, N for CDKF_LAYOUT_TCN ([T,w,N]).
  long t_sn, t_sk, y_sn, y_sk, y_si, m_sn, m_sk, m_si, P_sn, P_sk, P_si;
  const R* t;
  const R* y;
  R* ll;
  R* fm;
  R* fP;
  R* pm;
  R* pP;
  int* status;
  // inputs u[n][k][0 .. d_u-1] (element (n, k, i) at n * u_sn + k * u_sk + i * u_si, laid out like y): read only by drifts / emissions
  // given as source that name them (DriftInputs); null otherwise
  const R* u;
  long u_sn, u_sk, u_si;
};

// Unit (trajectory; (trajectory, parameter) pair in the gradient sweep) of this lane.  `lanes` consecutive units per
// wavefront, repeated over its 64 lanes (cdkf_filter_reg_body.inc).  Below 16 units a wavefront touches only part of each
// 128-byte line of the [T,comp,N] arrays, and workgroups are dealt to the 8 XCDs round-robin: the 2^xcd_shift groups that
// share a line are renumbered onto ONE XCD (blocks b, b + 8, ...), whose L2 then fetches the line once and writes it whole.
CDKF_DEV long reg_unit_index(int lanes, int xcd_shift) {
  const long b = blockIdx.x;
  const int sh = xcd_shift;
  const long grp = ((b >> (3 + sh)) << (3 + sh)) + ((b & 7) << sh) + ((b >> 3) & ((1 << sh) - 1));
  return grp * lanes + (threadIdx.x & (lanes - 1));
}
// true for the surplus wavefronts of the rounded-up grid (their whole group lies beyond the batch): they leave at once
CDKF_DEV bool reg_group_is_surplus(long unit, int lanes, long units) { return unit - (threadIdx.x & (lanes - 1)) >= units; }

// tableau the sweep integrates with: the VGPR-pinned Dormand-Prince constants, or the run-time tableau of the arguments
template <typename R, bool GENERIC>
struct TabSel {
  template <typename Args>
  static CDKF_DEV Dp5V<R> get(const Args&) {
    Dp5V<R> c;
    c.init();
    return c;
  }
};
template <typename R>
struct TabSel<R, true> {
  template <typename Args>
  static CDKF_DEV const RkTab<R>& get(const Args& a) {
    return a.rk;
  }
};

// ---- EKF moment ODE right-hand side (inference_ekf.py:76-123) --------------------------------
template <typename R, int D, typename Drift>
struct EkfRhs {
  static constexpr int NS = Dims<D>::NS;
  const Drift& drift;
  const R* LQL;
  int order;
  static constexpr bool kTime = DriftTime<Drift>::value;
  CDKF_DEV void set_time(R t) const {
    if constexpr (kTime) drift.set_time(t);
  }
  CDKF_DEV void operator()(const R (&y)[NS], R (&dy)[NS]) const {
    R F[D][D], f[D];
    drift.f(y, f);
    drift.jac(y, F);
    R A[D][D];  // A = F P
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) {
        R acc = R(0);
        bool first = true;
#pragma unroll
        for (int k = 0; k < D; ++k)
          if (Drift::nz(i, k)) {
            R p = y[D + sidx<D>(k, j)];
            acc = first ? F[i][k] * p : rfma(F[i][k], p, acc);
            first = false;
          }
        A[i][j] = acc;
      }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = i; j < D; ++j) dy[D + sidx<D>(i, j)] = (A[i][j] + A[j][i]) + LQL[sidx<D>(i, j)];
#pragma unroll
    for (int i = 0; i < D; ++i) dy[i] = f[i];
    if (Drift::HAS_G && order == 2) {
      R g[D];
      drift.divgrad(y, g);
#pragma unroll
      for (int l = 0; l < D; ++l) {
        R s = R(0);
#pragma unroll
        for (int k = 0; k < D; ++k) s = rfma(g[k], y[D + sidx<D>(k, l)], s);
        dy[l] = rfma(R(0.5), s, dy[l]);
      }
    }
  }
};

// mean-only ODE for state_order == 'zeroth' (inference_ekf.py:97-99,126-138)
template <typename R, int D, typename Drift>
struct MeanRhs {
  const Drift& drift;
  static constexpr bool kTime = DriftTime<Drift>::value;
  CDKF_DEV void set_time(R t) const {
    if constexpr (kTime) drift.set_time(t);
  }
  CDKF_DEV void operator()(const R (&y)[D], R (&dy)[D]) const { drift.f(y, dy); }
};

// ---- log-likelihood accumulator ------------------------------------------------------------------
// sum_k [-0.5 q_k - sum_i log L_ii - 0.5 M log 2pi] with sum_i log L_ii = -log prod_i (1 / L_ii): the
// reciprocal pivots are multiplied into a running product and ONE fp64 log is taken every 8 steps (or
// earlier if the product leaves [1e-150, 1e150]) instead of M logs per step.
struct LlAcc {
  double ll = 0.0, prod = 1.0;
  int cnt = 0;
  CDKF_DEV void add(double q, double pinv, int M) {
    ll += -0.5 * q - 0.5 * M * 1.8378770664093454835606594728112;
    prod *= pinv;
    ++cnt;
    const double ap = fabs(prod);
    if (cnt == 8 || !(ap > 1e-150 && ap < 1e150)) flush();
  }
  CDKF_DEV void flush() {
    ll += log(prod);
    prod = 1.0;
    cnt = 0;
  }
};

// ---- EKF measurement update + log-likelihood term ---------------------------------------------
// ll term: MVN(H m + b, H P H^T + R).log_prob(y)                       inference_ekf.py:285-286
// update : S = R + H P H^T; K = psd_solve(S, H P)^T; P+ = P - K S K^T; m+ = m + K (y - h(m)),
//          repeated num_iter times, then symmetrize                     inference_ekf.py:183-199
// HSEL: the emission picks the first M state coordinates (H = I[:M], bias = 0 -- the tutorials' H = I_3 and
// H = [1,0,0]); then H P = P[:M,:], H P H^T = P[:M,:M] and the products with H disappear.
template <typename R, int D, int M, bool HSEL, typename Args>
CDKF_DEV void ekf_update(const Args& a, R (&ys)[Dims<D>::NS], const R (&yobs)[M], LlAcc& ll, int& st) {
  bool bad = false;
  for (int it = 0; it < a.num_iter; ++it) {
    R HP[M][D], S[M][M], v[M];
    if constexpr (HSEL) {
#pragma unroll
      for (int r = 0; r < M; ++r) {
#pragma unroll
        for (int j = 0; j < D; ++j) HP[r][j] = ys[D + sidx<D>(r, j)];
#pragma unroll
        for (int c = 0; c < M; ++c) S[r][c] = ys[D + sidx<D>(r, c)] + a.Rm[r][c];
        v[r] = yobs[r] - ys[r];
      }
    } else {
#pragma unroll
      for (int r = 0; r < M; ++r)
#pragma unroll
        for (int j = 0; j < D; ++j) {
          R s = a.H[r][0] * ys[D + sidx<D>(0, j)];
#pragma unroll
          for (int k = 1; k < D; ++k) s = rfma(a.H[r][k], ys[D + sidx<D>(k, j)], s);
          HP[r][j] = s;
        }
#pragma unroll
      for (int r = 0; r < M; ++r)
#pragma unroll
        for (int c = 0; c < M; ++c) {
          R s = HP[r][0] * a.H[c][0];
#pragma unroll
          for (int k = 1; k < D; ++k) s = rfma(HP[r][k], a.H[c][k], s);
          S[r][c] = s + a.Rm[r][c];
        }
#pragma unroll
      for (int r = 0; r < M; ++r) {
        R s = a.H[r][0] * ys[0];
#pragma unroll
        for (int k = 1; k < D; ++k) s = rfma(a.H[r][k], ys[k], s);
        v[r] = yobs[r] - (s + a.hb[r]);
      }
    }
    R q = R(0), pinv = R(1);
    if (it == 0) {
      // TFP log_prob: Cholesky of S as given (lower triangle, no jitter).  The accumulator call (a data-dependent branch
      // around the occasional log) comes after the update so that this factorisation and psd_solve's -- two independent
      // chains of dependent fp64 operations -- share one basic block and are interleaved by the scheduler.
      R Lc[M][M], inv[M];
      chol_lower<R, M>(S, Lc, inv, bad);
      R z[M];
#pragma unroll
      for (int i = 0; i < M; ++i) {
        R w = v[i];
#pragma unroll
        for (int k = 0; k < i; ++k) w = rfma(-Lc[i][k], z[k], w);
        z[i] = w * inv[i];
        q = rfma(z[i], z[i], q);
        pinv *= inv[i];
      }
    }
    // psd_solve: symmetrize + 1e-9 I, Cholesky, cho_solve           dynamax/utils/utils.py:202-207
    R Sb[M][M];
#pragma unroll
    for (int r = 0; r < M; ++r)
#pragma unroll
      for (int c = 0; c <= r; ++c) {
        R s = R(0.5) * (S[r][c] + S[c][r]);
        if (r == c) s += R(1e-9);
        Sb[r][c] = s;
      }
    R Lb[M][M], invb[M];
    chol_lower<R, M>(Sb, Lb, invb, bad);
    R X[M][D];  // X = Sb^{-1} (H P);  K = X^T
#pragma unroll
    for (int r = 0; r < M; ++r)
#pragma unroll
      for (int j = 0; j < D; ++j) X[r][j] = HP[r][j];
    chol_solve<R, M, D>(Lb, invb, X);
    R SX[M][D];  // S X  (so that K S K^T = X^T (S X))
#pragma unroll
    for (int r = 0; r < M; ++r)
#pragma unroll
      for (int j = 0; j < D; ++j) {
        R s = S[r][0] * X[0][j];
#pragma unroll
        for (int c = 1; c < M; ++c) s = rfma(S[r][c], X[c][j], s);
        SX[r][j] = s;
      }
    // P+ = P - K S K^T, then symmetrize (applied every iteration here; the reference symmetrizes once
    // after the loop -- identical for num_iter == 1, rounding-level otherwise)
    R Pn[Dims<D>::NP];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = i; j < D; ++j) {
        R tij = X[0][i] * SX[0][j];
#pragma unroll
        for (int c = 1; c < M; ++c) tij = rfma(X[c][i], SX[c][j], tij);
        const R p = ys[D + sidx<D>(i, j)];
        if (i == j) {
          Pn[sidx<D>(i, j)] = p - tij;
        } else {
          R tji = X[0][j] * SX[0][i];
#pragma unroll
          for (int c = 1; c < M; ++c) tji = rfma(X[c][j], SX[c][i], tji);
          Pn[sidx<D>(i, j)] = R(0.5) * ((p - tij) + (p - tji));
        }
      }
#pragma unroll
    for (int i = 0; i < D; ++i) {
      R s = ys[i];
#pragma unroll
      for (int r = 0; r < M; ++r) s = rfma(X[r][i], v[r], s);
      ys[i] = s;
    }
#pragma unroll
    for (int e = 0; e < Dims<D>::NP; ++e) ys[D + e] = Pn[e];
    if (it == 0) {
      CDKF_OPAQUE("+v"(q), "+v"(pinv) : "v"(ys[D]), "v"(ys[0]));  // pin the accumulator call below the update (see above)
      ll.add((double)q, (double)pinv, M);
    }
  }
  if (bad) st |= kStatusNotPd;
}

// store helpers: mean [D] and full symmetric covariance [D,D] in the reference layout
template <typename R, int D>
CDKF_DEV void store_moments(R* __restrict__ mean_out, R* __restrict__ cov_out, long moff, long poff, long m_si,
                            long P_si, const R (&ys)[Dims<D>::NS]) {
  if (mean_out) {
    R* p = mean_out + moff;
#pragma unroll
    for (int i = 0; i < D; ++i) {
      *p = ys[i];
      p += m_si;
    }
  }
  if (cov_out) {
    R* p = cov_out + poff;
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) {
        *p = ys[D + sidx<D>(i, j)];
        p += P_si;
      }
  }
}

template <typename R, int D>
CDKF_DEV void store_moments_all(R* __restrict__ mean_out, R* __restrict__ cov_out, long moff, long poff, long m_si,
                                long P_si, const R (&ys)[Dims<D>::NS]) {
  R* p = mean_out + moff;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    *p = ys[i];
    p += m_si;
  }
  R* q = cov_out + poff;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      *q = ys[D + sidx<D>(i, j)];
      q += P_si;
    }
}

// ---- unpack a full symmetric D x D matrix from the packed state (for Cholesky: lower triangle) ----
template <typename R, int D>
CDKF_DEV void unpack_sym(const R* packed, R (&A)[D][D]) {
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) A[i][j] = packed[sidx<D>(i, j)];
}

// ---- UKF -----------------------------------------------------------------------------------------
// Sigma points (inference_ukf.py:45-60): X_0 = m, X_{i+-} = m +- o_i with o_i = c * chol(P)[:, i] (lower
// triangular: o_i[j] = 0 for j < i).  The reference's weighted sums are evaluated in antisymmetric form:
// with xbar = sum_s w_m[s] X_s = m (the weights sum to one) the pair (i+, i-) contributes
//   (g(X_i+) - gbar) o_i^T + (g(X_i-) - gbar) (-o_i)^T = (g(X_i+) - g(X_i-)) o_i^T,
// and the s = 0 term carries the factor (m - xbar) = 0.  Same numbers as the literal
// fX^T W X / tensordot forms up to rounding, at a third of the arithmetic.
template <typename R, int D>
CDKF_DEV void ukf_offsets(const R (&ys)[Dims<D>::NS], R c, R (&o)[D][D], bool& bad) {
  R P[D][D], Lc[D][D], inv[D];
  unpack_sym<R, D>(ys + D, P);
  chol_lower<R, D>(P, Lc, inv, bad);
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) o[i][j] = (j >= i) ? c * Lc[j][i] : R(0);
}

// UKF moment ODE (inference_ukf.py:128-152):  dm = fX^T w_m,  dP = fX^T W X + (.)^T + L Qc L^T.
template <typename R, int D, typename Args>
struct UkfRhs {
  static constexpr int NS = Dims<D>::NS;
  const Args& a;
  bool* bad;
  static constexpr bool kTime = DriftTime<decltype(Args::drift)>::value;  // (every sigma point's drift at the stage time, inference_ukf.py:142)
  CDKF_DEV void set_time(R t) const {
    if constexpr (kTime) a.drift.set_time(t);
  }
  CDKF_DEV void operator()(const R (&y)[NS], R (&dy)[NS]) const {
    R o[D][D];
    ukf_offsets<R, D>(y, a.ukf_c, o, *bad);
    R f0[D], sum[D], df[D][D];
    a.drift.f(y, f0);
#pragma unroll
    for (int j = 0; j < D; ++j) sum[j] = R(0);
#pragma unroll
    for (int i = 0; i < D; ++i) {
      R xp[D], xm[D], fp[D], fm[D];
#pragma unroll
      for (int j = 0; j < D; ++j) {
        xp[j] = (j >= i) ? y[j] + o[i][j] : y[j];
        xm[j] = (j >= i) ? y[j] - o[i][j] : y[j];
      }
      a.drift.f(xp, fp);
      a.drift.f(xm, fm);
#pragma unroll
      for (int j = 0; j < D; ++j) {
        sum[j] += fp[j] + fm[j];
        df[i][j] = fp[j] - fm[j];
      }
    }
#pragma unroll
    for (int j = 0; j < D; ++j) dy[j] = rfma(a.ukf_wm0, f0[j], a.ukf_wi * sum[j]);
    R foo[D][D];  // foo[r][b] = w_i sum_{i <= b} df_i[r] o_i[b]
#pragma unroll
    for (int r = 0; r < D; ++r)
#pragma unroll
      for (int b = 0; b < D; ++b) {
        R acc = df[0][r] * o[0][b];
#pragma unroll
        for (int i = 1; i <= b; ++i) acc = rfma(df[i][r], o[i][b], acc);
        foo[r][b] = a.ukf_wi * acc;
      }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = i; j < D; ++j) dy[D + sidx<D>(i, j)] = (foo[i][j] + foo[j][i]) + a.LQL[sidx<D>(i, j)];
  }
};

// Shared tail of the unscented update: log-likelihood term MVN(ybar, S).log_prob(y), gain K = psd_solve(S, C^T)^T,
// m+ = m + K v, P+ = P - K S K^T  (inference_ukf.py:196-203).  v = y - ybar, Sm = S, C = cross-covariance [D][M].
template <typename R, int D, int M>
CDKF_DEV void ukf_finish(R (&ys)[Dims<D>::NS], const R (&v)[M], const R (&Sm)[M][M], const R (&C)[D][M], LlAcc& ll, int& st,
                         bool bad) {
  {
    R Lc[M][M], inv[M];
    chol_lower<R, M>(Sm, Lc, inv, bad);
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
  }
  R Sb[M][M];
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      R s = R(0.5) * (Sm[r][c] + Sm[c][r]);
      if (r == c) s += R(1e-9);
      Sb[r][c] = s;
    }
  R Lb[M][M], invb[M];
  chol_lower<R, M>(Sb, Lb, invb, bad);
  R Xs[M][D];  // Sb^{-1} C^T ;  K = Xs^T
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int j = 0; j < D; ++j) Xs[r][j] = C[j][r];
  chol_solve<R, M, D>(Lb, invb, Xs);
  R SX[M][D];
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      R s = Sm[r][0] * Xs[0][j];
#pragma unroll
      for (int c = 1; c < M; ++c) s = rfma(Sm[r][c], Xs[c][j], s);
      SX[r][j] = s;
    }
  // P+ = P - K S K^T (no symmetrize in the UKF); the packed entry (i,j), i<=j, takes the LOWER element (j,i)
  // of the reference's result, which is what the next Cholesky reads.
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = i; j < D; ++j) {
      R tji = Xs[0][j] * SX[0][i];
#pragma unroll
      for (int c = 1; c < M; ++c) tji = rfma(Xs[c][j], SX[c][i], tji);
      ys[D + sidx<D>(i, j)] -= tji;
    }
#pragma unroll
  for (int i = 0; i < D; ++i) {
    R s = ys[i];
#pragma unroll
    for (int r = 0; r < M; ++r) s = rfma(Xs[r][i], v[r], s);
    ys[i] = s;
  }
  if (bad) st |= kStatusNotPd;
}

// UKF measurement update (inference_ukf.py:162-203) for the linear emission h(x) = H x + b:
// ybar = H m + b,  dY_i = H o_i,  S = 2 w_i sum_i dY_i dY_i^T + R,  C = 2 w_i sum_i o_i dY_i^T.
template <typename R, int D, int M, typename Args>
CDKF_DEV void ukf_update(const Args& a, R (&ys)[Dims<D>::NS], const R (&yobs)[M], LlAcc& ll, int& st) {
  bool bad = false;
  R o[D][D];
  ukf_offsets<R, D>(ys, a.ukf_c, o, bad);
  R dY[D][M];
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int r = 0; r < M; ++r) {
      R v = R(0);
#pragma unroll
      for (int k = i; k < D; ++k) v = rfma(a.H[r][k], o[i][k], v);
      dY[i][r] = v;
    }
  R v[M];
#pragma unroll
  for (int r = 0; r < M; ++r) {
    R s = a.H[r][0] * ys[0];
#pragma unroll
    for (int k = 1; k < D; ++k) s = rfma(a.H[r][k], ys[k], s);
    v[r] = yobs[r] - (s + a.hb[r]);
  }
  const R w2 = a.ukf_wi + a.ukf_wi;
  R Sm[M][M], C[D][M];
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int c = 0; c < M; ++c) {
      R acc = dY[0][r] * dY[0][c];
#pragma unroll
      for (int i = 1; i < D; ++i) acc = rfma(dY[i][r], dY[i][c], acc);
      Sm[r][c] = rfma(w2, acc, a.Rm[r][c]);
    }
#pragma unroll
  for (int k = 0; k < D; ++k)
#pragma unroll
    for (int c = 0; c < M; ++c) {
      R acc = o[0][k] * dY[0][c];
#pragma unroll
      for (int i = 1; i <= k; ++i) acc = rfma(o[i][k], dY[i][c], acc);
      C[k][c] = w2 * acc;
    }
  ukf_finish<R, D, M>(ys, v, Sm, C, ll, st, bad);
}

// ---- user-supplied emission functions (run-time compiled kernels, launch_custom.hip) -----------------------------------
// The reference accepts any callable h and linearises it with jacfwd (inference_ekf.py:258-259, 183-199) or pushes the
// sigma points through it (inference_ukf.py:162-203).  An emission type provides  h(x, hx[M])  and  jac(x, H[M][D]).
struct EmisLinearTag {  // the built-in kernels: linear emission from the argument block, code paths above
  static constexpr bool kCustom = false;
};

template <typename R, int D, int M, typename Emis, typename Args>
CDKF_DEV void ekf_update_custom(const Args& a, const Emis& em, R (&ys)[Dims<D>::NS], const R (&yobs)[M], LlAcc& ll, int& st) {
  bool bad = false;
  for (int it = 0; it < a.num_iter; ++it) {
    R H[M][D], hx[M];
    em.jac(ys, H);  // re-linearised at the current mean in every iteration
    em.h(ys, hx);
    R HP[M][D], S[M][M], v[M];
#pragma unroll
    for (int r = 0; r < M; ++r)
#pragma unroll
      for (int j = 0; j < D; ++j) {
        R s = H[r][0] * ys[D + sidx<D>(0, j)];
#pragma unroll
        for (int k = 1; k < D; ++k) s = rfma(H[r][k], ys[D + sidx<D>(k, j)], s);
        HP[r][j] = s;
      }
#pragma unroll
    for (int r = 0; r < M; ++r) {
#pragma unroll
      for (int c = 0; c < M; ++c) {
        R s = HP[r][0] * H[c][0];
#pragma unroll
        for (int k = 1; k < D; ++k) s = rfma(HP[r][k], H[c][k], s);
        S[r][c] = s + a.Rm[r][c];
      }
      v[r] = yobs[r] - hx[r];
    }
    if (it == 0) {
      R Lc[M][M], inv[M];
      chol_lower<R, M>(S, Lc, inv, bad);
      R q = R(0), pinv = R(1);
      R z[M];
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
    }
    R Sb[M][M];
#pragma unroll
    for (int r = 0; r < M; ++r)
#pragma unroll
      for (int c = 0; c <= r; ++c) {
        R s = R(0.5) * (S[r][c] + S[c][r]);
        if (r == c) s += R(1e-9);
        Sb[r][c] = s;
      }
    R Lb[M][M], invb[M];
    chol_lower<R, M>(Sb, Lb, invb, bad);
    R X[M][D];
#pragma unroll
    for (int r = 0; r < M; ++r)
#pragma unroll
      for (int j = 0; j < D; ++j) X[r][j] = HP[r][j];
    chol_solve<R, M, D>(Lb, invb, X);
    R SX[M][D];
#pragma unroll
    for (int r = 0; r < M; ++r)
#pragma unroll
      for (int j = 0; j < D; ++j) {
        R s = S[r][0] * X[0][j];
#pragma unroll
        for (int c = 1; c < M; ++c) s = rfma(S[r][c], X[c][j], s);
        SX[r][j] = s;
      }
    R Pn[Dims<D>::NP];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = i; j < D; ++j) {
        R tij = X[0][i] * SX[0][j], tji = X[0][j] * SX[0][i];
#pragma unroll
        for (int c = 1; c < M; ++c) {
          tij = rfma(X[c][i], SX[c][j], tij);
          tji = rfma(X[c][j], SX[c][i], tji);
        }
        const R p = ys[D + sidx<D>(i, j)];
        Pn[sidx<D>(i, j)] = (i == j) ? p - tij : R(0.5) * ((p - tij) + (p - tji));
      }
#pragma unroll
    for (int i = 0; i < D; ++i) {
      R s = ys[i];
#pragma unroll
      for (int r = 0; r < M; ++r) s = rfma(X[r][i], v[r], s);
      ys[i] = s;
    }
#pragma unroll
    for (int e = 0; e < Dims<D>::NP; ++e) ys[D + e] = Pn[e];
  }
  if (bad) st |= kStatusNotPd;
}

// unscented update through a general emission: Y_s = h(X_s), ybar = sum w_m Y, S = sum w_c (Y - ybar)(Y - ybar)^T + R,
// C = sum w_c (X - m)(Y - ybar)^T  (the s = 0 term of C vanishes; the pairs share the offset o_i)
template <typename R, int D, int M, typename Emis, typename Args>
CDKF_DEV void ukf_update_custom(const Args& a, const Emis& em, R (&ys)[Dims<D>::NS], const R (&yobs)[M], LlAcc& ll, int& st) {
  bool bad = false;
  R o[D][D];
  ukf_offsets<R, D>(ys, a.ukf_c, o, bad);
  R Y0[M], Yp[D][M], Ym[D][M], ybar[M];
  em.h(ys, Y0);
#pragma unroll
  for (int r = 0; r < M; ++r) ybar[r] = R(0);
#pragma unroll
  for (int i = 0; i < D; ++i) {
    R xp[D], xm[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      xp[j] = (j >= i) ? ys[j] + o[i][j] : ys[j];
      xm[j] = (j >= i) ? ys[j] - o[i][j] : ys[j];
    }
    em.h(xp, Yp[i]);
    em.h(xm, Ym[i]);
#pragma unroll
    for (int r = 0; r < M; ++r) ybar[r] += Yp[i][r] + Ym[i][r];
  }
  R v[M], Sm[M][M], C[D][M];
#pragma unroll
  for (int r = 0; r < M; ++r) {
    ybar[r] = rfma(a.ukf_wm0, Y0[r], a.ukf_wi * ybar[r]);
    v[r] = yobs[r] - ybar[r];
  }
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int c = 0; c < M; ++c) {
      R acc = R(0);
#pragma unroll
      for (int i = 0; i < D; ++i) {
        acc = rfma(Yp[i][r] - ybar[r], Yp[i][c] - ybar[c], acc);
        acc = rfma(Ym[i][r] - ybar[r], Ym[i][c] - ybar[c], acc);
      }
      Sm[r][c] = rfma(a.ukf_wc0 * (Y0[r] - ybar[r]), Y0[c] - ybar[c], rfma(a.ukf_wi, acc, a.Rm[r][c]));
    }
#pragma unroll
  for (int k = 0; k < D; ++k)
#pragma unroll
    for (int c = 0; c < M; ++c) {
      R acc = R(0);
#pragma unroll
      for (int i = 0; i <= k; ++i) acc = rfma(o[i][k], Yp[i][c] - Ym[i][c], acc);
      C[k][c] = a.ukf_wi * acc;
    }
  ukf_finish<R, D, M>(ys, v, Sm, C, ll, st, bad);
}

// ---- filter sweep (EKF and UKF) -------------------------------------------------------------------
// UKF:    unscented filter (inference_ukf.py:206-308) instead of the extended one (inference_ekf.py:202-326);
// ZEROTH: EKF state_order == 'zeroth' (mean-only ODE + sqrt(dt) L Qc L^T, inference_ekf.py:126-138);
// HSEL:   emission = first M coordinates (see ekf_update);
// OUT:    kOutNone (log-likelihood only), kOutAll (all four moment arrays, stored UNCONDITIONALLY) or
//         kOutSome (each pointer checked at run time).  The unconditional form matters: gfx950 has one
//         in-order vmcnt for loads and stores, and with stores under a branch the compiler must assume
//         none were issued, so every wait for a prefetched observation degenerates to "drain all
//         stores" (~300 cycles twice per step).  Idle lanes of the last wavefront shadow trajectory
//         N-1 and store the same values to the same addresses, so no lane predicate is needed either.
constexpr int kOutNone = 0, kOutAll = 1, kOutSome = 2;

template <typename R, int D, int M, typename Drift, bool UKF, bool ZEROTH, bool HSEL, int OUT, bool FORECAST = false,
          bool GENERIC = false, typename Emis = EmisLinearTag>
__global__ __launch_bounds__(64, 1) void filter_reg_kernel(const RegArgs<R, D, M, Drift> a) {
#include "cdkf_filter_reg_body.inc"
}

// the same sweep as a device function, for the run-time compiled kernels of user-supplied drifts (launch_custom.hip)
template <typename R, int D, int M, typename Drift, bool UKF, bool ZEROTH, bool HSEL, int OUT, bool FORECAST = false,
          bool GENERIC = false, typename Emis = EmisLinearTag>
CDKF_DEV void filter_reg_body(const RegArgs<R, D, M, Drift>& a) {
#include "cdkf_filter_reg_body.inc"
}

// ---- EKF (RTS) smoother backward sweep (inference_ekf.py:363-448, 503-531) ----------------------
// Over the interval [t_k, t_{k+1}] the filtered moments (m_f, P_f) at t_k are constants, so
// G = F(m_f) + psd_solve(P_f, L Qc L^T)^T and f(m_f) are hoisted out of the RK stages (the reference
// re-factorises P_f in every stage and gets the same numbers).
template <typename R, int D>
struct SmoothRhs {
  static constexpr int NS = Dims<D>::NS;
  R G[D][D];
  R fmf[D];
  R mf[D];
  const R* LQL;
  // reverse-time: the reference integrates -rhs(t1 - s) over s in [0, t1 - t0] (diffrax_utils.py:13-25)
  CDKF_DEV void operator()(const R (&y)[NS], R (&dy)[NS]) const {
    R dm[D];
#pragma unroll
    for (int i = 0; i < D; ++i) dm[i] = y[i] - mf[i];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      R s = G[i][0] * dm[0];
#pragma unroll
      for (int k = 1; k < D; ++k) s = rfma(G[i][k], dm[k], s);
      dy[i] = -(fmf[i] + s);
    }
    R A[D][D];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) {
        R s = G[i][0] * y[D + sidx<D>(0, j)];
#pragma unroll
        for (int k = 1; k < D; ++k) s = rfma(G[i][k], y[D + sidx<D>(k, j)], s);
        A[i][j] = s;
      }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = i; j < D; ++j) dy[D + sidx<D>(i, j)] = -((A[i][j] + A[j][i]) - LQL[sidx<D>(i, j)]);
  }
};

// ... for a drift that depends on time the hoisting is not available: f(m_f, u, t) and jacfwd(f)(m_f, u, t) are evaluated in every
// stage at t = t1 - s (inference_ekf.py:433-438 under reverse_rhs, diffrax_utils.py:13-25); aux = psd_solve(P_f, L Qc L^T)^T stays
template <typename R, int D, typename Drift>
struct SmoothRhsT {
  static constexpr int NS = Dims<D>::NS;
  static constexpr bool kTime = true;
  const Drift& drift;
  R aux[D][D];
  R mf[D];
  R tend;
  const R* LQL;
  CDKF_DEV void set_time(R s) const { drift.set_time(tend - s); }
  CDKF_DEV void operator()(const R (&y)[NS], R (&dy)[NS]) const {
    R G[D][D], fmf[D], dm[D];
    drift.f(mf, fmf);
    drift.jac(mf, G);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) G[i][j] += aux[i][j];
#pragma unroll
    for (int i = 0; i < D; ++i) dm[i] = y[i] - mf[i];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      R s = G[i][0] * dm[0];
#pragma unroll
      for (int k = 1; k < D; ++k) s = rfma(G[i][k], dm[k], s);
      dy[i] = -(fmf[i] + s);
    }
    R A[D][D];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) {
        R s = G[i][0] * y[D + sidx<D>(0, j)];
#pragma unroll
        for (int k = 1; k < D; ++k) s = rfma(G[i][k], y[D + sidx<D>(k, j)], s);
        A[i][j] = s;
      }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = i; j < D; ++j) dy[D + sidx<D>(i, j)] = -((A[i][j] + A[j][i]) - LQL[sidx<D>(i, j)]);
  }
};

template <typename R, int D, int M, typename Drift, bool GENERIC = false>
CDKF_DEV void ekf_smoother_reg_body(const RegArgs<R, D, M, Drift>& a, R* __restrict__ sm, R* __restrict__ sP) {
  constexpr int NS = Dims<D>::NS;
  const long gid = reg_unit_index(a.lanes, a.xcd_shift);
  if (reg_group_is_surplus(gid, a.lanes, a.N)) return;
  const bool live = gid < a.N;
  const long n = live ? gid : a.N - 1;
  const R* __restrict__ tp = a.t + n * a.t_sn;
  const R* __restrict__ fm = a.fm + n * a.m_sn;
  const R* __restrict__ fP = a.fP + n * a.P_sn;
  int st = 0;
  bool bad = false;
  const auto C = TabSel<R, GENERIC>::get(a);

  constexpr int NP = Dims<D>::NP;
  // idle / repeating lanes shadow a live trajectory and store the same values: the stores are unconditional (exact vmcnt
  // accounting, see filter_reg_kernel) and the outputs are never NULL here
  R ys[NS];  // smoothed moments at t_{k+1}
  {
    const long k = a.T - 1;
#pragma unroll
    for (int i = 0; i < D; ++i) ys[i] = fm[k * a.m_sk + i * a.m_si];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = i; j < D; ++j) ys[D + sidx<D>(i, j)] = fP[k * a.P_sk + (i * D + j) * a.P_si];
    store_moments_all<R, D>(sm, sP, n * a.m_sn + k * a.m_sk, n * a.P_sn + k * a.P_sk, a.m_si, a.P_si, ys);
  }
  R t1 = tp[(a.T - 1) * a.t_sk];
  // filtered moments of step k are loaded one step ahead (during the integration of step k + 1): every step reads fresh
  // lines, and used on arrival they exposed a full memory round trip per step
  R mf_n[D], Pf_n[NP], t0_n;
  {
    const long k = a.T >= 2 ? a.T - 2 : 0;
#pragma unroll
    for (int i = 0; i < D; ++i) mf_n[i] = fm[k * a.m_sk + i * a.m_si];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = i; j < D; ++j) Pf_n[sidx<D>(i, j)] = fP[k * a.P_sk + (i * D + j) * a.P_si];
    t0_n = tp[k * a.t_sk];
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) before the loop: see cdkf_filter_reg_body.inc
  for (long k = a.T - 2; k >= 0; --k) {
    SmoothRhs<R, D> rhs;
    rhs.LQL = a.LQL;
    R Pf[D][D];
#pragma unroll
    for (int i = 0; i < D; ++i) rhs.mf[i] = mf_n[i];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) Pf[i][j] = Pf_n[sidx<D>(i, j)];  // the filter stored a symmetric matrix
    const R t0 = t0_n;
    {
      const long kn = k >= 1 ? k - 1 : 0;
#pragma unroll
      for (int i = 0; i < D; ++i) mf_n[i] = fm[kn * a.m_sk + i * a.m_si];
#pragma unroll
      for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = i; j < D; ++j) Pf_n[sidx<D>(i, j)] = fP[kn * a.P_sk + (i * D + j) * a.P_si];
      t0_n = tp[kn * a.t_sk];
    }
    if constexpr (DriftInputs<Drift>::value) a.drift.load_inputs(a, n, k);  // u = inputs[t0_idx] of the interval (inference_ekf.py:516)
    if constexpr (!DriftTime<Drift>::value) {
      a.drift.f(rhs.mf, rhs.fmf);
      a.drift.jac(rhs.mf, rhs.G);
    } else {
#pragma unroll
      for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) rhs.G[i][j] = R(0);
    }
    // aux = psd_solve(P_f, LQL)^T
    R Sb[D][D], Lb[D][D], invb[D], X[D][D];
#pragma unroll
    for (int r = 0; r < D; ++r)
#pragma unroll
      for (int c = 0; c <= r; ++c) {
        R s = R(0.5) * (Pf[r][c] + Pf[c][r]);
        if (r == c) s += R(1e-9);
        Sb[r][c] = s;
      }
    chol_lower<R, D>(Sb, Lb, invb, bad);
#pragma unroll
    for (int r = 0; r < D; ++r)
#pragma unroll
      for (int c = 0; c < D; ++c) X[r][c] = a.LQL[sidx<D>(r, c)];
    chol_solve<R, D, D>(Lb, invb, X);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) rhs.G[i][j] += X[j][i];
    if constexpr (DriftTime<Drift>::value) {
      SmoothRhsT<R, D, Drift> rt{a.drift, {}, {}, t1, a.LQL};
#pragma unroll
      for (int i = 0; i < D; ++i) {
        rt.mf[i] = rhs.mf[i];
#pragma unroll
        for (int j = 0; j < D; ++j) rt.aux[i][j] = rhs.G[i][j];
      }
      if (integrate<R, NS>(ys, R(0), t1 - t0, a.dt0, a.max_steps, rt, C)) st |= kStatusMaxSteps;
    } else {
      if (integrate<R, NS>(ys, R(0), t1 - t0, a.dt0, a.max_steps, rhs, C)) st |= kStatusMaxSteps;
    }
    store_moments_all<R, D>(sm, sP, n * a.m_sn + k * a.m_sk, n * a.P_sn + k * a.P_sk, a.m_si, a.P_si, ys);
    t1 = t0;
  }
  if (bad) st |= kStatusNotPd;
  if (live && a.status && st) atomicOr(&a.status[n], st);
}

template <typename R, int D, int M, typename Drift, bool GENERIC = false>
__global__ __launch_bounds__(64) void ekf_smoother_reg_kernel(const RegArgs<R, D, M, Drift> a, R* __restrict__ sm,
                                                              R* __restrict__ sP) {
  ekf_smoother_reg_body<R, D, M, Drift, GENERIC>(a, sm, sP);
}

}  // namespace cdkf
