// cdkf_ukf_tangent_kernels.h -- value and gradient of the UNSCENTED filter's marginal log-likelihood for ANY drift and emission
// (ukf_tangent_body), and of the EXTENDED filter's where no forward-sensitivity kernel and no reverse sweep exists (ekf_tangent_body, at
// the end of the file): forward mode through the literal recursions on dual numbers.
//
// The reference differentiates unscented_kalman_filter with jax.value_and_grad whatever the model is (ssm_temissions.py:500, 555-568 ->
// models.py:393-408, 708 -> inference_ukf.py:93-203, the Cholesky factor's derivative included).  The closed forms of cdkf_grad_kernels.h /
// the reverse sweeps cover the drifts whose sigma-point sums collapse (quadratic, linear); this kernel covers the rest -- an MLP drift, a
// drift or an emission given as source -- by FORWARD mode through the literal sigma-point recursion: the recursion below is written once
// over a scalar type T, and T = Dual<R, 1> (cdkf_dual.h) carries the tangent along ONE leaf entry of the model through the Cholesky
// factorisations (sqrt, divisions), the sigma points, the drift / emission statements, the Dormand-Prince combination, the update's
// solves and the log-density.  Same arithmetic as inference_ukf.py:
//   _predict       dm/dt = sum_s w_mean_s f(X_s),  dP/dt = f_X^T W X + (.)^T + L Qc L^T          (inference_ukf.py:93-159)
//                  f_X^T W X = w_i sum_i (f(m + c L_i) - f(m - c L_i)) (c L_i)^T  (W's centring cancels on the symmetric set)
//   _condition_on  ybar, S = sum w_cov dY dY^T + R, C = sum w_cov dX dY^T, ll += MVN(ybar, S).log_prob(y),
//                  K = psd_solve(S, C^T)^T, m += K (y - ybar), P -= K S K^T                         (inference_ukf.py:162-203)
//   diffeqsolve    fixed-step Dormand-Prince, dt0, last step clipped to the interval's end       (diffrax_utils.py:40-165)
//
// Mapping: lane <-> (trajectory n, leaf entry p), p fastest.  Leaf entries, in this order:
//   theta [NTH] | m0 [D] | P0 (pairs i <= j, row-major) | L Qc L^T (pairs) | eta = H [M, D], h_bias [M] | R (pairs)
// A symmetric leaf is perturbed symmetrically (E_ij + E_ji) and its cotangent written symmetric -- what every symmetric parametrisation
// pairs with (oracle: ukf_loglik_grad_all_literal, ekf_loglik_grad_adjoint(full=True)).  The primal is repeated per leaf entry: a
// fallback for models no reverse sweep covers, not a fast path -- state in private arrays (scratch), loops not unrolled, built at -O1.
// The model (dimensions, f, h) comes from the translation unit launch_custom.hip generates around this header:
//   struct UtModel { static constexpr int D, M, NTH, DU;
//     template <typename R, typename T, typename TH> static __device__ void f(const T* x, const TH& theta, T (&fx)[D], const R* u, R t);
//     template <typename R, typename T, typename EH> static __device__ void h(const T* x, const EH& eta, T (&hx)[M], const R* u, R t); };
// theta / eta are VIEWS (operator[] returns the entry as a T with this lane's seed): a model with a thousand drift parameters does not
// hold a thousand dual numbers per lane.
#pragma once
#include "cdkf_math.h"
#include "cdkf_dual.h"

namespace cdkf {

template <typename R>
struct UtArgs {
  const R* par;  // theta [NTH] | m0 [D] | P0 [D, D] | L Qc L^T [D, D] | H [M, D] | h_bias [M] | R [M, M]   (row-major, full matrices)
  const R* t;
  const R* y;
  const R* u;  // inputs [.., DU] or null
  R* ll;          // [N]
  R* grad;        // [N, NTH]
  R* grad_model;  // [N, D + 2 D^2 + M D + M + M^2] or null (all == 0)
  int* status;    // [N] or null
  long N, T, t_sn, t_sk, y_sn, y_sk, y_si, u_sn, u_sk, u_si, max_steps;
  R dt0, dt_final;
  R c;    // sqrt(D + lambda)                    (inference_ukf.py:45-60)
  R wm0;  // lambda / (D + lambda)               (inference_ukf.py:63-89)
  R wc0;  // wm0 + 1 - alpha^2 + beta
  R wi;   // 1 / (2 (D + lambda))
  int all;  // 0: the drift parameters only (grad), 1: every leaf (grad and grad_model)
  int num_iter;  // the EXTENDED filter's update iterations (ekf_tangent_body; inference_ekf.py:153-199)
  int order;     // the extended filter's state_order: 1 or 2
  // VALUE mode (the FILTER of a model only these sweeps take -- an emission given as source above six dimensions): one lane per
  // trajectory, no seed, the moments written out as the filter entry points deliver them (each pointer nullable)
  int value_only;
  R* fm;  // filtered means        (after the update at t_k)
  R* fc;  // filtered covariances
  R* pm;  // predicted means       (carried to the next observation time; the last one to t_T + dt_final)
  R* pc;  // predicted covariances
  long m_sn, m_sk, m_si, P_sn, P_sk, P_si;
};

constexpr int kUtStatusNotPd = 1, kUtStatusNan = 2, kUtStatusMaxSteps = 4;  // (= kStatus* of cdkf_reg_kernels.h)

// a real with its parameter tangent lifted into the (possibly nested) dual type T: the innermost Dual<R, 1> carries (v, g), every outer
// level's directions are zero
template <typename R, typename T>
struct UtLift;
template <typename R>
struct UtLift<R, Dual<R, 1>> {
  static __device__ Dual<R, 1> make(R v, R g) {
    Dual<R, 1> r;
    r.v = v;
    r.g[0] = g;
    return r;
  }
};
template <typename R, typename S, int N>
struct UtLift<R, Dual<S, N>> {
  static __device__ Dual<S, N> make(R v, R g) {
    Dual<S, N> r;
    r.v = UtLift<R, S>::make(v, g);
    for (int k = 0; k < N; ++k) r.g[k] = UtLift<R, S>::make(R(0), R(0));
    return r;
  }
};
// entry k of a parameter vector as a T whose parameter tangent is 1 on this lane's entry
template <typename R, typename T>
struct UtView {
  const R* v;
  int seed;  // index (into v) of the entry this lane differentiates, or -1
  __device__ T operator[](int k) const { return UtLift<R, T>::make(v[k], (k == seed) ? R(1) : R(0)); }
};

__device__ inline int ut_lo(int r, int c) { return r * (r + 1) / 2 + c; }                  // lower-packed (r >= c)
__device__ inline int ut_pair(int i, int j, int n) { return i * n - i * (i - 1) / 2 + (j - i); }  // pair index (i <= j), row-major

// lower Cholesky factor of a lower-packed symmetric matrix (jnp.linalg.cholesky: a non-positive pivot yields NaN)
template <typename T, int N>
__device__ void ut_chol(const T* A, T* L, bool& bad) {
  for (int j = 0; j < N; ++j) {
    T s = A[ut_lo(j, j)];
    for (int k = 0; k < j; ++k) s -= L[ut_lo(j, k)] * L[ut_lo(j, k)];
    if (!(s.v > 0)) bad = true;
    const T piv = sqrt(s);
    L[ut_lo(j, j)] = piv;
    for (int i = j + 1; i < N; ++i) {
      T w = A[ut_lo(i, j)];
      for (int k = 0; k < j; ++k) w -= L[ut_lo(i, k)] * L[ut_lo(j, k)];
      L[ut_lo(i, j)] = w / piv;
    }
  }
}

template <typename R, typename MD, typename T>
struct UtCtx {
  UtView<R, T> th, eta;
  const T* q;  // L Qc L^T, lower-packed
  R ub[MD::DU > 0 ? MD::DU : 1];
  R c, wm0, wc0, wi;
};

// right-hand side of the moment equations at y = [m | P lower-packed]
template <typename R, typename MD, typename T>
__device__ void ut_rhs(const UtCtx<R, MD, T>& cx, const T* y, T* dy, R tt, bool& bad) {
  constexpr int D = MD::D, NPD = D * (D + 1) / 2;
  T L[NPD], f0[D], x[D], fp[D], fm[D], sum[D], foo[D * D];
  ut_chol<T, D>(y + D, L, bad);
  MD::template f<R, T>(y, cx.th, f0, cx.ub, tt);
  for (int j = 0; j < D; ++j) sum[j] = T(0.0);
  for (int e = 0; e < D * D; ++e) foo[e] = T(0.0);
  for (int i = 0; i < D; ++i) {
    for (int j = 0; j < D; ++j) x[j] = (j >= i) ? y[j] + cx.c * L[ut_lo(j, i)] : y[j];
    MD::template f<R, T>(x, cx.th, fp, cx.ub, tt);
    for (int j = 0; j < D; ++j) x[j] = (j >= i) ? y[j] - cx.c * L[ut_lo(j, i)] : y[j];
    MD::template f<R, T>(x, cx.th, fm, cx.ub, tt);
    for (int r = 0; r < D; ++r) {
      sum[r] += fp[r] + fm[r];
      const T df = fp[r] - fm[r];
      for (int b = i; b < D; ++b) foo[r * D + b] += df * (cx.c * L[ut_lo(b, i)]);
    }
  }
  for (int j = 0; j < D; ++j) dy[j] = cx.wm0 * f0[j] + cx.wi * sum[j];
  for (int r = 0; r < D; ++r)
    for (int c2 = 0; c2 <= r; ++c2) dy[D + ut_lo(r, c2)] = cx.wi * (foo[r * D + c2] + foo[c2 * D + r]) + cx.q[ut_lo(r, c2)];
}

// this lane's results: d ll / d (its leaf entry) to its place in grad / grad_model (symmetric leaves: both halves), the value from lane 0
template <typename R, typename MD>
__device__ void ut_store(const UtArgs<R>& a, long n, int p, R llv, R g, int st) {
  constexpr int D = MD::D, M = MD::M, NTH = MD::NTH;
  constexpr int NPD = D * (D + 1) / 2;
  constexpr int o_m0 = NTH, o_P0 = o_m0 + D, o_Q = o_P0 + NPD, o_H = o_Q + NPD, o_R = o_H + M * D + M;
  if (p < NTH) {
    a.grad[n * NTH + p] = g;
  } else if (a.all) {
    constexpr int GM = D + 2 * D * D + M * D + M + M * M;
    R* gm = a.grad_model + n * GM;
    auto put_pair = [&](R* base, int e, int dim) {
      int i = 0, left = e;
      while (left >= dim - i) {
        left -= dim - i;
        ++i;
      }
      const int j = i + left;
      const R val = (i == j) ? g : R(0.5) * g;
      base[i * dim + j] = val;
      base[j * dim + i] = val;
    };
    if (p < o_P0) gm[p - o_m0] = g;
    else if (p < o_Q) put_pair(gm + D, p - o_P0, D);
    else if (p < o_H) put_pair(gm + D + D * D, p - o_Q, D);
    else if (p < o_R) gm[D + 2 * D * D + (p - o_H)] = g;
    else put_pair(gm + D + 2 * D * D + M * D + M, p - o_R, M);
  }
  if (p == 0) {
    a.ll[n] = llv;
    if (a.status) a.status[n] = st;
  }
}

// value mode: the moments y = [m | P lower-packed] of trajectory n at observation k to the output arrays (full symmetric matrix)
template <typename R, typename MD, typename T>
__device__ void ut_put_moments(const UtArgs<R>& a, long n, long k, const T* y, R* mp, R* Pp) {
  constexpr int D = MD::D;
  if (mp)
    for (int i = 0; i < D; ++i) mp[n * a.m_sn + k * a.m_sk + i * a.m_si] = y[i].v;
  if (Pp)
    for (int r = 0; r < D; ++r)
      for (int c2 = 0; c2 < D; ++c2) Pp[n * a.P_sn + k * a.P_sk + (r * D + c2) * a.P_si] = y[D + (r >= c2 ? ut_lo(r, c2) : ut_lo(c2, r))].v;
}
template <typename R, typename MD>
__device__ void ukf_tangent_body(const UtArgs<R>& a) {
  constexpr int D = MD::D, M = MD::M, NTH = MD::NTH, DU = MD::DU;
  constexpr int NPD = D * (D + 1) / 2, NPM = M * (M + 1) / 2, NS = D + NPD;
  typedef Dual<R, 1> T;
  constexpr int o_m0 = NTH, o_P0 = o_m0 + D, o_Q = o_P0 + NPD, o_H = o_Q + NPD, o_R = o_H + M * D + M, n_all = o_R + NPM;
  const bool value_only = a.value_only != 0;
  const int nleaf = value_only ? 1 : (a.all ? n_all : (NTH > 0 ? NTH : 1));
  const long total = a.N * (long)nleaf;
  long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = gid < total;
  if (!live) gid = total - 1;  // idle lanes shadow the last pair; their stores are masked
  const long n = gid / nleaf;
  const int p = value_only ? -1 : (int)(gid - n * nleaf);  // (-1: no leaf entry is seeded)

  const R* p_m0 = a.par + NTH;
  const R* p_P0 = p_m0 + D;
  const R* p_Q = p_P0 + D * D;
  const R* p_eta = p_Q + D * D;
  const R* p_R = p_eta + M * D + M;

  T q[NPD], rm[NPM], y[NS];
  UtCtx<R, MD, T> cx;
  cx.th = UtView<R, T>{a.par, (p < NTH) ? p : -1};
  cx.eta = UtView<R, T>{p_eta, (p >= o_H && p < o_R) ? p - o_H : -1};
  cx.q = q;
  cx.c = a.c;
  cx.wm0 = a.wm0;
  cx.wc0 = a.wc0;
  cx.wi = a.wi;
  cx.ub[0] = R(0);
  auto seeded = [&](R v, bool on) {
    T r;
    r.v = v;
    r.g[0] = on ? R(1) : R(0);
    return r;
  };
  for (int i = 0; i < D; ++i) y[i] = seeded(p_m0[i], p == o_m0 + i);
  for (int r = 0; r < D; ++r)
    for (int c2 = 0; c2 <= r; ++c2) {
      // (the matrices arrive full; their symmetric part is what a symmetric leaf means -- cholesky symmetrises its input)
      y[D + ut_lo(r, c2)] = seeded(R(0.5) * (p_P0[r * D + c2] + p_P0[c2 * D + r]), p == o_P0 + ut_pair(c2, r, D));
      q[ut_lo(r, c2)] = seeded(R(0.5) * (p_Q[r * D + c2] + p_Q[c2 * D + r]), p == o_Q + ut_pair(c2, r, D));
    }
  for (int r = 0; r < M; ++r)
    for (int c2 = 0; c2 <= r; ++c2) rm[ut_lo(r, c2)] = seeded(R(0.5) * (p_R[r * M + c2] + p_R[c2 * M + r]), p == o_R + ut_pair(c2, r, M));

  const R* tp = a.t + n * a.t_sn;
  const R* yp = a.y + n * a.y_sn;
  double llv = 0.0, llg = 0.0;
  int st = 0;
  bool bad = false;
  constexpr R cs[6] = {R(0), R(1) / R(5), R(3) / R(10), R(4) / R(5), R(8) / R(9), R(1)};

  for (long k = 0; k < a.T; ++k) {
    const R tcur = tp[k * a.t_sk];
    for (int i = 0; i < DU; ++i) cx.ub[i] = a.u ? a.u[n * a.u_sn + k * a.u_sk + i * a.u_si] : R(0);
    {  // ---- _condition_on (inference_ukf.py:162-203) ----
      T L[NPD], x[D], Y0[M], Yp[D][M], Ym[D][M], ym[M], S[NPM], C[D][M], v[M];
      ut_chol<T, D>(y + D, L, bad);
      MD::template h<R, T>(y, cx.eta, Y0, cx.ub, tcur);
      for (int i = 0; i < D; ++i) {
        for (int j = 0; j < D; ++j) x[j] = (j >= i) ? y[j] + cx.c * L[ut_lo(j, i)] : y[j];
        MD::template h<R, T>(x, cx.eta, Yp[i], cx.ub, tcur);
        for (int j = 0; j < D; ++j) x[j] = (j >= i) ? y[j] - cx.c * L[ut_lo(j, i)] : y[j];
        MD::template h<R, T>(x, cx.eta, Ym[i], cx.ub, tcur);
      }
      for (int r = 0; r < M; ++r) {
        T s = T(0.0);
        for (int i = 0; i < D; ++i) s += Yp[i][r] + Ym[i][r];
        ym[r] = cx.wm0 * Y0[r] + cx.wi * s;
      }
      for (int r = 0; r < M; ++r)
        for (int c2 = 0; c2 <= r; ++c2) {
          T s = T(0.0);
          for (int i = 0; i < D; ++i) s += (Yp[i][r] - ym[r]) * (Yp[i][c2] - ym[c2]) + (Ym[i][r] - ym[r]) * (Ym[i][c2] - ym[c2]);
          S[ut_lo(r, c2)] = cx.wc0 * ((Y0[r] - ym[r]) * (Y0[c2] - ym[c2])) + cx.wi * s + rm[ut_lo(r, c2)];
        }
      for (int j = 0; j < D; ++j)
        for (int r = 0; r < M; ++r) {
          T s = T(0.0);
          for (int i = 0; i <= j; ++i) s += (cx.c * L[ut_lo(j, i)]) * (Yp[i][r] - Ym[i][r]);
          C[j][r] = cx.wi * s;
        }
      for (int r = 0; r < M; ++r) v[r] = T(yp[k * a.y_sk + r * a.y_si]) - ym[r];
      {  // log-density of MVN(ybar, S) at y (inference_ukf.py:197): S as given
        T Lc[NPM], z[M];
        ut_chol<T, M>(S, Lc, bad);
        T qf = T(0.0), ld = T(0.0);
        for (int i = 0; i < M; ++i) {
          T w = v[i];
          for (int kk = 0; kk < i; ++kk) w -= Lc[ut_lo(i, kk)] * z[kk];
          z[i] = w / Lc[ut_lo(i, i)];
          qf += z[i] * z[i];
          ld += log(Lc[ut_lo(i, i)]);
        }
        const T term = R(-0.5) * qf - ld - R(0.5 * 1.8378770664093453) * R(M);  // log(2 pi)
        llv += (double)term.v;
        llg += (double)term.g[0];
      }
      // K = psd_solve(S, C^T)^T (dynamax/utils/utils.py:202-207: symmetrised + 1e-9 I, Cholesky, two substitutions)
      T Lb[NPM], Sb[NPM], Kt[M][D];
      for (int e = 0; e < NPM; ++e) Sb[e] = S[e];
      for (int r = 0; r < M; ++r) Sb[ut_lo(r, r)] += T(1e-9);
      ut_chol<T, M>(Sb, Lb, bad);
      for (int j = 0; j < D; ++j) {
        T w[M];
        for (int i = 0; i < M; ++i) {
          T s = C[j][i];
          for (int kk = 0; kk < i; ++kk) s -= Lb[ut_lo(i, kk)] * w[kk];
          w[i] = s / Lb[ut_lo(i, i)];
        }
        for (int i = M - 1; i >= 0; --i) {
          T s = w[i];
          for (int kk = i + 1; kk < M; ++kk) s -= Lb[ut_lo(kk, i)] * Kt[kk][j];
          Kt[i][j] = s / Lb[ut_lo(i, i)];
        }
      }
      // m += K v,  P -= K S K^T   (no symmetrisation in the unscented update)
      for (int j = 0; j < D; ++j) {
        T s = y[j];
        for (int r = 0; r < M; ++r) s += Kt[r][j] * v[r];
        y[j] = s;
      }
      T KS[D][M];
      for (int j = 0; j < D; ++j)
        for (int b = 0; b < M; ++b) {
          T s = T(0.0);
          for (int r = 0; r < M; ++r) s += Kt[r][j] * S[r >= b ? ut_lo(r, b) : ut_lo(b, r)];
          KS[j][b] = s;
        }
      for (int r = 0; r < D; ++r)
        for (int c2 = 0; c2 <= r; ++c2) {
          T s = T(0.0);
          for (int b = 0; b < M; ++b) s += KS[r][b] * Kt[b][c2];
          y[D + ut_lo(r, c2)] -= s;
        }
    }
    if (y[0].v != y[0].v) st |= kUtStatusNan;
    if (value_only && live) ut_put_moments<R, MD, T>(a, n, k, y, a.fm, a.fc);
    const bool last = k + 1 >= a.T;
    if (!last || value_only) {  // ---- _predict (the last one, to t_T + dt_final, does not enter the log-likelihood: value mode only) ----
      const R t1 = last ? tcur + a.dt_final : tp[(k + 1) * a.t_sk];
      R tprev = tcur;
      R tnext = rmin(tcur + a.dt0, t1);
      long steps = 0;
      while (tprev < t1) {
        if (steps >= a.max_steps) {
          st |= kUtStatusMaxSteps;
          break;
        }
        const R dt = tnext - tprev;
        T ks[6][NS], ys[NS];
        for (int s = 0; s < 6; ++s) {
          for (int e = 0; e < NS; ++e) {
            T acc = T(0.0);
            for (int j = 0; j < s; ++j) acc += Dp5T<R>::a[s][j] * ks[j][e];
            ys[e] = y[e] + dt * acc;
          }
          ut_rhs<R, MD, T>(cx, ys, ks[s], tprev + cs[s] * dt, bad);
        }
        for (int e = 0; e < NS; ++e) {
          T acc = T(0.0);
          for (int s = 0; s < 6; ++s) acc += Dp5T<R>::b[s] * ks[s][e];
          y[e] += dt * acc;
        }
        tprev = rmin(tnext, t1);
        const R tn = tnext + a.dt0;
        tnext = (tn > t1 - Tol<R>::v) ? t1 : tn;
        ++steps;
      }
      if (value_only && live) ut_put_moments<R, MD, T>(a, n, k, y, a.pm, a.pc);
    }
  }
  if (bad) st |= kUtStatusNotPd;
  if (!live) return;
  if (value_only) {
    a.ll[n] = (R)llv;
    if (a.status) a.status[n] = st;
    return;
  }
  ut_store<R, MD>(a, n, p, (R)llv, (R)llg, st);
}

// ---- the EXTENDED filter on the same plan: value and gradient for any drift / emission, any num_iter ----------------------------------
// What the closed reverse sweeps do not cover (update iterations above eight dimensions, emissions given as source)
// the literal recursion on dual numbers does: jacfwd(f), jacfwd(h) by an outer Dual<T, D> over the parameter tangent T = Dual<R, 1>
// (the tangent of the Jacobian falls out with it), the second-order mean term 0.5 P grad(div f) by one more level (D <= 8).
//   _predict       dm/dt = f(m) [+ 0.5 P grad(div f)(m)],  dP/dt = F P + P F^T + L Qc L^T                  (inference_ekf.py:46-148)
//   _condition_on  num_iter x { H = jacfwd(h)(m); S = R + H P H^T; K = psd_solve(S, H P)^T; P -= K S K^T; m += K (y - h(m)) }; sym(P)
//   log-likelihood MVN(h(m_pred), H P H^T + R).log_prob(y) at the predicted moments                         (inference_ekf.py:153-199, 277-286)
template <typename R, typename MD, typename T>
__device__ void et_jac_f(const UtCtx<R, MD, T>& cx, const T* m, T* f0, T* F, R tt) {  // F row-major [D][D]
  constexpr int D = MD::D;
  typedef Dual<T, D> J;
  J x[D], fx[D];
  for (int i = 0; i < D; ++i) {
    x[i].v = m[i];
    for (int j = 0; j < D; ++j) x[i].g[j] = UtLift<R, T>::make(i == j ? R(1) : R(0), R(0));
  }
  MD::template f<R, J>(x, UtView<R, J>{cx.th.v, cx.th.seed}, fx, cx.ub, tt);
  for (int i = 0; i < D; ++i) {
    f0[i] = fx[i].v;
    for (int j = 0; j < D; ++j) F[i * D + j] = fx[i].g[j];
  }
}
// g_k = d/dx_k sum_i d f_i / d x_i (inference_ekf.py:108-116) with its parameter tangent: an inner direction set (d f_i / d x_j, all j) under
// ONE outer direction k at a time -- D evaluations on 2 (D + 1) components each.  (All D outer directions at once is the same work in one
// evaluation on (D + 1)^2 components: 1.3 KB per intermediate at D = 8, and a drift with a 90-wide hidden layer then asks for more
// private memory than a lane has -- "stack frame size exceeds limit", fresh-seed fuzz 915020.)
template <typename R, typename MD, typename T>
__device__ void et_divgrad(const UtCtx<R, MD, T>& cx, const T* m, T* g, R tt) {
  constexpr int D = MD::D;
  if constexpr (D <= 8) {
    typedef Dual<T, D> J1;
    typedef Dual<J1, 1> J2;
    for (int k = 0; k < D; ++k) {
      J2 x[D], fx[D];
      for (int i = 0; i < D; ++i) {
        x[i].v.v = m[i];
        for (int j = 0; j < D; ++j) x[i].v.g[j] = UtLift<R, T>::make(i == j ? R(1) : R(0), R(0));
        x[i].g[0] = UtLift<R, J1>::make(i == k ? R(1) : R(0), R(0));
      }
      MD::template f<R, J2>(x, UtView<R, J2>{cx.th.v, cx.th.seed}, fx, cx.ub, tt);
      T s = UtLift<R, T>::make(R(0), R(0));
      for (int i = 0; i < D; ++i) s += fx[i].g[0].g[i];
      g[k] = s;
    }
  } else {
    for (int k = 0; k < D; ++k) g[k] = UtLift<R, T>::make(R(0), R(0));
  }
}

template <typename R, typename MD, typename T>
__device__ void et_rhs(const UtCtx<R, MD, T>& cx, const T* y, T* dy, R tt, bool second) {
  constexpr int D = MD::D;
  T f0[D], F[D * D];
  et_jac_f<R, MD, T>(cx, y, f0, F, tt);
  auto P = [&](int r, int c) -> const T& { return y[D + (r >= c ? ut_lo(r, c) : ut_lo(c, r))]; };
  if (second) {
    T g[D];
    et_divgrad<R, MD, T>(cx, y, g, tt);
    for (int l = 0; l < D; ++l) {
      T s = UtLift<R, T>::make(R(0), R(0));
      for (int k = 0; k < D; ++k) s += g[k] * P(k, l);
      f0[l] += R(0.5) * s;
    }
  }
  for (int j = 0; j < D; ++j) dy[j] = f0[j];
  for (int r = 0; r < D; ++r)
    for (int c = 0; c <= r; ++c) {
      T s = cx.q[ut_lo(r, c)];
      for (int k = 0; k < D; ++k) s += F[r * D + k] * P(k, c) + P(r, k) * F[c * D + k];
      dy[D + ut_lo(r, c)] = s;
    }
}

template <typename R, typename MD>
__device__ void ekf_tangent_body(const UtArgs<R>& a) {
  constexpr int D = MD::D, M = MD::M, NTH = MD::NTH, DU = MD::DU;
  constexpr int NPD = D * (D + 1) / 2, NPM = M * (M + 1) / 2, NS = D + NPD;
  typedef Dual<R, 1> T;
  typedef Dual<T, D> J;
  constexpr int o_m0 = NTH, o_P0 = o_m0 + D, o_Q = o_P0 + NPD, o_H = o_Q + NPD, o_R = o_H + M * D + M, n_all = o_R + NPM;
  const bool value_only = a.value_only != 0;
  const int nleaf = value_only ? 1 : (a.all ? n_all : (NTH > 0 ? NTH : 1));
  const long total = a.N * (long)nleaf;
  long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = gid < total;
  if (!live) gid = total - 1;
  const long n = gid / nleaf;
  const int p = value_only ? -1 : (int)(gid - n * nleaf);
  const R* p_m0 = a.par + NTH;
  const R* p_P0 = p_m0 + D;
  const R* p_Q = p_P0 + D * D;
  const R* p_eta = p_Q + D * D;
  const R* p_R = p_eta + M * D + M;
  T q[NPD], rm[NPM], y[NS];
  UtCtx<R, MD, T> cx;
  cx.th = UtView<R, T>{a.par, (p < NTH) ? p : -1};
  cx.eta = UtView<R, T>{p_eta, (p >= o_H && p < o_R) ? p - o_H : -1};
  cx.q = q;
  cx.c = cx.wm0 = cx.wc0 = cx.wi = R(0);
  cx.ub[0] = R(0);
  auto seeded = [&](R v, bool on) { return UtLift<R, T>::make(v, on ? R(1) : R(0)); };
  for (int i = 0; i < D; ++i) y[i] = seeded(p_m0[i], p == o_m0 + i);
  for (int r = 0; r < D; ++r)
    for (int c2 = 0; c2 <= r; ++c2) {
      y[D + ut_lo(r, c2)] = seeded(R(0.5) * (p_P0[r * D + c2] + p_P0[c2 * D + r]), p == o_P0 + ut_pair(c2, r, D));
      q[ut_lo(r, c2)] = seeded(R(0.5) * (p_Q[r * D + c2] + p_Q[c2 * D + r]), p == o_Q + ut_pair(c2, r, D));
    }
  for (int r = 0; r < M; ++r)
    for (int c2 = 0; c2 <= r; ++c2) rm[ut_lo(r, c2)] = seeded(R(0.5) * (p_R[r * M + c2] + p_R[c2 * M + r]), p == o_R + ut_pair(c2, r, M));
  const R* tp = a.t + n * a.t_sn;
  const R* yp = a.y + n * a.y_sn;
  double llv = 0.0, llg = 0.0;
  int st = 0;
  bool bad = false;
  constexpr R cs[6] = {R(0), R(1) / R(5), R(3) / R(10), R(4) / R(5), R(8) / R(9), R(1)};
  const bool second = a.order == 2;

  for (long k = 0; k < a.T; ++k) {
    const R tcur = tp[k * a.t_sk];
    for (int i = 0; i < DU; ++i) cx.ub[i] = a.u ? a.u[n * a.u_sn + k * a.u_sk + i * a.u_si] : R(0);
    for (int it = 0; it < a.num_iter; ++it) {
      // h(m) and H = jacfwd(h)(m) in one evaluation on Dual<T, D>
      T hm[M], H[M * D], HP[M * D], S[NPM], v[M];
      {
        J x[D], hx[M];
        for (int i = 0; i < D; ++i) {
          x[i].v = y[i];
          for (int j = 0; j < D; ++j) x[i].g[j] = UtLift<R, T>::make(i == j ? R(1) : R(0), R(0));
        }
        MD::template h<R, J>(x, UtView<R, J>{cx.eta.v, cx.eta.seed}, hx, cx.ub, tcur);
        for (int r = 0; r < M; ++r) {
          hm[r] = hx[r].v;
          for (int j = 0; j < D; ++j) H[r * D + j] = hx[r].g[j];
        }
      }
      auto P = [&](int r, int c) -> const T& { return y[D + (r >= c ? ut_lo(r, c) : ut_lo(c, r))]; };
      for (int r = 0; r < M; ++r)
        for (int j = 0; j < D; ++j) {
          T s = UtLift<R, T>::make(R(0), R(0));
          for (int kk = 0; kk < D; ++kk) s += H[r * D + kk] * P(kk, j);
          HP[r * D + j] = s;
        }
      for (int r = 0; r < M; ++r)
        for (int c2 = 0; c2 <= r; ++c2) {
          T s = rm[ut_lo(r, c2)];
          for (int kk = 0; kk < D; ++kk) s += HP[r * D + kk] * H[c2 * D + kk];
          S[ut_lo(r, c2)] = s;
        }
      for (int r = 0; r < M; ++r) v[r] = T(yp[k * a.y_sk + r * a.y_si]) - hm[r];
      if (it == 0) {  // the log-likelihood term: at the predicted moments, S as given
        T Lc[NPM], z[M];
        ut_chol<T, M>(S, Lc, bad);
        T qf = T(0.0), ld = T(0.0);
        for (int i = 0; i < M; ++i) {
          T w = v[i];
          for (int kk = 0; kk < i; ++kk) w -= Lc[ut_lo(i, kk)] * z[kk];
          z[i] = w / Lc[ut_lo(i, i)];
          qf += z[i] * z[i];
          ld += log(Lc[ut_lo(i, i)]);
        }
        const T term = R(-0.5) * qf - ld - R(0.5 * 1.8378770664093453) * R(M);
        llv += (double)term.v;
        llg += (double)term.g[0];
      }
      T Lb[NPM], Sb[NPM], Kt[M][D];  // Kt = Sb^-1 (H P): K = Kt^T
      for (int e = 0; e < NPM; ++e) Sb[e] = S[e];
      for (int r = 0; r < M; ++r) Sb[ut_lo(r, r)] += T(1e-9);
      ut_chol<T, M>(Sb, Lb, bad);
      for (int j = 0; j < D; ++j) {
        T w[M];
        for (int i = 0; i < M; ++i) {
          T s = HP[i * D + j];
          for (int kk = 0; kk < i; ++kk) s -= Lb[ut_lo(i, kk)] * w[kk];
          w[i] = s / Lb[ut_lo(i, i)];
        }
        for (int i = M - 1; i >= 0; --i) {
          T s = w[i];
          for (int kk = i + 1; kk < M; ++kk) s -= Lb[ut_lo(kk, i)] * Kt[kk][j];
          Kt[i][j] = s / Lb[ut_lo(i, i)];
        }
      }
      T KS[D][M];
      for (int j = 0; j < D; ++j)
        for (int b = 0; b < M; ++b) {
          T s = T(0.0);
          for (int r = 0; r < M; ++r) s += Kt[r][j] * S[r >= b ? ut_lo(r, b) : ut_lo(b, r)];
          KS[j][b] = s;
        }
      for (int r = 0; r < D; ++r)
        for (int c2 = 0; c2 <= r; ++c2) {
          T s = T(0.0);
          for (int b = 0; b < M; ++b) s += KS[r][b] * Kt[b][c2];
          y[D + ut_lo(r, c2)] -= s;  // (K S K^T is symmetric: the packed triangle IS symmetrize(P - K S K^T))
        }
      for (int j = 0; j < D; ++j) {
        T s = y[j];
        for (int r = 0; r < M; ++r) s += Kt[r][j] * v[r];
        y[j] = s;
      }
    }
    if (y[0].v != y[0].v) st |= kUtStatusNan;
    if (value_only && live) ut_put_moments<R, MD, T>(a, n, k, y, a.fm, a.fc);
    const bool last = k + 1 >= a.T;
    if (!last || value_only) {
      const R t1 = last ? tcur + a.dt_final : tp[(k + 1) * a.t_sk];
      R tprev = tcur;
      R tnext = rmin(tcur + a.dt0, t1);
      long steps = 0;
      while (tprev < t1) {
        if (steps >= a.max_steps) {
          st |= kUtStatusMaxSteps;
          break;
        }
        const R dt = tnext - tprev;
        T ks[6][NS], ys[NS];
        for (int s = 0; s < 6; ++s) {
          for (int e = 0; e < NS; ++e) {
            T acc = T(0.0);
            for (int j = 0; j < s; ++j) acc += Dp5T<R>::a[s][j] * ks[j][e];
            ys[e] = y[e] + dt * acc;
          }
          et_rhs<R, MD, T>(cx, ys, ks[s], tprev + cs[s] * dt, second);
        }
        for (int e = 0; e < NS; ++e) {
          T acc = T(0.0);
          for (int s = 0; s < 6; ++s) acc += Dp5T<R>::b[s] * ks[s][e];
          y[e] += dt * acc;
        }
        tprev = rmin(tnext, t1);
        const R tn = tnext + a.dt0;
        tnext = (tn > t1 - Tol<R>::v) ? t1 : tn;
        ++steps;
      }
      if (value_only && live) ut_put_moments<R, MD, T>(a, n, k, y, a.pm, a.pc);
    }
  }
  if (bad) st |= kUtStatusNotPd;
  if (!live) return;
  if (value_only) {
    a.ll[n] = (R)llv;
    if (a.status) a.status[n] = st;
    return;
  }
  ut_store<R, MD>(a, n, p, (R)llv, (R)llg, st);
}

}  // namespace cdkf
