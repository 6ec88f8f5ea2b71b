// cdkf_wg_kernels.h -- workgroup-per-trajectory ("wg") sweep kernels for state dimensions the
// register-resident kernels do not cover (d = 5 ... 48: the MLP-drift d = 8 and Lorenz-96 d = 40 configs).
//
// Mapping: one WORKGROUP owns one trajectory for the whole time scan.  Mean, covariance,
// the six Dormand-Prince slopes, the Jacobian F and the product F P live in LDS for all T steps (d = 40, fp64:
// ten 40x40 matrices = 131 KB of the CU's 160 KB); every phase is a parallel loop over matrix entries with one
// lane per entry (or per 1x4 strip), separated by workgroup barriers.  The RK step count of an interval is
// uniform over the workgroup, so -- unlike the vmapped reference and the lane-per-trajectory kernels -- no lane
// ever waits for another trajectory's longer interval.  HBM traffic is the irregular (t_k, y_k) stream in and
// the moment arrays out, each touched once.
//
// Reference functions restated: the same as cdkf_reg_kernels.h (inference_ekf.py:46-148, 153-199, 202-326,
// 363-448, 450-539; diffrax_utils.py:40-165; dynamax/utils/utils.py:202-211).
#pragma once
#include "cdkf_math.h"

namespace cdkf {

constexpr int kDriftLinear = 0, kDriftLorenz63 = 1, kDriftLorenz96 = 2, kDriftMlp = 3;

template <typename R>
struct WgArgs {
  int kind, d, m, h1, h2;
  int q, lq;       // q = max(d, m); lq = (q rounded up to a multiple of 4) + 1: leading dimension of every LDS matrix
  int order, num_iter;
  long max_steps;
  R dt0, dt_final;
  const R* par;    // device block: theta | LQL[d*d] | LQLz[d*d] | H[m*d] | hb[m] | Rm[m*m] | m0[d] | P0[d*d]
  long o_theta, o_LQL, o_LQLz, o_H, o_hb, o_R, o_m0, o_P0;
  long N, T;
  long t_sn, t_sk, y_sn, y_sk, y_si, m_sn, m_sk, m_si, P_sn, P_sk, P_si;
  const R* t;
  const R* y;
  R* ll;
  R* fm;
  R* fP;
  R* pm;
  R* pP;
  R* sm;  // smoother outputs (smoother kernel only)
  R* sP;
  int* status;
};

// ---- LDS carve-up ---------------------------------------------------------------------------------------
template <typename R>
struct WgLds {
  int d, m, q, lq, msz, vsz;
  R* base;
  __device__ WgLds(R* b, int d_, int m_, int q_, int lq_) : d(d_), m(m_), q(q_), lq(lq_), base(b) {
    msz = q * lq;
    vsz = lq;
  }
  // matrices: 0 P (state), 1 Ps (stage), 2..7 k1..k6, 8 F, 9 A      vectors: same numbering + 10 f, 11 g
  __device__ R* mat(int i) const { return base + (long)i * msz; }
  __device__ R* vec(int i) const { return base + 10L * msz + (long)i * vsz; }
  static size_t bytes(int q, int lq, int extra_reals) { return sizeof(R) * (10L * q * lq + 14L * lq + extra_reals); }
  __device__ R* extra() const { return base + 10L * msz + 14L * vsz; }
};

// Integer division by a run-time divisor costs ~40 instructions on CDNA; every entry loop needs (row, col) from a
// flat index, so use a float reciprocal with a +-1 correction (exact for 0 <= e < 2^23, 0 < n < 2^12).
__device__ __forceinline__ int fdiv(int e, int n) {
  int q = (int)((float)e * __frcp_rn((float)n));
  const int r = e - q * n;
  q += (r >= n) - (r < 0);
  return q;
}
__device__ __forceinline__ int fmod_(int e, int n) { return e - fdiv(e, n) * n; }

#define CDKF_WG_FOR(idx, n) for (int idx = threadIdx.x; idx < (n); idx += blockDim.x)

// C[r x c] = A[r x k] * B[k x c]   (all in LDS, leading dimension lq); 1x4 strips per lane
template <typename R>
__device__ __forceinline__ void wg_matmul(R* __restrict__ C, const R* __restrict__ A, const R* __restrict__ B, int r,
                                          int k, int c, int lq) {
  const int c4 = (c + 3) >> 2;
  CDKF_WG_FOR(e, r * c4) {
    const int i = fdiv(e, c4), j = (e - i * c4) << 2;
    R a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    const R* Ai = A + i * lq;
    const R* Bj = B + j;
    for (int kk = 0; kk < k; ++kk) {
      const R f = Ai[kk];
      const R* b = Bj + kk * lq;
      a0 = rfma(f, b[0], a0);
      a1 = rfma(f, b[1], a1);
      a2 = rfma(f, b[2], a2);
      a3 = rfma(f, b[3], a3);
    }
    R* cij = C + i * lq + j;
    cij[0] = a0;
    cij[1] = a1;
    cij[2] = a2;
    cij[3] = a3;  // columns >= c of the padded strip hold garbage-free zeros or finite junk; never read as data
  }
}

// C[r x c] = A[r x k] * B^T, B is [c x k]
template <typename R>
__device__ __forceinline__ void wg_matmul_nt(R* __restrict__ C, const R* __restrict__ A, const R* __restrict__ B, int r,
                                             int k, int c, int lq) {
  CDKF_WG_FOR(e, r * c) {
    const int i = fdiv(e, c), j = e - i * c;
    R acc = 0;
    for (int kk = 0; kk < k; ++kk) acc = rfma(A[i * lq + kk], B[j * lq + kk], acc);
    C[i * lq + j] = acc;
  }
}

// C[r x c] = A^T * B, A is [k x r], B is [k x c]
template <typename R>
__device__ __forceinline__ void wg_matmul_tn(R* __restrict__ C, const R* __restrict__ A, const R* __restrict__ B, int r,
                                             int k, int c, int lq) {
  CDKF_WG_FOR(e, r * c) {
    const int i = fdiv(e, c), j = e - i * c;
    R acc = 0;
    for (int kk = 0; kk < k; ++kk) acc = rfma(A[kk * lq + i], B[kk * lq + j], acc);
    C[i * lq + j] = acc;
  }
}

// In-place lower Cholesky of the n x n matrix S (lower triangle read), right-looking, with reciprocal pivots in
// inv[].  A non-positive pivot yields NaN like jnp.linalg.cholesky; *bad is raised.
template <typename R>
__device__ __forceinline__ void wg_cholesky(R* S, R* inv, int n, int lq, int* bad) {
  for (int j = 0; j < n; ++j) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const R s = S[j * lq + j];
      if (!(s > R(0))) *bad = 1;
      const R r = rrsqrt(s);
      inv[j] = r;
      S[j * lq + j] = s * r;
    }
    __syncthreads();
    const R r = inv[j];
    CDKF_WG_FOR(i, n - j - 1) S[(j + 1 + i) * lq + j] *= r;
    __syncthreads();
    // trailing update of the lower triangle: S[a][b] -= L[a][j] L[b][j], j < b <= a
    const int rem = n - j - 1;
    CDKF_WG_FOR(e, rem * rem) {
      const int a = j + 1 + fdiv(e, rem), b = j + 1 + fmod_(e, rem);
      if (b <= a) S[a * lq + b] = rfma(-S[a * lq + j], S[b * lq + j], S[a * lq + b]);
    }
  }
  __syncthreads();
}

// Solve (L L^T) X = B in place, B is [n x c]; right-looking substitution, one barrier per row.
template <typename R>
__device__ __forceinline__ void wg_chol_solve(const R* L, const R* inv, R* B, int n, int c, int lq) {
  for (int i = 0; i < n; ++i) {  // forward: L Y = B
    __syncthreads();
    CDKF_WG_FOR(j, c) B[i * lq + j] *= inv[i];
    __syncthreads();
    const int rem = n - i - 1;
    CDKF_WG_FOR(e, rem * c) {
      const int r = i + 1 + fdiv(e, c), j = fmod_(e, c);
      B[r * lq + j] = rfma(-L[r * lq + i], B[i * lq + j], B[r * lq + j]);
    }
  }
  for (int i = n - 1; i >= 0; --i) {  // backward: L^T X = Y
    __syncthreads();
    CDKF_WG_FOR(j, c) B[i * lq + j] *= inv[i];
    __syncthreads();
    CDKF_WG_FOR(e, i * c) {
      const int r = fdiv(e, c), j = fmod_(e, c);
      B[r * lq + j] = rfma(-L[i * lq + r], B[i * lq + j], B[r * lq + j]);
    }
  }
  __syncthreads();
}

// MLP scratch in LDS behind the matrices: a1[h1] a2[h2] s2[h2] tq[h1] Cm[d*h1] Gm[h2*h1], then the weights
__host__ __device__ inline int wg_extra_reals(int kind, int d, int h1, int h2) {
  return kind == kDriftMlp ? (2 * h1 + 2 * h2 + d * h1 + h2 * h1) : 0;
}
__host__ __device__ inline int wg_mlp_theta_reals(int kind, int d, int h1, int h2) {
  return kind == kDriftMlp ? (h1 * d + h1 + h2 * h1 + h2 + d * h2 + d) : 0;
}

// ---- drift: f(x) -> fv, Jacobian -> F (dense, LDS), g = grad(div f) -> gv (MLP only) ------------------------
template <typename R>
__device__ __forceinline__ R rtanh(R x) {
  return (R)tanh((double)x);
}
template <>
__device__ __forceinline__ float rtanh<float>(float x) {
  return tanhf(x);
}

// returns true if F is banded (Lorenz-96) so that the caller may use the 4-term product
template <typename R>
__device__ void wg_drift(const WgArgs<R>& a, const WgLds<R>& L, const R* __restrict__ x, R* __restrict__ fv,
                         R* __restrict__ F, R* __restrict__ gv, bool want_jac) {
  const int d = a.d, lq = a.lq;
  // MLP: the weights were copied behind the scratch by wg_mlp_prepare (LDS); other drifts read the few
  // parameters they need from the L2-resident parameter block
  const R* th = (a.kind == kDriftMlp) ? L.extra() + wg_extra_reals(kDriftMlp, a.d, a.h1, a.h2) : a.par + a.o_theta;
  if (a.kind == kDriftLinear) {
    CDKF_WG_FOR(i, d) {
      R s = 0;
      for (int j = 0; j < d; ++j) s = rfma(th[i * d + j], x[j], s);
      fv[i] = s + th[d * d + i];
    }
    if (want_jac) CDKF_WG_FOR(e, d * d) F[(fdiv(e, d)) * lq + (fmod_(e, d))] = th[e];
  } else if (a.kind == kDriftLorenz63) {
    if (threadIdx.x == 0) {
      fv[0] = th[0] * (x[1] - x[0]);
      fv[1] = x[0] * (th[1] - x[2]) - x[1];
      fv[2] = x[0] * x[1] - th[2] * x[2];
      if (want_jac) {
        F[0] = -th[0]; F[1] = th[0]; F[2] = 0;
        F[lq] = th[1] - x[2]; F[lq + 1] = -1; F[lq + 2] = -x[0];
        F[2 * lq] = x[1]; F[2 * lq + 1] = x[0]; F[2 * lq + 2] = -th[2];
      }
    }
  } else if (a.kind == kDriftLorenz96) {
    CDKF_WG_FOR(i, d) {
      const R xp1 = x[(i + 1) % d], xm1 = x[(i + d - 1) % d], xm2 = x[(i + d - 2) % d];
      fv[i] = rfma(xp1 - xm2, xm1, th[0] - x[i]);
    }
    if (want_jac) {
      CDKF_WG_FOR(e, d * d) F[(fdiv(e, d)) * lq + (fmod_(e, d))] = 0;
      __syncthreads();
      CDKF_WG_FOR(i, d) {
        const R xp1 = x[(i + 1) % d], xm1 = x[(i + d - 1) % d], xm2 = x[(i + d - 2) % d];
        F[i * lq + (i + 1) % d] = xm1;
        F[i * lq + (i + d - 2) % d] = -xm1;
        F[i * lq + (i + d - 1) % d] = xp1 - xm2;
        F[i * lq + i] = -1;
      }
    }
  } else {  // MLP: f = W3 tanh(W2 tanh(W1 x + b1) + b2) + b3
    const int h1 = a.h1, h2 = a.h2;
    const R* W1 = th;
    const R* b1 = W1 + h1 * d;
    const R* W2 = b1 + h1;
    const R* b2 = W2 + h2 * h1;
    const R* W3 = b2 + h2;
    const R* b3 = W3 + d * h2;
    R* a1 = L.extra();          // [h1]
    R* a2 = a1 + h1;            // [h2]
    R* s2 = a2 + h2;            // [h2]   dd2 * (G d1)
    R* tq = s2 + h2;            // [h1]   t_direct + t_chain
    R* Cm = tq + h1;            // [d x h1]  W3 D2 W2 D1
    R* Gm = Cm + d * h1;        // [h2 x h1] ((W1 W3)^T * W2), constant: filled once by wg_mlp_prepare
    CDKF_WG_FOR(p, h1) {
      R s = b1[p];
      for (int j = 0; j < d; ++j) s = rfma(W1[p * d + j], x[j], s);
      a1[p] = rtanh(s);
    }
    __syncthreads();
    CDKF_WG_FOR(p, h2) {
      R s = b2[p];
      for (int j = 0; j < h1; ++j) s = rfma(W2[p * h1 + j], a1[j], s);
      a2[p] = rtanh(s);
    }
    __syncthreads();
    CDKF_WG_FOR(i, d) {
      R s = b3[i];
      for (int p = 0; p < h2; ++p) s = rfma(W3[i * h2 + p], a2[p], s);
      fv[i] = s;
    }
    if (want_jac) {
      // C = W3 diag(1 - a2^2) W2 diag(1 - a1^2)   [d x h1];   J = C W1
      CDKF_WG_FOR(e, d * h1) {
        const int i = fdiv(e, h1), qq = fmod_(e, h1);
        R s = 0;
        for (int p = 0; p < h2; ++p) s = rfma(W3[i * h2 + p] * (R(1) - a2[p] * a2[p]), W2[p * h1 + qq], s);
        Cm[e] = s * (R(1) - a1[qq] * a1[qq]);
      }
      __syncthreads();
      CDKF_WG_FOR(e, d * d) {
        const int i = fdiv(e, d), j = fmod_(e, d);
        R s = 0;
        for (int qq = 0; qq < h1; ++qq) s = rfma(Cm[i * h1 + qq], W1[qq * d + j], s);
        F[i * lq + j] = s;
      }
      if (gv) {
        // g = d tr(J) / dx  with tr(J) = sum_pq G_pq d2_p d1_q, G = ((W1 W3)^T) * W2 (elementwise)
        CDKF_WG_FOR(p, h2) {
          R s = 0;
          for (int qq = 0; qq < h1; ++qq) s = rfma(Gm[p * h1 + qq], R(1) - a1[qq] * a1[qq], s);
          s2[p] = R(-2) * a2[p] * (R(1) - a2[p] * a2[p]) * s;
        }
        __syncthreads();
        CDKF_WG_FOR(qq, h1) {
          const R d1 = R(1) - a1[qq] * a1[qq];
          R td = 0, tc = 0;
          for (int p = 0; p < h2; ++p) {
            td = rfma(R(1) - a2[p] * a2[p], Gm[p * h1 + qq], td);
            tc = rfma(s2[p], W2[p * h1 + qq], tc);
          }
          tq[qq] = td * (R(-2) * a1[qq] * d1) + tc * d1;
        }
        __syncthreads();
        CDKF_WG_FOR(l, d) {
          R s = 0;
          for (int qq = 0; qq < h1; ++qq) s = rfma(tq[qq], W1[qq * d + l], s);
          gv[l] = s;
        }
      }
    }
  }
  __syncthreads();
}

// G_pq = (sum_i W3_ip W1_qi) * W2_pq : constant of the MLP, built once per workgroup
template <typename R>
__device__ void wg_mlp_prepare(const WgArgs<R>& a, const WgLds<R>& L) {
  if (a.kind != kDriftMlp) return;
  const int d = a.d, h1 = a.h1, h2 = a.h2;
  R* thl = L.extra() + wg_extra_reals(kDriftMlp, d, h1, h2);
  CDKF_WG_FOR(e, wg_mlp_theta_reals(kDriftMlp, d, h1, h2)) thl[e] = (a.par + a.o_theta)[e];
  __syncthreads();
  const R* th = thl;
  const R* W1 = th;
  const R* W2 = W1 + h1 * d + h1;
  const R* W3 = W2 + h2 * h1 + h2;
  R* Gm = L.extra() + (2 * h1 + 2 * h2 + d * h1);
  CDKF_WG_FOR(e, h2 * h1) {
    const int p = fdiv(e, h1), qq = fmod_(e, h1);
    R s = 0;
    for (int i = 0; i < d; ++i) s = rfma(W3[i * h2 + p], W1[qq * d + i], s);
    Gm[e] = s * W2[e];
  }
  __syncthreads();
}

// ---- right-hand sides: write slope (km, kP) for the stage value (ms, Ps) --------------------------------
// EKF moment ODE (inference_ekf.py:76-123).  kP = F Ps + (F Ps)^T + LQL ; km = f (+ 0.5 Ps g).
template <typename R>
__device__ void wg_rhs_ekf(const WgArgs<R>& a, const WgLds<R>& L, const R* ms, const R* Ps, R* km, R* kP, bool mean_only) {
  const int d = a.d, lq = a.lq;
  R* F = L.mat(8);
  R* A = L.mat(9);
  R* gv = L.vec(11);
  const bool second = (a.order == 2) && (a.kind == kDriftMlp);
  wg_drift(a, L, ms, km, F, second ? gv : nullptr, !mean_only);
  if (mean_only) return;
  if (a.kind == kDriftLorenz96) {  // banded Jacobian: 4 terms per entry
    CDKF_WG_FOR(e, d * d) {
      const int i = fdiv(e, d), j = fmod_(e, d);
      const int k1 = (i + 1) % d, k2 = (i + d - 2) % d, k3 = (i + d - 1) % d;
      R s = F[i * lq + k2] * Ps[k2 * lq + j];
      s = rfma(F[i * lq + k3], Ps[k3 * lq + j], s);
      s = rfma(F[i * lq + i], Ps[i * lq + j], s);
      s = rfma(F[i * lq + k1], Ps[k1 * lq + j], s);
      A[i * lq + j] = s;
    }
  } else {
    wg_matmul(A, F, Ps, d, d, d, lq);
  }
  __syncthreads();
  const R* LQL = a.par + a.o_LQL;
  CDKF_WG_FOR(e, d * d) {
    const int j = fdiv(e, d), i = fmod_(e, d);  // i fastest: A[j][i] row reads are contiguous, A[i][j] strided
    kP[i * lq + j] = (A[i * lq + j] + A[j * lq + i]) + LQL[i * d + j];
  }
  if (second) {
    CDKF_WG_FOR(l, d) {
      R s = 0;
      for (int k = 0; k < d; ++k) s = rfma(gv[k], Ps[k * lq + l], s);
      km[l] = rfma(R(0.5), s, km[l]);
    }
  }
  __syncthreads();
}

// Dormand-Prince step on the LDS-resident state (m, P) = (vec(0), mat(0)); slopes in vec/mat(2..7), stage in (1).
template <typename R, typename RhsFn>
__device__ void wg_dopri5_step(const WgLds<R>& L, int d, int lq, R dt, bool with_P, RhsFn rhs) {
  using C = Dp5<R>;
  const R A_[6][5] = {{0, 0, 0, 0, 0},
                      {C::a21, 0, 0, 0, 0},
                      {C::a31, C::a32, 0, 0, 0},
                      {C::a41, C::a42, C::a43, 0, 0},
                      {C::a51, C::a52, C::a53, C::a54, 0},
                      {C::a61, C::a62, C::a63, C::a64, C::a65}};
  const R B_[6] = {C::b1, 0, C::b3, C::b4, C::b5, C::b6};
  R* m0 = L.vec(0);
  R* P0 = L.mat(0);
  R* ms = L.vec(1);
  R* Ps = L.mat(1);
  const int np = with_P ? d * lq : 0;
  for (int s = 0; s < 6; ++s) {
    if (s == 0) {
      rhs(m0, P0, L.vec(2), L.mat(2));
    } else {
      CDKF_WG_FOR(e, d + np) {
        const bool isP = e >= d;
        const int o = isP ? e - d : e;
        R acc = 0;
        for (int j = 0; j < s; ++j) acc = rfma(A_[s][j], (isP ? L.mat(2 + j) : L.vec(2 + j))[o], acc);
        if (isP)
          Ps[o] = rfma(dt, acc, P0[o]);
        else
          ms[o] = rfma(dt, acc, m0[o]);
      }
      __syncthreads();
      rhs(ms, Ps, L.vec(2 + s), L.mat(2 + s));
    }
  }
  CDKF_WG_FOR(e, d + np) {
    const bool isP = e >= d;
    const int o = isP ? e - d : e;
    R acc = 0;
    for (int j = 0; j < 6; ++j) acc = rfma(B_[j], (isP ? L.mat(2 + j) : L.vec(2 + j))[o], acc);
    if (isP)
      P0[o] = rfma(dt, acc, P0[o]);
    else
      m0[o] = rfma(dt, acc, m0[o]);
  }
  __syncthreads();
}

template <typename R, typename RhsFn>
__device__ bool wg_integrate(const WgLds<R>& L, int d, int lq, R t0, R t1, R dt0, long max_steps, bool with_P, RhsFn rhs) {
  R tprev = t0;
  R tnext = rmin(t0 + dt0, t1);
  long steps = 0;
  while (tprev < t1) {  // uniform over the workgroup
    if (steps >= max_steps) return true;
    wg_dopri5_step(L, d, lq, tnext - tprev, with_P, rhs);
    tprev = rmin(tnext, t1);
    const R tn = tnext + dt0;
    tnext = (tn > t1 - Tol<R>::v) ? t1 : tn;
    ++steps;
  }
  return false;
}

// EKF update on the LDS state (inference_ekf.py:153-199, 285-286); scratch = slope matrices 2..7.
template <typename R>
__device__ void wg_ekf_update(const WgArgs<R>& a, const WgLds<R>& L, const R* __restrict__ yobs_lds, double* ll, int* bad) {
  const int d = a.d, m = a.m, lq = a.lq;
  const R* H = a.par + a.o_H;
  const R* hb = a.par + a.o_hb;
  const R* Rm = a.par + a.o_R;
  R* mm = L.vec(0);
  R* P = L.mat(0);
  R* HP = L.mat(2);
  R* S = L.mat(3);
  R* Lc = L.mat(4);
  R* X = L.mat(5);
  R* SX = L.mat(6);
  R* Tm = L.mat(7);
  R* Hl = L.mat(8);  // H copied to LDS with leading dimension lq (F is dead during the update)
  R* v = L.vec(2);
  R* z = L.vec(3);
  R* inv = L.vec(4);
  CDKF_WG_FOR(e, m * d) Hl[(fdiv(e, d)) * lq + (fmod_(e, d))] = H[e];
  __syncthreads();
  for (int it = 0; it < a.num_iter; ++it) {
    wg_matmul(HP, Hl, P, m, d, d, lq);
    __syncthreads();
    wg_matmul_nt(S, HP, Hl, m, d, m, lq);
    CDKF_WG_FOR(r, m) {
      R s = 0;
      for (int k = 0; k < d; ++k) s = rfma(Hl[r * lq + k], mm[k], s);
      v[r] = yobs_lds[r] - (s + hb[r]);
    }
    __syncthreads();
    CDKF_WG_FOR(e, m * m) S[(fdiv(e, m)) * lq + (fmod_(e, m))] += Rm[e];
    __syncthreads();
    if (it == 0) {  // TFP log_prob with the un-jittered S
      CDKF_WG_FOR(e, m * m) Lc[(fdiv(e, m)) * lq + (fmod_(e, m))] = S[(fdiv(e, m)) * lq + (fmod_(e, m))];
      CDKF_WG_FOR(r, m) z[r] = v[r];
      wg_cholesky(Lc, inv, m, lq, bad);
      for (int i = 0; i < m; ++i) {  // forward substitution z = L^-1 v, one barrier per row
        if (threadIdx.x == 0) z[i] *= inv[i];
        __syncthreads();
        CDKF_WG_FOR(r, m - i - 1) z[i + 1 + r] = rfma(-Lc[(i + 1 + r) * lq + i], z[i], z[i + 1 + r]);
        __syncthreads();
      }
      if (threadIdx.x == 0) {
        double qd = 0.0, ld = 0.0;
        for (int i = 0; i < m; ++i) {
          qd += (double)z[i] * (double)z[i];
          ld += log((double)inv[i]);
        }
        *ll += -0.5 * qd + ld - 0.5 * m * 1.8378770664093454835606594728112;
      }
    }
    // psd_solve(S, HP): symmetrize + 1e-9 I, Cholesky, solve
    CDKF_WG_FOR(e, m * m) {
      const int r = fdiv(e, m), c = fmod_(e, m);
      R s = R(0.5) * (S[r * lq + c] + S[c * lq + r]);
      if (r == c) s += R(1e-9);
      Lc[r * lq + c] = s;
    }
    CDKF_WG_FOR(e, m * d) X[(fdiv(e, d)) * lq + (fmod_(e, d))] = HP[(fdiv(e, d)) * lq + (fmod_(e, d))];
    wg_cholesky(Lc, inv, m, lq, bad);
    wg_chol_solve(Lc, inv, X, m, d, lq);
    wg_matmul(SX, S, X, m, m, d, lq);  // S X
    __syncthreads();
    wg_matmul_tn(Tm, X, SX, d, m, d, lq);  // X^T S X = K S K^T
    CDKF_WG_FOR(i, d) {
      R s = mm[i];
      for (int r = 0; r < m; ++r) s = rfma(X[r * lq + i], v[r], s);
      z[i] = s;  // new mean staged in z (v still needed by other lanes)
    }
    __syncthreads();
    CDKF_WG_FOR(i, d) mm[i] = z[i];
    CDKF_WG_FOR(e, d * d) {
      const int i = fdiv(e, d), j = fmod_(e, d);
      P[i * lq + j] -= Tm[i * lq + j];
    }
    __syncthreads();
  }
  // symmetrize (dynamax/utils/utils.py:209-211)
  CDKF_WG_FOR(e, d * d) {
    const int i = fdiv(e, d), j = fmod_(e, d);
    if (i < j) {
      const R s = R(0.5) * (P[i * lq + j] + P[j * lq + i]);
      P[i * lq + j] = s;
      P[j * lq + i] = s;
    }
  }
  __syncthreads();
}

template <typename R>
__device__ __forceinline__ void wg_store(const WgArgs<R>& a, const WgLds<R>& L, R* mo, R* Po, long n, long k) {
  const int d = a.d, lq = a.lq;
  if (mo) {
    R* p = mo + n * a.m_sn + k * a.m_sk;
    const R* mm = L.vec(0);
    CDKF_WG_FOR(i, d) p[i * a.m_si] = mm[i];
  }
  if (Po) {
    R* p = Po + n * a.P_sn + k * a.P_sk;
    const R* P = L.mat(0);
    CDKF_WG_FOR(e, d * d) p[e * a.P_si] = P[(fdiv(e, d)) * lq + (fmod_(e, d))];
  }
}

// ---- EKF filter sweep, one workgroup per trajectory -------------------------------------------------------
template <typename R>
__global__ void ekf_filter_wg_kernel(const WgArgs<R> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  WgLds<R> L(reinterpret_cast<R*>(smem_raw), a.d, a.m, a.q, a.lq);
  __shared__ int bad;
  __shared__ double ll;
  const long n = blockIdx.x;
  const int d = a.d, m = a.m, lq = a.lq;
  if (threadIdx.x == 0) {
    bad = 0;
    ll = 0.0;
  }
  CDKF_WG_FOR(e, (int)(10L * a.q * lq + 14L * lq)) L.base[e] = 0;
  __syncthreads();
  CDKF_WG_FOR(i, d) L.vec(0)[i] = (a.par + a.o_m0)[i];
  CDKF_WG_FOR(e, d * d) {
    const int i = fdiv(e, d), j = fmod_(e, d);
    L.mat(0)[i * lq + j] = R(0.5) * ((a.par + a.o_P0)[i * d + j] + (a.par + a.o_P0)[j * d + i]);
  }
  wg_mlp_prepare(a, L);
  __syncthreads();
  const R* tp = a.t + n * a.t_sn;
  const R* yp = a.y + n * a.y_sn;
  R* yl = L.vec(12);
  int st = 0;
  const bool zeroth = a.order == 0;
  auto rhs = [&](const R* ms, const R* Ps, R* km, R* kP) { wg_rhs_ekf(a, L, ms, Ps, km, kP, zeroth); };
  for (long k = 0; k < a.T; ++k) {
    CDKF_WG_FOR(r, m) yl[r] = yp[k * a.y_sk + r * a.y_si];
    const R t0 = tp[k * a.t_sk];
    const R t1 = (k + 1 < a.T) ? tp[(k + 1) * a.t_sk] : t0 + a.dt_final;
    __syncthreads();
    wg_ekf_update(a, L, yl, &ll, &bad);
    wg_store(a, L, a.fm, a.fP, n, k);
    __syncthreads();
    if (wg_integrate(L, d, lq, t0, t1, a.dt0, a.max_steps, !zeroth, rhs)) st |= kStatusMaxSteps;
    if (zeroth) {
      const R sq = rsqrt_(t1 - t0);
      const R* Qz = a.par + a.o_LQLz;
      CDKF_WG_FOR(e, d * d) L.mat(0)[(fdiv(e, d)) * lq + (fmod_(e, d))] = rfma(sq, Qz[e], L.mat(0)[(fdiv(e, d)) * lq + (fmod_(e, d))]);
      __syncthreads();
    }
    wg_store(a, L, a.pm, a.pP, n, k);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (bad) st |= kStatusNotPd;
    if (ll != ll) st |= kStatusNan;
    a.ll[n] = (R)ll;
    if (a.status) a.status[n] = st;
  }
}

// ---- EKF smoother backward sweep (inference_ekf.py:363-448, 503-531) ------------------------------------------
// Per interval [t_k, t_{k+1}] the filtered (m_f, P_f) at t_k are constants: G = F(m_f) + psd_solve(P_f, LQL)^T
// and f(m_f) are formed once (mat 8, vec 10); the reverse-time right-hand side is
//   dm = -[f(m_f) + G (m_s - m_f)],   dP = -[G P_s + (G P_s)^T - LQL].
template <typename R>
__global__ void ekf_smoother_wg_kernel(const WgArgs<R> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  WgLds<R> L(reinterpret_cast<R*>(smem_raw), a.d, a.m, a.q, a.lq);
  __shared__ int bad;
  const long n = blockIdx.x;
  const int d = a.d, lq = a.lq;
  if (threadIdx.x == 0) bad = 0;
  CDKF_WG_FOR(e, (int)(10L * a.q * lq + 14L * lq)) L.base[e] = 0;
  __syncthreads();
  wg_mlp_prepare(a, L);
  const R* tp = a.t + n * a.t_sn;
  const R* fm = a.fm + n * a.m_sn;
  const R* fP = a.fP + n * a.P_sn;
  int st = 0;
  {
    const long k = a.T - 1;
    CDKF_WG_FOR(i, d) L.vec(0)[i] = fm[k * a.m_sk + i * a.m_si];
    CDKF_WG_FOR(e, d * d) L.mat(0)[(fdiv(e, d)) * lq + (fmod_(e, d))] = fP[k * a.P_sk + e * a.P_si];
    __syncthreads();
    wg_store(a, L, a.sm, a.sP, n, k);
    __syncthreads();
  }
  R* G = L.mat(8);
  R* A = L.mat(9);
  R* fmf = L.vec(10);
  R* mf = L.vec(13);
  R* inv = L.vec(4);
  const R* LQL = a.par + a.o_LQL;
  auto rhs = [&](const R* ms, const R* Ps, R* km, R* kP) {
    wg_matmul(A, G, Ps, d, d, d, lq);
    CDKF_WG_FOR(i, d) {
      R s = 0;
      for (int k = 0; k < d; ++k) s = rfma(G[i * lq + k], ms[k] - mf[k], s);
      km[i] = -(fmf[i] + s);
    }
    __syncthreads();
    CDKF_WG_FOR(e, d * d) {
      const int j = fdiv(e, d), i = fmod_(e, d);
      kP[i * lq + j] = -((A[i * lq + j] + A[j * lq + i]) - LQL[i * d + j]);
    }
    __syncthreads();
  };
  R t1 = tp[(a.T - 1) * a.t_sk];
  for (long k = a.T - 2; k >= 0; --k) {
    const R t0 = tp[k * a.t_sk];
    R* Lc = L.mat(9);
    R* X = L.mat(1);
    CDKF_WG_FOR(i, d) mf[i] = fm[k * a.m_sk + i * a.m_si];
    CDKF_WG_FOR(e, d * d) L.mat(2)[(fdiv(e, d)) * lq + (fmod_(e, d))] = fP[k * a.P_sk + e * a.P_si];
    __syncthreads();
    CDKF_WG_FOR(e, d * d) {
      const int r = fdiv(e, d), c = fmod_(e, d);
      R s = R(0.5) * (L.mat(2)[r * lq + c] + L.mat(2)[c * lq + r]);
      if (r == c) s += R(1e-9);
      Lc[r * lq + c] = s;
      X[r * lq + c] = LQL[e];
    }
    wg_cholesky(Lc, inv, d, lq, &bad);
    wg_chol_solve(Lc, inv, X, d, d, lq);  // X = P_f^{-1} LQL
    wg_drift(a, L, mf, fmf, G, (R*)nullptr, true);
    CDKF_WG_FOR(e, d * d) {
      const int i = fdiv(e, d), j = fmod_(e, d);
      G[i * lq + j] += X[j * lq + i];
    }
    __syncthreads();
    if (wg_integrate(L, d, lq, R(0), t1 - t0, a.dt0, a.max_steps, true, rhs)) st |= kStatusMaxSteps;
    wg_store(a, L, a.sm, a.sP, n, k);
    __syncthreads();
    t1 = t0;
  }
  if (threadIdx.x == 0 && a.status) {
    if (bad) st |= kStatusNotPd;
    if (st) atomicOr(&a.status[n], st);
  }
}

}  // namespace cdkf
