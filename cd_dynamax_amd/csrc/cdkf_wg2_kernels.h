// cdkf_wg2_kernels.h -- workgroup-per-trajectory sweep kernels, second generation.
//
// Mapping: one WORKGROUP owns one trajectory for the whole time scan.
//  * Every thread OWNS up to EPT fixed entries (i,j) of the d x d covariance for the whole sweep; the six
//    Dormand-Prince slopes of those entries (and of the mean, threads 0..d-1) live in its REGISTERS.  LDS holds only
//    what other threads must see: the covariance P, the current stage value Ps, the stage mean and -- for drifts with
//    a dense Jacobian -- F and F Ps.  Two barriers per RK stage.
//  * The measurement update uses blocked (8-wide) right-looking Cholesky factorisations and blocked triangular
//    solves over all threads: ~45 barriers per update at m = 40 instead of one or two per row.
//  * Lorenz-96's Jacobian has 4 entries per row: its rows are evaluated on the fly from the stage mean, no F matrix.
//  * The RK step count of an interval is uniform over the workgroup: no lane waits for another trajectory.
// Reference functions restated: as in cdkf_reg_kernels.h.
#pragma once
#include "cdkf_math.h"

namespace cdkf {
#ifdef CDKF_PHASE_PROFILE
#define CDKF_TICK(i) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 0) prof_t[i] = wall_clock64(); } while (0)
#else
#define CDKF_TICK(i) do {} while (0)
#endif

constexpr int kDriftLinear = 0, kDriftLorenz63 = 1, kDriftLorenz96 = 2, kDriftMlp = 3;
constexpr int kDriftCustomBase = 1000;  // CDKF_DRIFT_CUSTOM_BASE: drifts that arrive as C source (launch_custom.hip)
constexpr int kWgBlock = 8;  // panel width of the blocked factorisations

template <typename R>
struct WgArgs {
  int kind, d, m, h1, h2;
  int q, lq;  // q = max(d, m); lq = (q rounded up to a multiple of 4) + 1: leading dimension of every LDS matrix
  int order, num_iter, hsel, forecast, ukf;
  R ukf_c, ukf_wm0, ukf_wi;  // sigma scale sqrt(n + lambda), w_mean[0], 1 / (2 (n + lambda))   (inference_ukf.py:63-89)
  long max_steps;
  R dt0, dt_final;
  RkTab<R> rk;   // Runge-Kutta tableau of the predict step (opts.solver; fixed steps)
  const R* par;  // device block: theta | LQL[d*d] | LQLz[d*d] | H[m*d] | hb[m] | Rm[m*m] | m0[d] | P0[d*d]
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
  R* sm;
  R* sP;
  int* status;
  // reverse sweep only: the forward sweep leaves the six Dormand-Prince slopes of the first ck_smax steps of every interval here
  // ([N][T-1][ck_smax][6][64 P + 8 m]) and the reverse sweep reads them back instead of re-integrating the interval
  R* ck;
  int ck_smax;
  // adaptive solves under the reverse sweep: the forward (workgroup) sweep logs the accepted step sizes of every interval here
  // ([N][T-1][1 + dtlog_cap]: their number, then the sizes) and the reverse sweep replays exactly those steps
  R* dtlog;
  int dtlog_cap;
  // ... and, for the MLP drift, what a right-hand side evaluates between its state and its Jacobian -- per stage of the FIRST step of
  // every interval ([N][T-1][6][ckm_nf][64], kMlpCk*: lane p = hidden unit p): a1, the tangent row T[p][0..7], a2, the Jacobian tile
  // entry, and for state_order 'second' s, E1[p][0..7], td, tq, g.  The reverse sweep's right-hand-side adjoint then starts from these
  // instead of repeating the forward pass (two tanh layers, the 64 x 64 x 9 products): 288 GB of HBM are cheaper than the matrix cores.
  R* ckm;
  int ckm_nf;
  // MLP drift: W2 once more in the parameter block, zero-padded to [64][64] (cdkf_wave8s_kernels.h loads its register slices from it with
  // one lane offset and immediates -- no bounds checks, no clamped addresses); -1: not there
  long o_w2pad;
  // Per-step jumps of the PREDICTED mean (cdkf_ekf_loglik_grad_jumps_*: the linear front-end's dynamics bias / inputs, which the
  // reference adds to the pushed-forward mean without integrating them, continuous_discrete_linear_gaussian_ssm/inference.py:185-205):
  // cj[n][k][:] is added to the mean predicted from t_k to t_{k+1} ([N][T][d], contiguous).  Reverse sweep: gcj[n][k][:] receives the
  // cotangent of that jump, gy[n][k][:] the cotangent of the observation y_k ([N][T][d], [N][T][m]).  All null elsewhere.
  const R* cj;
  R* gcj;
  R* gy;
  // Inputs and time for a drift given as source (f(x, u, t): inference_ekf.py:95, 101-114; the registry drifts ignore both, as the
  // reference's own do): u[n][k][0 .. du-1] at n * u_sn + k * u_sk + i * u_si (null: zeros).  The sweep writes the context of the
  // evaluation in hand into the mutable fields -- every thread the same values, no barrier needed: the offset of this interval's
  // inputs row, the (stage) time, and for the smoother's reverse-time solves the interval's end (t = ctx_tend - s, diffrax_utils.py:13-25)
  const R* u;
  long u_sn, u_sk, u_si;
  int du;
  mutable long ctx_uoff;
  mutable R ctx_t, ctx_tend;
  mutable int ctx_rev;
  // R is diagonal with entries >= 1e-2: the wavefront Lorenz-96 filter then takes the log-likelihood's determinant and quadratic form from
  // psd_solve's factor of S + 1e-9 I (first-order corrections in 1e-9, the next order below 1e-14 of a step's term) instead of a second
  // factorisation (cdkf_wave40_kernels.h, ONE = true)
  int r_diag;
};
constexpr int kCkStep = 6 * 72;  // reals per checkpointed step
// fields of the MLP stage checkpoint (each 64 reals, lane-major): first order kMlpCkFirst of them, 'second' kMlpCkSecond
constexpr int kMlpCkA1 = 0, kMlpCkT = 1, kMlpCkA2 = 9, kMlpCkF = 10, kMlpCkFirst = 11;
constexpr int kMlpCkS = 11, kMlpCkE1 = 12, kMlpCkTd = 20, kMlpCkTq = 21, kMlpCkG = 22, kMlpCkSecond = 23;
constexpr int kAdjCk = 4;         // Dormand-Prince step starts the reverse sweep keeps in LDS per replay chunk

// Integer division by a run-time divisor costs ~40 instructions on CDNA: float reciprocal with +-1 correction
// (exact for 0 <= e < 2^23, 0 < n < 2^12).
__device__ __forceinline__ int fdiv(int e, int n) {
  int q = (int)((float)e * __frcp_rn((float)n));
  const int r = e - q * n;
  q += (r >= n) - (r < 0);
  return q;
}

#define CDKF_WG_FOR(idx, n) for (int idx = threadIdx.x; idx < (n); idx += blockDim.x)

// the time the drift sees for solver time s (WgArgs::ctx_*)
template <typename R>
__device__ __forceinline__ void wg_set_time(const WgArgs<R>& a, R s) {
  a.ctx_t = a.ctx_rev ? a.ctx_tend - s : s;
}

// ---- LDS carve-up ---------------------------------------------------------------------------------------------
// matrices (q x lq each): 0 P, 1 Ps (= X in the update), 2 S, 3 L1 (= S X once the log-likelihood is done), 4 L2,
// [5 F, 6 A if dense Jacobian], [next 2: HP, Hl if !hsel]
// vectors (lq each): 0 m, 1 ms, 2 f, 3 g, 4 v, 5 z, 6 inv1, 7 inv2, 8 y, 9 m_f, 10 f(m_f), 11 tmp
struct WgPlan {
  int nmat, nvec, i_F, i_A, i_HP, i_Hl, extra;
};
__host__ __device__ inline int wg_mlp_scratch(int kind, int d, int h1, int h2) {
  // a1[h1] a2[h2] s2[h2] tq[h1] T[h2*d] Gm[h2*h1] U[h1*d]  (T was sized d * h1 until round 3: with h2 > h1 the tail of U ran into the
  // copy of the weights behind it -- 1-3 % errors in every output; found by scripts/gpu_fuzz_filters.py, every test had h1 >= h2)
  return kind == kDriftMlp ? (2 * h1 + 2 * h2 + h2 * d + h2 * h1 + h1 * d + 4) : 0;
}
__host__ __device__ inline int wg_mlp_theta(int kind, int d, int h1, int h2) {
  return kind == kDriftMlp ? (h1 * d + h1 + h2 * h1 + h2 + d * h2 + d) : 0;
}
__host__ __device__ inline WgPlan wg_plan(int kind, int d, int h1, int h2, int hsel, bool smoother, bool ukf = false) {
  WgPlan p;
  int n = 5;
  const bool dense = (kind != kDriftLorenz96) || smoother || ukf;  // smoother: G = F + aux is dense; UKF: O = c chol(P)
  p.i_F = dense ? n++ : -1;
  p.i_A = dense ? n++ : -1;
  p.i_HP = hsel ? -1 : n++;
  p.i_Hl = hsel ? -1 : n++;
  p.nmat = n;
  p.nvec = 12;
  p.extra = wg_mlp_scratch(kind, d, h1, h2) + wg_mlp_theta(kind, d, h1, h2);
  return p;
}
__host__ __device__ inline long wg_lds_reals(const WgPlan& p, int q, int lq) {
  return (long)p.nmat * q * lq + (long)p.nvec * lq + p.extra;
}

template <typename R>
struct WgLds {
  R* base;
  int msz, lq;
  WgPlan plan;
  __device__ WgLds(R* b, int q, int lq_, const WgPlan& pl) : base(b), msz(q * lq_), lq(lq_), plan(pl) {}
  __device__ R* mat(int i) const { return base + (long)i * msz; }
  __device__ R* vec(int i) const { return base + (long)plan.nmat * msz + (long)i * lq; }
  __device__ R* extra() const { return base + (long)plan.nmat * msz + (long)plan.nvec * lq; }
};

// ---- dense helpers (all operands in LDS, leading dimension lq) ---------------------------------------------------
// C[r x c] = A[r x k] * B[k x c]; 1x4 strips per thread
template <typename R>
__device__ __forceinline__ void wg_matmul(R* __restrict__ C, const R* __restrict__ A, const R* __restrict__ B, int r,
                                          int k, int c, int lq) {
  const int c4 = (c + 3) >> 2;
  CDKF_WG_FOR(e, r * c4) {
    const int i = fdiv(e, c4), j = (e - i * c4) << 2;
    R a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    const R* Ai = A + i * lq;
    const R* Bj = B + j;
#pragma unroll 4
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
    cij[3] = a3;  // strip columns >= c land in the row padding (lq >= roundup4(c) + 1) and are never read as data
  }
}
// C[r x c] = A[r x k] * B^T, B is [c x k]
template <typename R>
__device__ __forceinline__ void wg_matmul_nt(R* __restrict__ C, const R* __restrict__ A, const R* __restrict__ B, int r,
                                             int k, int c, int lq) {
  CDKF_WG_FOR(e, r * c) {
    const int i = fdiv(e, c), j = e - i * c;
    R acc = 0;
#pragma unroll 4
    for (int kk = 0; kk < k; ++kk) acc = rfma(A[i * lq + kk], B[j * lq + kk], acc);
    C[i * lq + j] = acc;
  }
}
// C[r x c] = A^T * B, A is [k x r], B is [k x c]; 1x4 strips
template <typename R>
__device__ __forceinline__ void wg_matmul_tn(R* __restrict__ C, const R* __restrict__ A, const R* __restrict__ B, int r,
                                             int k, int c, int lq) {
  const int c4 = (c + 3) >> 2;
  CDKF_WG_FOR(e, r * c4) {
    const int i = fdiv(e, c4), j = (e - i * c4) << 2;
    R a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll 4
    for (int kk = 0; kk < k; ++kk) {
      const R f = A[kk * lq + i];
      const R* b = B + kk * lq + j;
      a0 = rfma(f, b[0], a0);
      a1 = rfma(f, b[1], a1);
      a2 = rfma(f, b[2], a2);
      a3 = rfma(f, b[3], a3);
    }
    R* cij = C + i * lq + j;
    cij[0] = a0;
    cij[1] = a1;
    cij[2] = a2;
    cij[3] = a3;
  }
}

// ---- matrix-core products: C (op)= op(A) op(B) on 16 x 16 output tiles, one tile per wavefront at a time ---------------
// v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 (exact f64 / f32 arithmetic).  On MI355X the f64 MFMA rate equals the
// vector FMA rate; the gain is operand traffic: a k-step of a tile (1024 FMAs) takes two LDS reads per lane, where the
// 1 x 4 register strips above take five per four FMAs and were LDS-bound.  Lane l feeds A[l & 15][k0 + (l >> 4)] and
// B[k0 + (l >> 4)][l & 15]; results: column l & 15, rows (l >> 4) + 4 r (f64) or 4 (l >> 4) + r (f32), r = 0..3
// (cdna_hip_programming.md, "Fragment layout").  Out-of-range rows / columns / k are fed zeros.
typedef double wg_f64x4 __attribute__((ext_vector_type(4)));
typedef float wg_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ wg_f64x4 wg_mfma(double a, double b, wg_f64x4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ wg_f32x4 wg_mfma(float a, float b, wg_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
template <typename R>
struct WgAcc;
template <>
struct WgAcc<double> {
  typedef wg_f64x4 type;
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <>
struct WgAcc<float> {
  typedef wg_f32x4 type;
  static __device__ __forceinline__ int row(int lane, int r) { return 4 * (lane >> 4) + r; }
};
// TA: A is stored transposed (A'(i,k) = A[k][i]); TB: B is stored transposed (B'(k,j) = B[j][k]); SUB: C -= product
template <typename R, bool TA, bool TB, bool SUB>
__device__ __forceinline__ void wg_mm(R* __restrict__ C, const R* __restrict__ A, const R* __restrict__ B, int r, int kdim,
                                      int c, int lq) {
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6, lane = threadIdx.x & 63;
  const int tc = (c + 15) >> 4, ntile = ((r + 15) >> 4) * tc;
  for (int tile = wave; tile < ntile; tile += nw) {
    const int ti = fdiv(tile, tc), tj = tile - ti * tc;
    const int ai = ti * 16 + (lane & 15), bj = tj * 16 + (lane & 15), kq = lane >> 4;
    const bool aok = ai < r, bok = bj < c;
    typename WgAcc<R>::type acc = {0, 0, 0, 0};
    for (int k0 = 0; k0 < kdim; k0 += 4) {
      const int k = k0 + kq;
      const bool kok = k < kdim;
      const R av = (aok && kok) ? (TA ? A[k * lq + ai] : A[ai * lq + k]) : R(0);
      const R bv = (bok && kok) ? (TB ? B[bj * lq + k] : B[k * lq + bj]) : R(0);
      acc = wg_mfma(av, bv, acc);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = ti * 16 + WgAcc<R>::row(lane, q);
      if (row < r && bok) {
        R* p = C + row * lq + bj;
        *p = SUB ? *p - acc[q] : acc[q];
      }
    }
  }
}

// Register copy of an 8 x 8 diagonal block (lower triangle) of an LDS matrix, padded with the identity beyond nb:
// all loads are independent, so their LDS latency overlaps -- the serial parts of the blocked algorithms below then
// run out of registers instead of chasing ~100-cycle ds_read round trips.
template <typename R>
__device__ __forceinline__ void wg_load_block(const R* S, int kb, int nb, int lq, R (&Lr)[kWgBlock][kWgBlock]) {
#pragma unroll
  for (int i = 0; i < kWgBlock; ++i)
#pragma unroll
    for (int k = 0; k <= i; ++k)
      Lr[i][k] = (i < nb) ? S[(kb + i) * lq + kb + k] : (i == k ? R(1) : R(0));
}

// Blocked right-looking Cholesky (lower, in place) of up to TWO n x n matrices at once (S2 may be null): the LL
// factor and the boosted gain factor of one update are independent.  inv*[j] = 1 / L[j][j].  3 barriers per
// 8-wide panel.  Non-positive pivots give NaN (jnp.linalg.cholesky semantics) and raise *bad.
template <typename R>
__device__ void wg_cholesky2(R* S1, R* inv1, R* S2, R* inv2, int n, int lq, int* bad) {
  const int nmat = S2 ? 2 : 1;
  for (int kb = 0; kb < n; kb += kWgBlock) {
    const int nb = (n - kb < kWgBlock) ? n - kb : kWgBlock;
    __syncthreads();
    // (a) diagonal block, factored in the registers of one thread per matrix (threads 0 and 64: different wavefronts)
    const int who = (blockDim.x > 64) ? 64 : 1;
    if (threadIdx.x == 0 || (nmat == 2 && threadIdx.x == who)) {
      R* S = (threadIdx.x == 0) ? S1 : S2;
      R* inv = (threadIdx.x == 0) ? inv1 : inv2;
      R Lr[kWgBlock][kWgBlock], rv[kWgBlock];
      wg_load_block(S, kb, nb, lq, Lr);
      bool nonpd = false;
#pragma unroll
      for (int j = 0; j < kWgBlock; ++j) {
        R sj = Lr[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) sj = rfma(-Lr[j][k], Lr[j][k], sj);
        nonpd = nonpd || !(sj > R(0));
        const R r = rrsqrt(sj);
        rv[j] = r;
        Lr[j][j] = sj * r;
#pragma unroll
        for (int i2 = j + 1; i2 < kWgBlock; ++i2) {
          R v = Lr[i2][j];
#pragma unroll
          for (int k = 0; k < j; ++k) v = rfma(-Lr[i2][k], Lr[j][k], v);
          Lr[i2][j] = v * r;
        }
      }
      if (nonpd) *bad = 1;
#pragma unroll
      for (int i2 = 0; i2 < kWgBlock; ++i2)
        if (i2 < nb) {
          inv[kb + i2] = rv[i2];
#pragma unroll
          for (int k = 0; k <= i2; ++k) S[(kb + i2) * lq + kb + k] = Lr[i2][k];
        }
    }
    __syncthreads();
    // (b) panel below the block: row r solves  X L_blk^T = S[r][kb:kb+nb], out of registers
    const int rest = n - kb - nb;
    CDKF_WG_FOR(e, rest * nmat) {
      const int w = (e >= rest) ? 1 : 0;
      const int r = kb + nb + (e - w * rest);
      R* S = w ? S2 : S1;
      const R* inv = w ? inv2 : inv1;
      R Lr[kWgBlock][kWgBlock], row[kWgBlock], rv[kWgBlock];
      wg_load_block(S, kb, nb, lq, Lr);
#pragma unroll
      for (int j = 0; j < kWgBlock; ++j) {
        row[j] = (j < nb) ? S[r * lq + kb + j] : R(0);
        rv[j] = (j < nb) ? inv[kb + j] : R(1);
      }
#pragma unroll
      for (int j = 0; j < kWgBlock; ++j) {
        R v = row[j];
#pragma unroll
        for (int k = 0; k < j; ++k) v = rfma(-row[k], Lr[j][k], v);
        row[j] = v * rv[j];
      }
#pragma unroll
      for (int j = 0; j < kWgBlock; ++j)
        if (j < nb) S[r * lq + kb + j] = row[j];
    }
    __syncthreads();
    // (c) trailing update S[a][b] -= sum_k L[a][k] L[b][k] as a rank-nb matrix-core product over the whole trailing square
    //     (the upper triangle receives values nobody reads: every consumer takes the lower one)
    if (rest > 0) {
      R* T1 = S1 + (kb + nb) * lq;
      wg_mm<R, false, true, true>(T1 + kb + nb, T1 + kb, T1 + kb, rest, nb, rest, lq);
      if (nmat == 2) {
        R* T2 = S2 + (kb + nb) * lq;
        wg_mm<R, false, true, true>(T2 + kb + nb, T2 + kb, T2 + kb, rest, nb, rest, lq);
      }
    }
  }
  __syncthreads();
}

// ---- wavefront helpers ----------------------------------------------------------------------------------------------------
// (A single-wavefront Cholesky / substitution with the matrix in LDS was tried for the 40 x 40 update: no workgroup
//  barriers, but one wavefront cannot issue LDS reads fast enough -- 84 us against 58 us for the blocked versions below.)
template <typename R>
__device__ __forceinline__ R wave_bcast(R x, int src);
template <>
__device__ __forceinline__ float wave_bcast<float>(float x, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), src));
}
template <>
__device__ __forceinline__ double wave_bcast<double>(double x, int src) {
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
  return __longlong_as_double((long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo));  // (unsigned: a left shift of a negative value is undefined before C++20)
}
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Solve (L L^T) X = B in place, B is [n x c]; blocked substitution, 2 barriers per panel and direction; the
// 8 x 8 triangular solves run out of registers.
template <typename R>
__device__ void wg_chol_solve(const R* L, const R* inv, R* B, int n, int c, int lq) {
  for (int kb = 0; kb < n; kb += kWgBlock) {  // forward: L Y = B
    const int nb = (n - kb < kWgBlock) ? n - kb : kWgBlock;
    __syncthreads();
    CDKF_WG_FOR(j, c) {
      R Lr[kWgBlock][kWgBlock], col[kWgBlock], rv[kWgBlock];
      wg_load_block(L, kb, nb, lq, Lr);
#pragma unroll
      for (int i = 0; i < kWgBlock; ++i) {
        col[i] = (i < nb) ? B[(kb + i) * lq + j] : R(0);
        rv[i] = (i < nb) ? inv[kb + i] : R(1);
      }
#pragma unroll
      for (int i = 0; i < kWgBlock; ++i) {
        R v = col[i];
#pragma unroll
        for (int k = 0; k < i; ++k) v = rfma(-Lr[i][k], col[k], v);
        col[i] = v * rv[i];
      }
#pragma unroll
      for (int i = 0; i < kWgBlock; ++i)
        if (i < nb) B[(kb + i) * lq + j] = col[i];
    }
    __syncthreads();
    const int rest = n - kb - nb;
    if (rest > 0) wg_mm<R, false, false, true>(B + (kb + nb) * lq, L + (kb + nb) * lq + kb, B + kb * lq, rest, nb, c, lq);
  }
  const int nblk = (n + kWgBlock - 1) / kWgBlock;
  for (int bi = nblk - 1; bi >= 0; --bi) {  // backward: L^T X = Y
    const int kb = bi * kWgBlock;
    const int nb = (n - kb < kWgBlock) ? n - kb : kWgBlock;
    __syncthreads();
    CDKF_WG_FOR(j, c) {
      R Lr[kWgBlock][kWgBlock], col[kWgBlock], rv[kWgBlock];
      wg_load_block(L, kb, nb, lq, Lr);
#pragma unroll
      for (int i = 0; i < kWgBlock; ++i) {
        col[i] = (i < nb) ? B[(kb + i) * lq + j] : R(0);
        rv[i] = (i < nb) ? inv[kb + i] : R(1);
      }
#pragma unroll
      for (int i = kWgBlock - 1; i >= 0; --i) {
        R v = col[i];
#pragma unroll
        for (int k = i + 1; k < kWgBlock; ++k) v = rfma(-Lr[k][i], col[k], v);
        col[i] = v * rv[i];
      }
#pragma unroll
      for (int i = 0; i < kWgBlock; ++i)
        if (i < nb) B[(kb + i) * lq + j] = col[i];
    }
    __syncthreads();
    if (kb > 0) wg_mm<R, true, false, true>(B, L + kb * lq, B + kb * lq, kb, nb, c, lq);
  }
  __syncthreads();
}

// ---- drifts ------------------------------------------------------------------------------------------------------
template <typename R>
__device__ __forceinline__ R rtanh(R x) {
  return (R)tanh((double)x);
}
template <>
__device__ __forceinline__ float rtanh<float>(float x) {
  return tanhf(x);
}
// fp64 tanh in ~35 instructions (the library routine is ~130, and a right-hand side of the MLP drift evaluates two per hidden unit):
//   u = expm1(2|x|) = 2^k p + (2^k - 1),  2|x| = k ln 2 + r,  |r| <= ln 2 / 2,  p = expm1(r) by its Taylor sum to r^13 (4e-18 relative),
//   tanh|x| = u / (u + 2)  -- no cancellation anywhere (k = 0: u = p carries the full relative accuracy of small arguments);
// the quotient by v_rcp_f64 + two Newton steps + one residual correction.  Measured against tanhl on 2e7 arguments
// (|x| < 20, 1e-12 .. 1): 2.6 ulp at worst; NaN in, NaN out; +-inf -> +-1; signed zeros and denormals pass through.
// Used by the wavefront-per-trajectory kernels (cdkf_wave8_kernels.h, cdkf_adjoint_kernels.h), where the whole kernel is ONE inlined
// function.  The workgroup kernels below keep the library routine (rtanh): they really CALL wg_drift / wg_cholesky2 / wg_chol_solve
// (s_swappc), and with this routine inlined into wg_drift the EPT = 1 instantiations returned garbage on gfx950 / ROCm 7.2 -- the
// Lorenz-96 one, which never evaluates a tanh, included (A/B of two libraries differing in this function only, scripts/dbg_wg.py;
// tests/test_gpu_wg.py::test_workgroup_kernels_other_runge_kutta_methods is the guard).
template <typename R>
__device__ __forceinline__ R rtanh_fast(R x) {
  double ax = __builtin_fabs((double)x);
  ax = (ax > 20.0) ? 20.0 : ax;  // (tanh 20 = 1 - 8e-18; a NaN fails the comparison and travels on)
  const double t = ax + ax;
  const double kf = __builtin_rint(t * 1.4426950408889634);
  double r = __builtin_fma(kf, -6.93147180369123816490e-01, t);
  r = __builtin_fma(kf, -1.90821492927058770002e-10, r);
  double q = 1.0 / 6227020800.0;
  q = __builtin_fma(q, r, 1.0 / 479001600.0);
  q = __builtin_fma(q, r, 1.0 / 39916800.0);
  q = __builtin_fma(q, r, 1.0 / 3628800.0);
  q = __builtin_fma(q, r, 1.0 / 362880.0);
  q = __builtin_fma(q, r, 1.0 / 40320.0);
  q = __builtin_fma(q, r, 1.0 / 5040.0);
  q = __builtin_fma(q, r, 1.0 / 720.0);
  q = __builtin_fma(q, r, 1.0 / 120.0);
  q = __builtin_fma(q, r, 1.0 / 24.0);
  q = __builtin_fma(q, r, 1.0 / 6.0);
  q = __builtin_fma(q, r, 0.5);
  const double p = __builtin_fma(q * r, r, r);
  const double s = __builtin_ldexp(1.0, (int)kf);
  const double u = __builtin_fma(s, p, s - 1.0);
  const double den = u + 2.0;
  double y = __builtin_amdgcn_rcp(den);
  y = __builtin_fma(__builtin_fma(-den, y, 1.0), y, y);
  y = __builtin_fma(__builtin_fma(-den, y, 1.0), y, y);
  double res = u * y;
  res = __builtin_fma(__builtin_fma(-den, res, u), y, res);
  return (R)__builtin_copysign(res, (double)x);
}
template <>
__device__ __forceinline__ float rtanh_fast<float>(float x) {
  return tanhf(x);
}

// MLP constants in LDS (once per workgroup): the weights, and G_pq = (sum_i W3_ip W1_qi) W2_pq for grad(div f)
template <typename R>
__device__ void wg_mlp_prepare(const WgArgs<R>& a, const WgLds<R>& L) {
  if (a.kind != kDriftMlp) return;
  const int d = a.d, h1 = a.h1, h2 = a.h2;
  R* thl = L.extra() + wg_mlp_scratch(kDriftMlp, d, h1, h2);
  CDKF_WG_FOR(e, wg_mlp_theta(kDriftMlp, d, h1, h2)) thl[e] = (a.par + a.o_theta)[e];
  __syncthreads();
  const R* W1 = thl;
  const R* W2 = W1 + h1 * d + h1;
  const R* W3 = W2 + h2 * h1 + h2;
  R* Gm = L.extra() + (2 * h1 + 2 * h2 + h2 * d);
  CDKF_WG_FOR(e, h2 * h1) {
    const int p = fdiv(e, h1), qq = e - p * h1;
    R s = 0;
    for (int i = 0; i < d; ++i) s = rfma(W3[i * h2 + p], W1[qq * d + i], s);
    Gm[e] = s * W2[e];
  }
  __syncthreads();
}

#ifdef CDKF_WG_CUSTOM
// A drift given as C source (state dimension above the register-resident kernels' six): this header is then compiled at run time
// (launch_custom.hip) together with the definitions of these two, which differentiate the source by dual numbers (cdkf_dual.h):
//   wg_custom_drift: f(x) -> fv, the dense Jacobian -> F (if non-null), grad(div f) -> gv (if non-null); no barrier at the end;
//   wg_custom_sigma: the drift at the 2 d + 1 sigma points m, m +- O[:, i] -> f0, DF[:, i] = f+ - f-, foo[:, i] = f+ + f-.
template <typename R>
__device__ void wg_custom_drift(const WgArgs<R>& a, const WgLds<R>& L, const R* x, R* fv, R* F, R* gv);
template <typename R>
__device__ void wg_custom_sigma(const WgArgs<R>& a, const WgLds<R>& L, const R* ms, const R* O, R* f0, R* DF, R* foo);
#endif

// f(x) -> fv; dense Jacobian -> F (if F != null); g = grad(div f) -> gv (MLP and custom drifts, if gv != null).  Ends with a
// barrier.  Lorenz-96 callers that only need f and use the banded product pass F = null.
template <typename R>
__device__ void wg_drift(const WgArgs<R>& a, const WgLds<R>& L, const R* __restrict__ x, R* __restrict__ fv,
                         R* __restrict__ F, R* __restrict__ gv) {
  const int d = a.d, lq = a.lq;
#ifdef CDKF_WG_CUSTOM
  if (a.kind >= kDriftCustomBase) {
    wg_custom_drift<R>(a, L, x, fv, F, gv);
    __syncthreads();
    return;
  }
#endif
  if (a.kind == kDriftLinear) {
    const R* th = a.par + a.o_theta;
    CDKF_WG_FOR(i, d) {
      R s = 0;
      for (int j = 0; j < d; ++j) s = rfma(th[i * d + j], x[j], s);
      fv[i] = s + th[d * d + i];
    }
    if (F) CDKF_WG_FOR(e, d * d) {
        const int i = fdiv(e, d);
        F[i * lq + (e - i * d)] = th[e];
      }
  } else if (a.kind == kDriftLorenz63) {
    const R* th = a.par + a.o_theta;
    if (threadIdx.x == 0) {
      fv[0] = th[0] * (x[1] - x[0]);
      fv[1] = x[0] * (th[1] - x[2]) - x[1];
      fv[2] = x[0] * x[1] - th[2] * x[2];
      if (F) {
        F[0] = -th[0]; F[1] = th[0]; F[2] = 0;
        F[lq] = th[1] - x[2]; F[lq + 1] = -1; F[lq + 2] = -x[0];
        F[2 * lq] = x[1]; F[2 * lq + 1] = x[0]; F[2 * lq + 2] = -th[2];
      }
    }
  } else if (a.kind == kDriftLorenz96) {
    const R forcing = (a.par + a.o_theta)[0];
    CDKF_WG_FOR(i, d) {
      const int ip1 = (i + 1 == d) ? 0 : i + 1, im1 = (i == 0) ? d - 1 : i - 1, im2 = (im1 == 0) ? d - 1 : im1 - 1;
      fv[i] = rfma(x[ip1] - x[im2], x[im1], forcing - x[i]);
    }
    if (F) {
      CDKF_WG_FOR(e, d * d) {
        const int i = fdiv(e, d), j = e - i * d;
        const int ip1 = (i + 1 == d) ? 0 : i + 1, im1 = (i == 0) ? d - 1 : i - 1, im2 = (im1 == 0) ? d - 1 : im1 - 1;
        R v = 0;
        if (j == ip1) v = x[im1];
        if (j == im2) v = -x[im1];
        if (j == im1) v = x[ip1] - x[im2];
        if (j == i) v = R(-1);
        F[i * lq + j] = v;
      }
    }
  } else {  // MLP: f = W3 tanh(W2 tanh(W1 x + b1) + b2) + b3;  J = W3 D2 W2 D1 W1 via the tangent T = D2 W2 D1 W1
    const int h1 = a.h1, h2 = a.h2;
    const R* W1 = L.extra() + wg_mlp_scratch(kDriftMlp, d, h1, h2);
    const R* b1 = W1 + h1 * d;
    const R* W2 = b1 + h1;
    const R* b2 = W2 + h2 * h1;
    const R* W3 = b2 + h2;
    const R* b3 = W3 + d * h2;
    R* a1 = L.extra();   // [h1]
    R* a2 = a1 + h1;     // [h2]
    R* s2 = a2 + h2;     // [h2]
    R* tq = s2 + h2;     // [h1]
    R* Tm = tq + h1;     // [h2 x d] tangent of the second layer (stored as h2 rows of d)
    R* Gm = Tm + h2 * d; // [h2 x h1]
    // layer 1: a1 = tanh(W1 x + b1) and, for the Jacobian, U[q][:] = (1 - a1_q^2) W1[q][:]   (U lives in Tm's tail)
    R* Um = Gm + h2 * h1;  // [h1 x d]
    CDKF_WG_FOR(p, h1) {
      R s = b1[p];
      for (int j = 0; j < d; ++j) s = rfma(W1[p * d + j], x[j], s);
      const R t = rtanh(s);
      a1[p] = t;
      if (F) {
        const R d1 = R(1) - t * t;
        for (int j = 0; j < d; ++j) Um[p * d + j] = d1 * W1[p * d + j];
      }
    }
    __syncthreads();
    // layer 2, one row p and one 4-column strip per thread: z2_p = W2[p] . a1 + b2_p (every strip recomputes it),
    // T[p][j..j+3] = W2[p] . U[:, j..j+3]; stored already scaled by d2_p = 1 - tanh(z2_p)^2
    {
      const int d4 = F ? ((d + 3) >> 2) : 1;
      CDKF_WG_FOR(e, h2 * d4) {
        const int p = fdiv(e, d4), j = (e - p * d4) << 2;
        R z = b2[p], t0 = 0, t1 = 0, t2 = 0, t3 = 0;
        const R* w = W2 + p * h1;
        if (F) {
#pragma unroll 4
          for (int qq = 0; qq < h1; ++qq) {
            const R wq = w[qq];
            const R* u = Um + qq * d + j;
            z = rfma(wq, a1[qq], z);
            t0 = rfma(wq, u[0], t0);
            t1 = rfma(wq, u[1], t1);
            t2 = rfma(wq, u[2], t2);
            t3 = rfma(wq, u[3], t3);
          }
        } else {
#pragma unroll 4
          for (int qq = 0; qq < h1; ++qq) z = rfma(w[qq], a1[qq], z);
        }
        const R av = rtanh(z);
        if (j == 0) a2[p] = av;
        if (F) {
          const R d2 = R(1) - av * av;
          R* tp = Tm + p * d + j;
          tp[0] = d2 * t0;
          if (j + 1 < d) tp[1] = d2 * t1;
          if (j + 2 < d) tp[2] = d2 * t2;
          if (j + 3 < d) tp[3] = d2 * t3;
        }
      }
    }
    __syncthreads();
    // layer 3: f = W3 a2 + b3;  J = W3 (D2 T)
    CDKF_WG_FOR(e, d * (d + 1)) {
      const int i = fdiv(e, d + 1), j = e - i * (d + 1);
      if (j == d) {
        R s = b3[i];
#pragma unroll 4
        for (int p = 0; p < h2; ++p) s = rfma(W3[i * h2 + p], a2[p], s);
        fv[i] = s;
      } else if (F) {
        R s = 0;
#pragma unroll 4
        for (int p = 0; p < h2; ++p) s = rfma(W3[i * h2 + p], Tm[p * d + j], s);
        F[i * lq + j] = s;
      }
    }
    if (gv) {
      // g = d tr(J)/dx, tr(J) = sum_pq G_pq d2_p d1_q
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
  __syncthreads();
}

// ---- per-thread ownership of covariance entries ------------------------------------------------------------------
template <typename R, int EPT>
struct Own {
  int n;         // number of owned entries (<= EPT): entry u is e = threadIdx.x + u * blockDim.x < d * d
  int d_, lq_;   // uniform
  R lql[EPT];    // (L Qc L^T)[i][j]
  // row / column / LDS offset of entry u, re-derived at each use (a handful of integer instructions): kept in registers
  // they were the values the allocator spilled, and every Runge-Kutta stage re-loaded them from scratch
  __device__ __forceinline__ void at(int u, int& i, int& j) const {
    int t = threadIdx.x;
    CDKF_OPAQUE("+v"(t));  // opaque: keeps the optimiser from hoisting these few integer ops out of the Runge-Kutta
                                 // loops, where their results (and everything derived from them) were spilled to scratch
    const int e = t + u * blockDim.x;
    i = fdiv(e, d_);
    j = e - i * d_;
  }
  __device__ __forceinline__ int off(int u) const {
    int i, j;
    at(u, i, j);
    return i * lq_ + j;
  }
  __device__ void init(int d, int lq, const R* LQL) {
    n = 0;
    d_ = d;
    lq_ = lq;
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
      const int e = threadIdx.x + u * blockDim.x;
      const bool ok = e < d * d;
      lql[u] = ok ? LQL[e] : R(0);
      n += ok;
    }
  }
};

// One Dormand-Prince step with register-resident slopes.  `rhs(msrc, Psrc, km, kP)` must fill this thread's slopes from
// the LDS stage value and END with a barrier-free state (it may use barriers inside).  The six stages are separate
// instantiations (wg_stage<S>) rather than an unrolled loop: when the optimiser declined to unroll the loop around the
// inlined right-hand side, the slope arrays were indexed dynamically and moved to scratch memory (5x slower sweeps).
template <int S, typename R, int EPT, typename RhsFn>
__device__ __forceinline__ void wg_stage(const WgLds<R>& L, const Own<R, EPT>& own, int d, R dt, bool with_P, RhsFn& rhs,
                                         R (&kM)[6], R (&kP)[6][EPT], const RkTab<R>& tb, const WgArgs<R>& a, R tstep) {
  if (S >= tb.stages) {  // methods with fewer stages (uniform over the workgroup): no slope, no barrier
    kM[S] = R(0);
#pragma unroll
    for (int u = 0; u < EPT; ++u) kP[S][u] = R(0);
    return;
  }
#ifdef CDKF_WG_CUSTOM
  wg_set_time(a, rfma(rk_stage_c(tb, S), dt, tstep));  // stage time t + c_S dt for f(x, u, t)
#endif
  R* mcur = L.vec(0);
  R* Pm = L.mat(0);
  R* ms = L.vec(1);
  R* Ps = L.mat(1);
  if constexpr (S == 0) {
    rhs(mcur, Pm, kM[0], kP[0]);
  } else {
    if (threadIdx.x < d) {
      R acc = tb.a[S][0] * kM[0];
#pragma unroll
      for (int j = 1; j < S; ++j) acc = rfma(tb.a[S][j], kM[j], acc);
      ms[threadIdx.x] = rfma(dt, acc, mcur[threadIdx.x]);
    }
    if (with_P) {
#pragma unroll
      for (int u = 0; u < EPT; ++u)
        if (u < own.n) {
          R acc = tb.a[S][0] * kP[0][u];
#pragma unroll
          for (int j = 1; j < S; ++j) acc = rfma(tb.a[S][j], kP[j][u], acc);
          {
            const int o_ = own.off(u);
            Ps[o_] = rfma(dt, acc, Pm[o_]);
          }
        }
    }
    __syncthreads();
    rhs(ms, Ps, kM[S], kP[S]);
  }
  __syncthreads();  // every thread is done reading the stage value before it is overwritten
}

template <typename R, int EPT, typename RhsFn>
__device__ __forceinline__ void wg_dopri5_step(const WgLds<R>& L, const Own<R, EPT>& own, int d, R dt, bool with_P,
                                               RhsFn rhs, const RkTab<R>& tb, const WgArgs<R>& a, R tstep) {
  R* mcur = L.vec(0);
  R* Pm = L.mat(0);
  R kM[6];
  R kP[6][EPT];
#ifdef CDKF_WG_ZERO_INIT  // (diagnostic build, scripts/bisect_wg8.sh: does an uninitialised slope slot reach a result?)
#pragma unroll
  for (int s_ = 0; s_ < 6; ++s_) {
    kM[s_] = R(0);
#pragma unroll
    for (int u = 0; u < EPT; ++u) kP[s_][u] = R(0);
  }
#endif
  wg_stage<0>(L, own, d, dt, with_P, rhs, kM, kP, tb, a, tstep);
  wg_stage<1>(L, own, d, dt, with_P, rhs, kM, kP, tb, a, tstep);
  wg_stage<2>(L, own, d, dt, with_P, rhs, kM, kP, tb, a, tstep);
  wg_stage<3>(L, own, d, dt, with_P, rhs, kM, kP, tb, a, tstep);
  wg_stage<4>(L, own, d, dt, with_P, rhs, kM, kP, tb, a, tstep);
  wg_stage<5>(L, own, d, dt, with_P, rhs, kM, kP, tb, a, tstep);
  if (threadIdx.x < d) {
    R acc = tb.b[0] * kM[0];
#pragma unroll
    for (int sg = 1; sg < 6; ++sg) acc = rfma(tb.b[sg], kM[sg], acc);
    mcur[threadIdx.x] = rfma(dt, acc, mcur[threadIdx.x]);
  }
  if (with_P) {
#pragma unroll
    for (int u = 0; u < EPT; ++u)
      if (u < own.n) {
        R acc = tb.b[0] * kP[0][u];
#pragma unroll
        for (int sg = 1; sg < 6; ++sg) acc = rfma(tb.b[sg], kP[sg][u], acc);
        {
          const int o_ = own.off(u);
          Pm[o_] = rfma(dt, acc, Pm[o_]);
        }
      }
  }
  __syncthreads();
}

// Adaptive steps for the workgroup kernels: diffrax.PIDController around the embedded pair, as integrate_adaptive (cdkf_math.h) does
// it per lane -- here the error estimate's RMS runs over the mean and the FULL d x d covariance the workgroup's threads own between
// them (wavefront sums through __shfl_xor, their partial sums through LDS in a fixed order: every thread then holds the same number
// and takes the same accept / reject decision).  A rejected step leaves the state where it was: the candidate lives in registers
// until it is accepted.  max_steps counts attempts.
template <typename R, int EPT, typename RhsFn>
__device__ __forceinline__ bool wg_integrate_adaptive(const WgLds<R>& L, const Own<R, EPT>& own, int d, R t0, R t1, R dt0,
                                                      long max_steps, bool with_P, RhsFn rhs, const RkTab<R>& tb, const WgArgs<R>& a,
                                                      R* dtlog = nullptr, int dtlog_cap = 0) {
  __shared__ double red[16];
  R* mcur = L.vec(0);
  R* Pm = L.mat(0);
  R* ms = L.vec(1);
  R* Ps = L.mat(1);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  const R count = with_P ? R(d) + R(d) * R(d) : R(d);
  R tprev = t0;
  const R dt_first = rmin(dt0, tb.dtmax);  // (the first size clipped to [dtmin, dtmax], a step at dtmin kept -- integrate_adaptive, cdkf_math.h)
  bool at_min = dt_first <= tb.dtmin;
  R tnext = rmin(t0 + rmax(dt_first, tb.dtmin), t1);
  R inv1 = R(1), inv2 = R(1);
  long steps = 0;
  int nacc = 0;  // accepted steps (logged for the reverse sweep)
  while (tprev < t1) {  // uniform over the workgroup
    if (steps >= max_steps) {
      if (dtlog && threadIdx.x == 0) dtlog[0] = R(nacc);
      return true;
    }
    const R dt = tnext - tprev;
    R kM[6];
    R kP[6][EPT];
    wg_stage<0>(L, own, d, dt, with_P, rhs, kM, kP, tb, a, tprev);
    wg_stage<1>(L, own, d, dt, with_P, rhs, kM, kP, tb, a, tprev);
    wg_stage<2>(L, own, d, dt, with_P, rhs, kM, kP, tb, a, tprev);
    wg_stage<3>(L, own, d, dt, with_P, rhs, kM, kP, tb, a, tprev);
    wg_stage<4>(L, own, d, dt, with_P, rhs, kM, kP, tb, a, tprev);
    wg_stage<5>(L, own, d, dt, with_P, rhs, kM, kP, tb, a, tprev);
    // the candidate
    R ynM = R(0), ynP[EPT];
    if (threadIdx.x < d) {
      R acc = tb.b[0] * kM[0];
#pragma unroll
      for (int sg = 1; sg < 6; ++sg) acc = rfma(tb.b[sg], kM[sg], acc);
      ynM = rfma(dt, acc, mcur[threadIdx.x]);
    }
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
      ynP[u] = R(0);
      if (with_P && u < own.n) {
        R acc = tb.b[0] * kP[0][u];
#pragma unroll
        for (int sg = 1; sg < 6; ++sg) acc = rfma(tb.b[sg], kP[sg][u], acc);
        ynP[u] = rfma(dt, acc, Pm[own.off(u)]);
      }
    }
    // first-same-as-last stage f(y_new) of the embedded estimate
    R k7M = R(0), k7P[EPT];
#pragma unroll
    for (int u = 0; u < EPT; ++u) k7P[u] = R(0);
    if (tb.fsal) {
      if (threadIdx.x < d) ms[threadIdx.x] = ynM;
      if (with_P) {
#pragma unroll
        for (int u = 0; u < EPT; ++u)
          if (u < own.n) Ps[own.off(u)] = ynP[u];
      }
      __syncthreads();
#ifdef CDKF_WG_CUSTOM
      wg_set_time(a, tprev + dt);
#endif
      rhs(ms, Ps, k7M, k7P);
      __syncthreads();
    }
    double sq = 0.0;
    if (threadIdx.x < d) {
      R err = tb.berr[6] * k7M;
#pragma unroll
      for (int sg = 0; sg < 6; ++sg) err = rfma(tb.berr[sg], kM[sg], err);
      const R sc = (dt * err) / rfma(rmax(rabs(mcur[threadIdx.x]), rabs(ynM)), tb.rtol, tb.atol);
      sq += (double)(sc * sc);
    }
    if (with_P) {
#pragma unroll
      for (int u = 0; u < EPT; ++u)
        if (u < own.n) {
          R err = tb.berr[6] * k7P[u];
#pragma unroll
          for (int sg = 0; sg < 6; ++sg) err = rfma(tb.berr[sg], kP[sg][u], err);
          const R sc = (dt * err) / rfma(rmax(rabs(Pm[own.off(u)]), rabs(ynP[u])), tb.rtol, tb.atol);
          sq += (double)(sc * sc);
        }
    }
#pragma unroll
    for (int o_ = 32; o_ >= 1; o_ >>= 1) sq += __shfl_xor(sq, o_);
    if (lane == 0) red[wv] = sq;
    __syncthreads();
    double tot = 0.0;
    for (int w = 0; w < nw; ++w) tot += red[w];
    const R scaled = rsqrt_((R)tot / count);
    const bool keep = scaled < R(1) || at_min;
    const R inv = (scaled == R(0)) ? R(__builtin_huge_val()) : R(1) / scaled;
    R factor = tb.safety * rpow(inv, tb.c1);
    if (tb.c2 != R(0)) factor *= rpow(inv1, tb.c2);
    if (tb.c3 != R(0)) factor *= rpow(inv2, tb.c3);
    factor = rmin(rmax(factor, keep ? R(1) : tb.fmin), tb.fmax);  // (fmax / fmin: a NaN estimate rejects and shrinks, cdkf_math.h)
    const R nt0 = keep ? tnext : tprev;
    R dtn = rmin(dt * factor, tb.dtmax);
    at_min = dtn <= tb.dtmin;
    dtn = rmax(dtn, tb.dtmin);
    const R nt1 = nt0 + dtn;
    if (keep) {
      if (threadIdx.x < d) mcur[threadIdx.x] = ynM;
      if (with_P) {
#pragma unroll
        for (int u = 0; u < EPT; ++u)
          if (u < own.n) Pm[own.off(u)] = ynP[u];
      }
      inv2 = inv1;
      inv1 = inv;
      if (dtlog && threadIdx.x == 0 && nacc < dtlog_cap) dtlog[1 + nacc] = dt;
      ++nacc;
    }
    __syncthreads();  // the state is complete (and `red` free) before the next attempt
    tprev = rmin(nt0, t1);
    tnext = (nt1 > t1 - Tol<R>::v) ? (keep ? t1 : rfma(R(0.5), t1 - tprev, tprev)) : nt1;
    ++steps;
  }
  if (dtlog && threadIdx.x == 0) dtlog[0] = R(nacc);
  return false;
}

template <typename R, int EPT, typename RhsFn>
__device__ __forceinline__ bool wg_integrate(const WgLds<R>& L, const Own<R, EPT>& own, int d, R t0, R t1, R dt0,
                                             long max_steps, bool with_P, RhsFn rhs, const RkTab<R>& tb, const WgArgs<R>& a,
                                             R* dtlog = nullptr, int dtlog_cap = 0) {
  if (tb.adaptive) return wg_integrate_adaptive<R, EPT>(L, own, d, t0, t1, dt0, max_steps, with_P, rhs, tb, a, dtlog, dtlog_cap);
  R tprev = t0;
  R tnext = rmin(t0 + dt0, t1);
  long steps = 0;
  while (tprev < t1) {  // uniform over the workgroup
    if (steps >= max_steps) return true;
    wg_dopri5_step<R, EPT>(L, own, d, tnext - tprev, with_P, rhs, tb, a, tprev);
    tprev = rmin(tnext, t1);
    const R tn = tnext + dt0;
    tnext = (tn > t1 - Tol<R>::v) ? t1 : tn;
    ++steps;
  }
  return false;
}

// EKF moment right-hand side for this thread's entries (inference_ekf.py:76-123):
//   kP_ij = sum_k F_ik Ps_kj + sum_k F_jk Ps_ki + LQL_ij;   kM_i = f_i (+ 0.5 sum_k g_k Ps_ki)
template <typename R, int EPT>
__device__ __forceinline__ void wg_rhs_ekf(const WgArgs<R>& a, const WgLds<R>& L, const Own<R, EPT>& own, const R* ms,
                                           const R* Ps, R& kM, R (&kP)[EPT], bool mean_only) {
  const int d = a.d, lq = a.lq;
  R* fv = L.vec(2);
  R* gv = L.vec(3);
#ifdef CDKF_WG_CUSTOM
  const bool second = (a.order == 2) && (a.kind == kDriftMlp || (a.kind >= kDriftCustomBase && CDKF_WG_CUSTOM_SECOND));
#else
  const bool second = (a.order == 2) && (a.kind == kDriftMlp);
#endif
  if (a.kind == kDriftLorenz96) {
    const R forcing = (a.par + a.o_theta)[0];
    if (threadIdx.x < d) {
      const int i = threadIdx.x;
      const int ip1 = (i + 1 == d) ? 0 : i + 1, im1 = (i == 0) ? d - 1 : i - 1, im2 = (im1 == 0) ? d - 1 : im1 - 1;
      kM = rfma(ms[ip1] - ms[im2], ms[im1], forcing - ms[i]);
    }
    if (mean_only) return;
    // banded Jacobian rows evaluated on the fly: F_i,i+1 = x_{i-1}, F_i,i-2 = -x_{i-1}, F_i,i-1 = x_{i+1} - x_{i-2}, F_ii = -1
#pragma unroll
    for (int u = 0; u < EPT; ++u)
      if (u < own.n) {
        int i, j;
        own.at(u, i, j);
        const int ip1 = (i + 1 == d) ? 0 : i + 1, im1 = (i == 0) ? d - 1 : i - 1, im2 = (im1 == 0) ? d - 1 : im1 - 1;
        const int jp1 = (j + 1 == d) ? 0 : j + 1, jm1 = (j == 0) ? d - 1 : j - 1, jm2 = (jm1 == 0) ? d - 1 : jm1 - 1;
        const R xi1 = ms[im1], xj1 = ms[jm1];
        R s = -xi1 * Ps[im2 * lq + j];
        s = rfma(ms[ip1] - ms[im2], Ps[im1 * lq + j], s);
        s -= Ps[i * lq + j];
        s = rfma(xi1, Ps[ip1 * lq + j], s);
        R w = -xj1 * Ps[jm2 * lq + i];
        w = rfma(ms[jp1] - ms[jm2], Ps[jm1 * lq + i], w);
        w -= Ps[j * lq + i];
        w = rfma(xj1, Ps[jp1 * lq + i], w);
        kP[u] = (s + w) + own.lql[u];
      }
    return;
  }
  R* F = L.mat(L.plan.i_F);
  R* A = L.mat(L.plan.i_A);
  wg_drift(a, L, ms, fv, mean_only ? (R*)nullptr : F, second ? gv : (R*)nullptr);
  if (threadIdx.x < d) kM = fv[threadIdx.x];
  if (mean_only) return;
  wg_mm<R, false, false, false>(A, F, Ps, d, d, d, lq);
  __syncthreads();
#pragma unroll
  for (int u = 0; u < EPT; ++u)
    if (u < own.n) {
      int i, j;
      own.at(u, i, j);
      kP[u] = (A[i * lq + j] + A[j * lq + i]) + own.lql[u];
    }
  if (second && threadIdx.x < d) {
    R s = 0;
    for (int k = 0; k < d; ++k) s = rfma(gv[k], Ps[k * lq + threadIdx.x], s);
    kM = rfma(R(0.5), s, kM);
  }
}

// UKF moment right-hand side (inference_ukf.py:128-152) in the antisymmetric form of cdkf_reg_kernels.h:
//   O = c chol(Ps);  dm = w_m0 f(m) + w_i sum_i [f(m + o_i) + f(m - o_i)];
//   dP = foo + foo^T + LQL,  foo = w_i DF O^T,  DF[:, i] = f(m + o_i) - f(m - o_i).
// LDS use during the RK stages: O in the F slot, DF in the A slot, foo in the S slot (free between updates).
template <typename R, int EPT>
__device__ __forceinline__ void wg_rhs_ukf(const WgArgs<R>& a, const WgLds<R>& L, const Own<R, EPT>& own, const R* ms,
                                           const R* Ps, R& kM, R (&kP)[EPT], int* bad) {
  const int d = a.d, lq = a.lq;
  R* O = L.mat(L.plan.i_F);
  R* DF = L.mat(L.plan.i_A);
  R* foo = L.mat(2);
  R* f0 = L.vec(2);
  R* sum = L.vec(3);
  R* xs = L.vec(4);   // a sigma point
  R* fx = L.vec(5);   // its drift
  R* inv = L.vec(6);
  CDKF_WG_FOR(e, d * d) {
    const int i = fdiv(e, d), j = e - i * d;
    O[i * lq + j] = Ps[i * lq + j];
  }
  wg_cholesky2(O, inv, (R*)nullptr, (R*)nullptr, d, lq, bad);
  CDKF_WG_FOR(e, d * d) {
    const int i = fdiv(e, d), j = e - i * d;
    O[i * lq + j] = (j <= i) ? a.ukf_c * O[i * lq + j] : R(0);
  }
  __syncthreads();
  if (a.kind == kDriftLinear) {
    // f(m +- o) = W (m +- o) + b:  DF = 2 W O,  sum = 2 d f(m)
    const R* th = a.par + a.o_theta;
    wg_drift(a, L, ms, f0, (R*)nullptr, (R*)nullptr);
    CDKF_WG_FOR(e, d * d) {
      const int r = fdiv(e, d), i = e - r * d;
      R acc = 0;
      for (int k = i; k < d; ++k) acc = rfma(th[r * d + k], O[k * lq + i], acc);
      DF[r * lq + i] = acc + acc;
    }
    CDKF_WG_FOR(r, d) sum[r] = R(2 * d) * f0[r];
    __syncthreads();
  } else if (a.kind == kDriftLorenz96) {
    const R forcing = (a.par + a.o_theta)[0];
    CDKF_WG_FOR(r, d) {
      const int rp1 = (r + 1 == d) ? 0 : r + 1, rm1 = (r == 0) ? d - 1 : r - 1, rm2 = (rm1 == 0) ? d - 1 : rm1 - 1;
      f0[r] = rfma(ms[rp1] - ms[rm2], ms[rm1], forcing - ms[r]);
      sum[r] = 0;
    }
    CDKF_WG_FOR(e, d * d) {
      const int r = fdiv(e, d), i = e - r * d;
      const int rp1 = (r + 1 == d) ? 0 : r + 1, rm1 = (r == 0) ? d - 1 : r - 1, rm2 = (rm1 == 0) ? d - 1 : rm1 - 1;
      const R op1 = O[rp1 * lq + i], om1 = O[rm1 * lq + i], om2 = O[rm2 * lq + i], o0 = O[r * lq + i];
      const R fp = rfma((ms[rp1] + op1) - (ms[rm2] + om2), ms[rm1] + om1, forcing - (ms[r] + o0));
      const R fm = rfma((ms[rp1] - op1) - (ms[rm2] - om2), ms[rm1] - om1, forcing - (ms[r] - o0));
      DF[r * lq + i] = fp - fm;
      foo[r * lq + i] = fp + fm;  // summed over i below
    }
    __syncthreads();
    CDKF_WG_FOR(r, d) {
      R acc = 0;
      for (int i = 0; i < d; ++i) acc += foo[r * lq + i];
      sum[r] = acc;
    }
    __syncthreads();
#ifdef CDKF_WG_CUSTOM
  } else if (a.kind >= kDriftCustomBase) {  // a thread per sigma-point pair
    wg_custom_sigma<R>(a, L, ms, O, f0, DF, foo);
    __syncthreads();
    CDKF_WG_FOR(r, d) {
      R acc = 0;
      for (int i = 0; i < d; ++i) acc += foo[r * lq + i];
      sum[r] = acc;
    }
    __syncthreads();
#endif
  } else {  // generic (Lorenz-63, MLP): one drift evaluation per sigma point
    wg_drift(a, L, ms, f0, (R*)nullptr, (R*)nullptr);
    CDKF_WG_FOR(r, d) sum[r] = 0;
    __syncthreads();
    for (int i = 0; i < d; ++i) {
      for (int sgn = 0; sgn < 2; ++sgn) {
        CDKF_WG_FOR(r, d) xs[r] = sgn ? ms[r] - O[r * lq + i] : ms[r] + O[r * lq + i];
        __syncthreads();
        wg_drift(a, L, xs, fx, (R*)nullptr, (R*)nullptr);
        CDKF_WG_FOR(r, d) {
          sum[r] += fx[r];
          DF[r * lq + i] = sgn ? DF[r * lq + i] - fx[r] : fx[r];
        }
        __syncthreads();
      }
    }
  }
  if (threadIdx.x < d) kM = rfma(a.ukf_wm0, f0[threadIdx.x], a.ukf_wi * sum[threadIdx.x]);
  // foo[r][b] = w_i sum_{i <= b} DF[r][i] O[b][i]
  CDKF_WG_FOR(e, d * d) {
    const int r = fdiv(e, d), b = e - r * d;
    R acc = 0;
    for (int i = 0; i <= b; ++i) acc = rfma(DF[r * lq + i], O[b * lq + i], acc);
    foo[r * lq + b] = a.ukf_wi * acc;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < EPT; ++u)
    if (u < own.n) {
      int i, j;
      own.at(u, i, j);
      kP[u] = (foo[i * lq + j] + foo[j * lq + i]) + own.lql[u];
    }
}

// ---- EKF update on the LDS state (inference_ekf.py:153-199, 285-286) --------------------------------------------
template <typename R>
__device__ void wg_ekf_update(const WgArgs<R>& a, const WgLds<R>& L, const R* __restrict__ yl, double* ll, int* bad) {
#ifdef CDKF_PHASE_PROFILE
  __shared__ long long prof_t[8];
#endif
  CDKF_TICK(0);
  const int d = a.d, m = a.m, lq = a.lq;
  const R* hb = a.par + a.o_hb;
  const R* Rm = a.par + a.o_R;
  R* mm = L.vec(0);
  R* P = L.mat(0);
  R* X = L.mat(1);   // stage matrix is free during the update
  R* S = L.mat(2);
  R* L1 = L.mat(3);
  R* L2 = L.mat(4);
  R* SX = L.mat(3);  // aliases L1: the TFP factor is dead once z and the log-determinant are taken
  R* v = L.vec(4);
  R* z = L.vec(5);
  R* inv1 = L.vec(6);
  R* inv2 = L.vec(7);
  R* tmp = L.vec(11);
  const bool hsel = a.hsel != 0;
  R* HP = hsel ? P : L.mat(L.plan.i_HP);  // with H = I[:m] the first m rows of P ARE H P
  R* Hl = hsel ? (R*)nullptr : L.mat(L.plan.i_Hl);
  if (!hsel) {
    const R* H = a.par + a.o_H;
    CDKF_WG_FOR(e, m * d) {
      const int r = fdiv(e, d);
      Hl[r * lq + (e - r * d)] = H[e];
    }
    __syncthreads();
  }
  const bool ukf = a.ukf != 0;
  R* O = ukf ? L.mat(L.plan.i_F) : (R*)nullptr;   // UKF: O = c chol(P), columns are the sigma offsets o_i
  R* HO = ukf ? (hsel ? O : L.mat(L.plan.i_A)) : (R*)nullptr;
  for (int it = 0; it < (ukf ? 1 : a.num_iter); ++it) {
    if (ukf) {
      // unscented update for the linear emission (inference_ukf.py:162-203), antisymmetric form (cdkf_reg_kernels.h):
      // dY_i = H o_i,  S = 2 w_i sum_i dY_i dY_i^T + R,  C = 2 w_i sum_i o_i dY_i^T,  ybar = H m + b
      CDKF_WG_FOR(e, d * d) {
        const int i = fdiv(e, d), j = e - i * d;
        O[i * lq + j] = P[i * lq + j];
      }
      wg_cholesky2(O, inv1, (R*)nullptr, (R*)nullptr, d, lq, bad);
      CDKF_WG_FOR(e, d * d) {
        const int i = fdiv(e, d), j = e - i * d;
        O[i * lq + j] = (j <= i) ? a.ukf_c * O[i * lq + j] : R(0);
      }
      __syncthreads();
      if (!hsel) {
        wg_mm<R, false, false, false>(HO, Hl, O, m, d, d, lq);
        __syncthreads();
      }
      const R w2 = a.ukf_wi + a.ukf_wi;
      CDKF_WG_FOR(e, m * m) {
        const int r = fdiv(e, m), c = e - r * m;
        R acc = 0;
        for (int k = 0; k < d; ++k) acc = rfma(HO[r * lq + k], HO[c * lq + k], acc);
        S[r * lq + c] = rfma(w2, acc, Rm[e]);
      }
      CDKF_WG_FOR(e, m * d) {  // X0 = C^T = 2 w_i HO O^T, kept in the HP slot semantics below via X directly
        const int r = fdiv(e, d), c = e - r * d;
        R acc = 0;
        for (int k = 0; k <= c; ++k) acc = rfma(HO[r * lq + k], O[c * lq + k], acc);
        X[r * lq + c] = w2 * acc;
      }
      CDKF_WG_FOR(r, m) {
        R sv = 0;
        if (hsel) {
          sv = mm[r];
        } else {
          for (int k = 0; k < d; ++k) sv = rfma(Hl[r * lq + k], mm[k], sv);
          sv += hb[r];
        }
        v[r] = yl[r] - sv;
      }
    } else if (hsel) {
      CDKF_WG_FOR(e, m * m) {
        const int r = fdiv(e, m), c = e - r * m;
        S[r * lq + c] = P[r * lq + c] + Rm[e];
      }
      CDKF_WG_FOR(r, m) v[r] = yl[r] - mm[r];
    } else {
      wg_mm<R, false, false, false>(HP, Hl, P, m, d, d, lq);
      __syncthreads();
      CDKF_WG_FOR(e, m * m) {
        const int r = fdiv(e, m), c = e - r * m;
        R acc = 0;
        for (int k = 0; k < d; ++k) acc = rfma(HP[r * lq + k], Hl[c * lq + k], acc);
        S[r * lq + c] = acc + Rm[e];
      }
      CDKF_WG_FOR(r, m) {
        R s = 0;
        for (int k = 0; k < d; ++k) s = rfma(Hl[r * lq + k], mm[k], s);
        v[r] = yl[r] - (s + hb[r]);
      }
    }
    __syncthreads();
    // L1 <- S (TFP log_prob, no jitter; only on the first iteration), L2 <- symmetrize(S) + 1e-9 I (psd_solve), X <- H P
    CDKF_WG_FOR(e, m * m) {
      const int r = fdiv(e, m), c = e - r * m;
      L1[r * lq + c] = S[r * lq + c];
      R s = R(0.5) * (S[r * lq + c] + S[c * lq + r]);
      if (r == c) s += R(1e-9);
      L2[r * lq + c] = s;
    }
    if (!ukf) CDKF_WG_FOR(e, m * d) {
        const int r = fdiv(e, d), c = e - r * d;
        X[r * lq + c] = HP[r * lq + c];
      }
    CDKF_TICK(1);
    wg_cholesky2(it == 0 ? L1 : (R*)L2, it == 0 ? inv1 : inv2, it == 0 ? L2 : (R*)nullptr, inv2, m, lq, bad);
    CDKF_TICK(2);
    if (it == 0 && threadIdx.x < 64) {
      // z = L1^-1 v and the log-likelihood term on one wavefront (the others go on to the gain solve and meet it at that
      // routine's first barrier): lane r carries v_r, pivots of the step broadcast with v_readlane, the column entries and
      // reciprocal pivots of four steps in flight per LDS round trip; log-determinant from the reciprocal pivots in parallel
      const int lane = threadIdx.x;
      R vr = (lane < m) ? v[lane] : R(0);
      double qd = 0.0;
      const R* lrow = L1 + ((lane < m) ? lane : m - 1) * lq;
      for (int j0 = 0; j0 < m; j0 += 4) {
        R lj[4], ij[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int jj = (j0 + q < m) ? j0 + q : m - 1;
          lj[q] = lrow[jj];
          ij[q] = inv1[jj];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int j = j0 + q;
          if (j < m) {
            const R zj = wave_bcast<R>(vr, j) * ij[q];
            if (lane > j && lane < m) vr = rfma(-lj[q], zj, vr);
            qd += (double)zj * (double)zj;
          }
        }
      }
      double ld = (lane < m) ? log((double)inv1[lane]) : 0.0;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) ld += __shfl_down(ld, off, 64);
      if (lane == 0) *ll += -0.5 * qd + ld - 0.5 * m * 1.8378770664093454835606594728112;
    }
    CDKF_TICK(3);
    wg_chol_solve(L2, inv2, X, m, d, lq);            // X = Sb^-1 (H P);   K = X^T
    CDKF_TICK(4);
    wg_mm<R, false, false, false>(SX, S, X, m, m, d, lq);  // S X
    CDKF_WG_FOR(i, d) {                              // m+ = m + K v (staged: v is still being read)
      R s = mm[i];
      for (int r = 0; r < m; ++r) s = rfma(X[r * lq + i], v[r], s);
      tmp[i] = s;
    }
    __syncthreads();
    wg_mm<R, true, false, true>(P, X, SX, d, m, d, lq);  // P <- P - X^T (S X)
    CDKF_WG_FOR(i, d) mm[i] = tmp[i];
    __syncthreads();
    CDKF_TICK(5);
  }
  // symmetrize (dynamax/utils/utils.py:209-211); the reference's UKF does not
  if (!ukf) CDKF_WG_FOR(e, d * d) {
    const int i = fdiv(e, d), j = e - i * d;
    if (i < j) {
      const R s = R(0.5) * (P[i * lq + j] + P[j * lq + i]);
      P[i * lq + j] = s;
      P[j * lq + i] = s;
    }
  }
  __syncthreads();
#ifdef CDKF_PHASE_PROFILE
  CDKF_TICK(6);
  if (threadIdx.x == 0 && blockIdx.x == 0 && a.T > 20)
    printf("update phases (x10ns): setup %lld chol %lld ll %lld solve %lld SX+P %lld sym %lld\n", prof_t[1] - prof_t[0],
           prof_t[2] - prof_t[1], prof_t[3] - prof_t[2], prof_t[4] - prof_t[3], prof_t[5] - prof_t[4], prof_t[6] - prof_t[5]);
#endif
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
    CDKF_WG_FOR(e, d * d) {
      const int i = fdiv(e, d);
      p[e * a.P_si] = P[i * lq + (e - i * d)];
    }
  }
}

// ---- EKF filter sweep ----------------------------------------------------------------------------------------------
// UKF: unscented instead of extended filter.  KIND: kDriftAny (drift chosen at run time) or one drift kind -- a specialised
// instantiation does not carry the register pressure of the code paths it cannot take (the union spilled to scratch).
constexpr int kDriftAny = -1;
template <typename R, int EPT, bool UKF, int KIND>
__global__ __launch_bounds__(512) void ekf_filter_wg_kernel(const WgArgs<R> a_in) {
  WgArgs<R> a = a_in;
  if constexpr (KIND != kDriftAny) a.kind = KIND;
  a.ukf = UKF ? 1 : 0;
#ifdef CDKF_WG_STATIC_LDS  // run-time compiled for one shape: the carve-up's size is a constant (no dynamic-LDS cap to raise on a module function)
  __shared__ __attribute__((aligned(16))) unsigned char smem_raw[CDKF_WG_STATIC_LDS];
#else
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
#endif
  const WgPlan plan = wg_plan(a.kind, a.d, a.h1, a.h2, a.hsel, false, a.ukf != 0);
  WgLds<R> L(reinterpret_cast<R*>(smem_raw), a.q, a.lq, plan);
  __shared__ int bad;
  __shared__ double ll;
  const long n = blockIdx.x;
  const int d = a.d, m = a.m, lq = a.lq;
  if (threadIdx.x == 0) {
    bad = 0;
    ll = 0.0;
  }
  CDKF_WG_FOR(e, (int)((long)plan.nmat * a.q * lq + (long)plan.nvec * lq)) L.base[e] = 0;
  __syncthreads();
  CDKF_WG_FOR(i, d) L.vec(0)[i] = (a.par + a.o_m0)[i];
  CDKF_WG_FOR(e, d * d) {
    const int i = fdiv(e, d), j = e - i * d;
    L.mat(0)[i * lq + j] = R(0.5) * ((a.par + a.o_P0)[i * d + j] + (a.par + a.o_P0)[j * d + i]);
  }
  wg_mlp_prepare(a, L);
  Own<R, EPT> own;
  own.init(d, lq, a.par + a.o_LQL);
  __syncthreads();
  const R* tp = a.t + n * a.t_sn;
  const R* yp = a.y + n * a.y_sn;
  R* yl = L.vec(8);
  int st = 0;
  const bool zeroth = (a.order == 0) && !a.ukf;
  auto rhs = [&](const R* ms, const R* Ps, R& kM, R (&kP)[EPT]) {
    if constexpr (UKF)
      wg_rhs_ukf<R, EPT>(a, L, own, ms, Ps, kM, kP, &bad);
    else
      wg_rhs_ekf<R, EPT>(a, L, own, ms, Ps, kM, kP, zeroth);
  };
  a.ctx_rev = 0;
  a.ctx_tend = R(0);
  a.ctx_t = R(0);
  for (long k = 0; k < a.T; ++k) {
    a.ctx_uoff = n * a.u_sn + k * a.u_sk;  // u = inputs[t0_idx]: this step's interval (inference_ekf.py:277)
    CDKF_WG_FOR(r, m) yl[r] = yp[k * a.y_sk + r * a.y_si];
    const R t0 = tp[k * a.t_sk];
    const R t1 = (k + 1 < a.T) ? tp[(k + 1) * a.t_sk] : t0 + a.dt_final;
    __syncthreads();
    if (!a.forecast) wg_ekf_update(a, L, yl, &ll, &bad);
    wg_store(a, L, a.fm, a.fP, n, k);
    __syncthreads();
    // (re)derive this thread's entry indices here: values that live across the measurement update would be spilled to
    // scratch by its register pressure and re-loaded inside every Runge-Kutta stage (measured: 3.4x on the d = 40 sweep)
    own.init(d, lq, a.par + a.o_LQL);
    if (wg_integrate<R, EPT>(L, own, d, t0, t1, a.dt0, a.max_steps, !zeroth, rhs, a.rk, a,
                             (a.dtlog && k + 1 < a.T) ? a.dtlog + (n * (a.T - 1) + k) * (1 + a.dtlog_cap) : (R*)nullptr, a.dtlog_cap))
      st |= kStatusMaxSteps;
    if (zeroth) {
      const R sq = rsqrt_(t1 - t0);
      const R* Qz = a.par + a.o_LQLz;
      CDKF_WG_FOR(e, d * d) {
        const int i = fdiv(e, d), j = e - i * d;
        L.mat(0)[i * lq + j] = rfma(sq, Qz[e], L.mat(0)[i * lq + j]);
      }
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

// ---- EKF smoother backward sweep (inference_ekf.py:363-448, 503-531) ---------------------------------------------
// Per interval the filtered (m_f, P_f) at t_k are constants: G = F(m_f) + psd_solve(P_f, LQL)^T and f(m_f) are formed
// once; the reverse-time right-hand side is  dm = -[f(m_f) + G (m_s - m_f)],  dP = -[G P_s + (G P_s)^T - LQL].
template <typename R, int EPT>
__global__ __launch_bounds__(512) void ekf_smoother_wg_kernel(const WgArgs<R> a) {
#ifdef CDKF_WG_STATIC_LDS  // run-time compiled for one shape: the carve-up's size is a constant (no dynamic-LDS cap to raise on a module function)
  __shared__ __attribute__((aligned(16))) unsigned char smem_raw[CDKF_WG_STATIC_LDS];
#else
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
#endif
  const WgPlan plan = wg_plan(a.kind, a.d, a.h1, a.h2, a.hsel, true);
  WgLds<R> L(reinterpret_cast<R*>(smem_raw), a.q, a.lq, plan);
  __shared__ int bad;
  const long n = blockIdx.x;
  const int d = a.d, lq = a.lq;
  if (threadIdx.x == 0) bad = 0;
  CDKF_WG_FOR(e, (int)((long)plan.nmat * a.q * lq + (long)plan.nvec * lq)) L.base[e] = 0;
  __syncthreads();
  wg_mlp_prepare(a, L);
  Own<R, EPT> own;
  own.init(d, lq, a.par + a.o_LQL);
  const R* tp = a.t + n * a.t_sn;
  const R* fm = a.fm + n * a.m_sn;
  const R* fP = a.fP + n * a.P_sn;
  int st = 0;
  {
    const long k = a.T - 1;
    CDKF_WG_FOR(i, d) L.vec(0)[i] = fm[k * a.m_sk + i * a.m_si];
    CDKF_WG_FOR(e, d * d) {
      const int i = fdiv(e, d);
      L.mat(0)[i * lq + (e - i * d)] = fP[k * a.P_sk + e * a.P_si];
    }
    __syncthreads();
    wg_store(a, L, a.sm, a.sP, n, k);
    __syncthreads();
  }
  R* G = L.mat(plan.i_F);
  R* A = L.mat(plan.i_A);
  R* Lc = L.mat(3);
  R* X = L.mat(4);
  R* inv = L.vec(6);
  R* mf = L.vec(9);
  R* fmf = L.vec(10);
  const R* LQL = a.par + a.o_LQL;
#ifdef CDKF_WG_CUSTOM
  const bool time_dep = CDKF_WG_CUSTOM_TIME != 0;
#else
  const bool time_dep = false;
#endif
  auto rhs = [&](const R* ms, const R* Ps, R& kM, R (&kP)[EPT]) {
    if (time_dep) {  // f(m_f, u, t) and jacfwd(f)(m_f, u, t) at t = t1 - s in EVERY stage (inference_ekf.py:433-438); aux = X^T stays
      wg_drift(a, L, mf, fmf, G, (R*)nullptr);
      CDKF_WG_FOR(e, d * d) {
        const int i = fdiv(e, d), j = e - i * d;
        G[i * lq + j] += X[j * lq + i];
      }
      __syncthreads();
    }
    wg_mm<R, false, false, false>(A, G, Ps, d, d, d, lq);
    if (threadIdx.x < d) {
      R s = 0;
      for (int k = 0; k < d; ++k) s = rfma(G[threadIdx.x * lq + k], ms[k] - mf[k], s);
      kM = -(fmf[threadIdx.x] + s);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < EPT; ++u)
      if (u < own.n) {
        int i, j;
        own.at(u, i, j);
        kP[u] = -((A[i * lq + j] + A[j * lq + i]) - own.lql[u]);
      }
  };
  R t1 = tp[(a.T - 1) * a.t_sk];
  for (long k = a.T - 2; k >= 0; --k) {
    const R t0 = tp[k * a.t_sk];
    a.ctx_uoff = n * a.u_sn + k * a.u_sk;  // u = inputs[t0_idx] of the interval (inference_ekf.py:516); t = t1 - s below
    a.ctx_rev = 1;
    a.ctx_tend = t1;
    a.ctx_t = t1;
    CDKF_WG_FOR(i, d) mf[i] = fm[k * a.m_sk + i * a.m_si];
    CDKF_WG_FOR(e, d * d) {
      const int i = fdiv(e, d);
      L.mat(2)[i * lq + (e - i * d)] = fP[k * a.P_sk + e * a.P_si];
    }
    __syncthreads();
    CDKF_WG_FOR(e, d * d) {
      const int r = fdiv(e, d), c = e - r * d;
      R s = R(0.5) * (L.mat(2)[r * lq + c] + L.mat(2)[c * lq + r]);
      if (r == c) s += R(1e-9);
      Lc[r * lq + c] = s;
      X[r * lq + c] = LQL[e];
    }
    wg_cholesky2(Lc, inv, (R*)nullptr, (R*)nullptr, d, lq, &bad);
    wg_chol_solve(Lc, inv, X, d, d, lq);  // X = P_f^{-1} LQL
    wg_drift(a, L, mf, fmf, G, (R*)nullptr);
    CDKF_WG_FOR(e, d * d) {
      const int i = fdiv(e, d), j = e - i * d;
      G[i * lq + j] += X[j * lq + i];
    }
    __syncthreads();
    own.init(d, lq, a.par + a.o_LQL);  // see the filter kernel
    if (wg_integrate<R, EPT>(L, own, d, R(0), t1 - t0, a.dt0, a.max_steps, true, rhs, a.rk, a)) st |= kStatusMaxSteps;
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
