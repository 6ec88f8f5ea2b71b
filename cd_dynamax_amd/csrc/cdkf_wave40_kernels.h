// cdkf_wave40_kernels.h -- wavefront-per-trajectory EKF filter sweep AND smoother backward sweep for the Lorenz-96 model at state
// dimensions 12 .. 40 (BASELINE config 4: d = m = 40), emission = a selection of state components (H = I, or any m <= d rows of it
// in any order; any symmetric R).
//
// The workgroup-per-trajectory kernel (cdkf_wg2_kernels.h) spends a d = 40 observation step in ~35 barrier-separated phases with a
// few dozen flops per thread in each: 512 threads wait on each other most of the time (SQ_WAIT_ANY 69 % of the wave cycles,
// profiles/r02_c_config4_counters.json).  Here ONE wavefront owns a trajectory for the whole scan and nothing inside the time loop
// waits on another wavefront; four wavefronts (four trajectories) share a workgroup only to fill the CU's four SIMDs.
//
//  * State: the packed upper triangle of P (820 entries at d = 40) dealt round-robin to the 64 lanes -- 13 entries per lane, with
//    their six Dormand-Prince slopes, in registers; lanes < d own the mean.  The entries' indices live in an LDS table (two words
//    per entry) and are fetched where they are used.
//  * Predict (inference_ekf.py:76-123): the stage covariance is written to an LDS image with a two-element halo (the Lorenz-96
//    neighbourhood i-2, i-1, i+1 wraps around), so an entry's right-hand side
//        dP_ij = a_i (P_{i+1,j} - P_{i-2,j}) + b_i P_{i-1,j} + a_j (P_{i,j+1} - P_{i,j-2}) + b_j P_{i,j-1} - 2 P_ij + (L Qc L^T)_ij,
//        a_i = x_{i-1},  b_i = x_{i+1} - x_{i-2}                      (the four non-zeros of the Jacobian's row i)
//    is six LDS reads at constant offsets from ONE per-entry address, two 16-byte coefficient reads and eight flops; two
//    wavefront-scope synchronisations per stage.
//  * Dense linear algebra (W40Lin below): Cholesky factorisation and triangular solves in 16-wide blocks -- every matrix product
//    among them on the matrix cores, only the 16 x 16 triangles on the diagonal as scalar recurrences.
//  * Update (inference_ekf.py:153-199, 285-286; H = I so H P = P, S = P + R):
//      - both Cholesky factors (TFP's of S for the log-likelihood, psd_solve's of S + 1e-9 I for the gain) in lockstep; the
//        innovation rides along as row d of the un-jittered system, which makes its forward substitution (the log-likelihood's
//        quadratic form) part of the factorisation;
//      - X = (S + 1e-9 I)^-1 P with lane c = right-hand side c;  m+ = m + X^T (y - m);
//      - P+ = P - X^T S X as P - Y^T Y + 1e-9 X^T X with Y = L_b^-1 P the forward solve's result (S = L_b L_b^T - 1e-9 I): two
//        symmetric rank-d products on the matrix cores (v_mfma_*_16x16x4, six upper 16 x 16 tiles, operands straight from the LDS
//        image of Y / X, whose A- and B-operand layouts coincide), the -1e-9 folded into the A operand.  Needs R symmetric.
//  * Backward sweep (ekf_smoother_wave_l96_kernel, further down): G = F(m_f) + psd_solve(P_f, L Qc L^T)^T per interval from the same
//    W40Lin, G P_s per Runge-Kutta stage as 3 x 3 tiles on the matrix cores.
//  * Outputs stream from the LDS images (full d x d rows, coalesced).
//
// Scope: drift Lorenz-96, H = rows of the identity (m <= d), num_iter = 1, state_order first / second (the same for this drift: grad(div f) = 0), fixed-
// step Dormand-Prince, no forecast.  Everything else stays on cdkf_wg2_kernels.h; CDKF_NO_WAVE40=1 forces that (A/B, tests).
#pragma once
#include "cdkf_wave8_kernels.h"

namespace cdkf {

template <int D>
struct W40 {
  static_assert(D > 8 && D <= 48, "three 16-wide tiles");
  static constexpr int NP = D * (D + 1) / 2;        // packed upper triangle
  static constexpr int EPL = (NP + 63) / 64;        // entries per lane
  static constexpr int LDP = (D + 4) | 1;           // stage image: rows / columns -2 .. D (halo), leading dimension -- odd: the transposed
                                                    // writes of the symmetric image then spread over the banks (D = 28: 32 put all 64 lanes on one)
  static constexpr int LDY = D + 2;                 // update image: 48 rows (three tiles) x D; even (16-byte rows for ds_read_b128)
  static constexpr int BUF = (((D + 3) * LDP > 48 * LDY) ? (D + 3) * LDP : 48 * LDY) + 1 & ~1;
  // packed lower triangle of the (D + 1)-row augmented system, every row padded to an even length: row i starts at rs(i), an
  // even offset, so eight consecutive entries from a column multiple of eight are four aligned 16-byte reads
  __host__ __device__ static constexpr int rs(int i) { return 2 * ((i + 1) / 2) * (i / 2 + 1); }
  static constexpr int LPK = rs(D + 1);
  static constexpr int VEC = 64;
  // per-wavefront carve-up (reals)
  static constexpr int o_buf = 0, o_L1 = BUF, o_L2 = o_L1 + LPK, o_xs = o_L2 + LPK, o_ca = o_xs + VEC, o_cb = o_ca + VEC,
                       o_i2 = o_cb + VEC, o_v = o_i2 + VEC, o_end = o_v + VEC;
  static constexpr int kWaves = 4;
  // per workgroup: (L Qc L^T) and R entries in ownership order [s][lane], then the index table of the owned entries (two 32-bit
  // words per entry: (i, j), and the entry's three offsets in the update's images), sized for 4-byte reals
  static constexpr int SHQ = 2 * 64 * EPL, SHT = SHQ + 2 * 64 * EPL, SH = SHT + 64;
  // table word A: i | j << 8 | (state component i observed) << 16 | (j observed) << 17;
  // word B: (i LDY + j) | (j LDY + i) << 11 | (rs(j) + i) << 22;  behind the tables: obs[i] = the emission row that observes state
  // component i, or -1 (64 words)
  static_assert(48 * (D + 2) < 2048 && 2 * ((D + 2) / 2) * ((D + 1) / 2 + 1) < 1024, "offsets fit their fields");
  __host__ __device__ static constexpr bool owned(int s, int lane) { return lane + 64 * s < NP; }
};
template <int D>
__host__ __device__ constexpr long wave40_lds_reals() { return (long)W40<D>::SH + (long)W40<D>::kWaves * W40<D>::o_end; }

template <typename R>
struct W40Tile;
template <>
struct W40Tile<double> {
  using V4 = wg_f64x4;
  static CDKF_DEV int row(int g, int r) { return g + 4 * r; }  // f64 16x16x4: row = (lane >> 4) + 4 reg
};
template <>
struct W40Tile<float> {
  using V4 = wg_f32x4;
  static CDKF_DEV int row(int g, int r) { return 4 * g + r; }  // f32 16x16x4: row = 4 (lane >> 4) + reg
};

CDKF_DEV double w40_rsqrt(double x) {
  const double y0 = __builtin_amdgcn_rsq(x);
  const double e = rfma(-(x * y0), y0, 1.0);
  return rfma(y0 * e, rfma(e, 0.375, 0.5), y0);
}
CDKF_DEV float w40_rsqrt(float x) { return rrsqrt(x); }
CDKF_DEV double w40_readlane(double v, int l) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)b, l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
CDKF_DEV float w40_readlane(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }

// ---- dense linear algebra of one wavefront on LDS-resident operands ------------------------------------------------------------
// A symmetric positive-definite system of order D (+ one augmented row) in packed lower storage (row i at W40<D>::rs(i)), its
// right-hand sides as the ROWS of an image with leading dimension LDY (lane c owns row c = right-hand side c).  Everything that is
// a matrix product -- the contributions of the finished columns to a panel of the factorisation, of the solved blocks to the next
// block of a triangular solve -- runs on the matrix cores in 16 x 16 tiles (v_mfma_*_16x16x4: A[m = lane & 15][k = lane >> 4],
// B[k = lane >> 4][n = lane & 15]); only the 16-wide triangles on the diagonal are scalar recurrences (lane = row, the panel in
// registers, multipliers through v_readlane / LDS broadcasts).  Callers synchronise the wavefront before the first call.
#ifdef CDKF_W40_PROFILE  // local diagnostic build: cycles per phase (s_memtime), printed by trajectory 0
__device__ long long w40_prof[16];
#define W40_TICK(i)                                 \
  {                                                 \
    const long long w40_now = clock64();            \
    if (threadIdx.x == 0 && blockIdx.x == 0) w40_prof[i] += w40_now - w40_last; \
    w40_last = clock64();                            \
  }
#define W40_TICK_DECL long long w40_last = clock64();
#define W40_TICK_ARG , long long& w40_last
#define W40_TICK_PASS , w40_last
#else
#define W40_TICK(i)
#define W40_TICK_DECL
#define W40_TICK_ARG
#define W40_TICK_PASS
#endif

template <typename R, int D>
struct W40Lin {
  using W = W40<D>;
  using Tile = W40Tile<R>;
  using V4 = typename Tile::V4;
  static constexpr int LDY = W::LDY;
  static constexpr int NB = (D + 15) / 16;  // blocks of sixteen rows / columns
  static constexpr int NBR = (D + 16) / 16; // ... of the D + 1 rows of the augmented system (D = 32: row 32 is a third block of its own --
                                            // left out, the factor was right and the log-likelihood's quadratic form wrong: found by
                                            // scripts/gpu_fuzz_r03.py, no test had run D = 32)
  static_assert(D % 4 == 0, "panels of four, eight, twelve or sixteen columns");

  // L[i][c0 + c] -= sum_{k < c0} L[i][k] L[c0 + c][k]  for the rows i >= c0 (the augmented row D included), c < 16: the left-looking
  // update of the panel of columns from c0 (a multiple of sixteen), for NS systems at once (independent accumulator chains)
  template <int NS, int P>
  static CDKF_DEV void chol_gemm(R* const (&L)[NS], const int lane) {
    constexpr int c0 = 16 * P;
    const int lm = lane & 15, lg = lane >> 4;
    const int brow = c0 + lm;
    const bool bin = brow <= D;
    const int boff = W::rs(bin ? brow : D);
    V4 acc[NS][NBR];
#pragma unroll
    for (int it = 0; it < NBR; ++it)
      if (it >= P) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * it + Tile::row(lg, r);
          const bool ok = row <= D && brow <= row;
          const int o = W::rs(ok ? row : D) + brow;
#pragma unroll
          for (int s = 0; s < NS; ++s) acc[s][it][r] = ok ? L[s][o] : R(0);
        }
      }
#pragma unroll
    for (int k0 = 0; k0 < c0; k0 += 4) {
      const int kk = k0 + lg;
      R bv[NS], av[NS][NBR];
#pragma unroll
      for (int s = 0; s < NS; ++s) bv[s] = bin ? L[s][boff + kk] : R(0);
#pragma unroll
      for (int it = 0; it < NBR; ++it)
        if (it >= P) {
          const int arow = 16 * it + lm;
          const int ao = W::rs(arow <= D ? arow : D) + kk;
#pragma unroll
          for (int s = 0; s < NS; ++s) av[s][it] = arow <= D ? -L[s][ao] : R(0);
        }
#pragma unroll
      for (int it = 0; it < NBR; ++it)
        if (it >= P) {
#pragma unroll
          for (int s = 0; s < NS; ++s) acc[s][it] = wg_mfma(av[s][it], bv[s], acc[s][it]);
        }
    }
#pragma unroll
    for (int it = 0; it < NBR; ++it)
      if (it >= P) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * it + Tile::row(lg, r);
          if (row <= D && brow <= row) {
#pragma unroll
            for (int s = 0; s < NS; ++s) L[s][W::rs(row) + brow] = acc[s][it][r];
          }
        }
      }
  }

  // the W_ columns from c0 of the rows from c0, lane = row, the panel's entries in registers.  A finished column goes to a 64-entry
  // LDS scratch indexed by row (col[s]); the multipliers L[c0 + q][c] of the columns still to update come back from there as
  // broadcast reads, two per instruction (LDS executes a wavefront's accesses in order: no synchronisation in between).  The
  // pivot and the next column's multiplier travel through v_readlane (the dependency chain then skips the LDS round trip).
  template <int NS, int W_>
  static CDKF_DEV void chol_panel(R* const (&L)[NS], R* const (&col)[NS], R* inv, const int c0, const int rowi, const int ri,
                                  const int lane, R& quad, R& pinv, bool& bad) {
    R u[NS][W_];
#pragma unroll
    for (int r = 0; r < W_; ++r)
#pragma unroll
      for (int s = 0; s < NS; ++s) u[s][r] = L[s][ri + c0 + r];
#pragma unroll
    for (int r = 0; r < W_; ++r) {
      const int c = c0 + r;
      R rr[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const R p = w40_readlane(u[s][r], c);
        bad = bad || !(p > R(0));
        rr[s] = w40_rsqrt(p);
        u[s][r] *= rr[s];  // L[i][c] for the rows below the pivot
        if (r + 1 < W_) col[s][lane] = u[s][r];
      }
      if (lane == D) quad = rfma(u[0][r], u[0][r], quad);
      if (lane == 0) inv[c] = rr[NS - 1];
      pinv *= rr[0];
      R mult[NS][W_];
      // (16-byte reads at immediate offsets from ONE address: c0 is a multiple of sixteen and the scratch is 16-byte aligned)
      typedef R Pair __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const Pair* cp = reinterpret_cast<const Pair*>(__builtin_assume_aligned(col[s] + c0, 16));
#pragma unroll
        for (int h = (r + 1) / 2; h < W_ / 2; ++h) {
          const Pair pr = cp[h];
          mult[s][2 * h] = pr[0];
          mult[s][2 * h + 1] = pr[1];
        }
      }
      // the NEXT column's multiplier through v_readlane: the next pivot then does not wait for the LDS round trip of the scratch
      if (r + 1 < W_) {
#pragma unroll
        for (int s = 0; s < NS; ++s) u[s][r + 1] = rfma(-u[s][r], w40_readlane(u[s][r], c0 + r + 1), u[s][r + 1]);
      }
#pragma unroll
      for (int q = r + 2; q < W_; ++q)
#pragma unroll
        for (int s = 0; s < NS; ++s) u[s][q] = rfma(-u[s][r], mult[s][q], u[s][q]);
    }
#pragma unroll
    for (int r = 0; r < W_; ++r)
      if (rowi > c0 + r) {
#pragma unroll
        for (int s = 0; s < NS; ++s) L[s][ri + c0 + r] = u[s][r];
      }
  }

  // Cholesky factorisation of NS systems in lockstep, lane = row (rowi; lanes above D shadow row D, ri = rs(rowi)).  System 0's
  // augmented row is forward-substituted along the way: quad accumulates its squares on lane D; logdet the logs of system 0's
  // reciprocal pivots; inv[] (LDS) the reciprocal pivots of the LAST system (the one the solves use); col[s]: 64 reals of LDS
  // scratch per system.  bad: a pivot was not positive.
  template <int NS>
  static CDKF_DEV void cholesky(R* const (&L)[NS], R* const (&col)[NS], R* inv, const int rowi, const int ri, const int lane,
                                R& quad, double& logdet, bool& bad W40_TICK_ARG) {
    static_assert(NB <= 3, "panels beyond the third: add a case");
    for (int P = 0; P < NB; ++P) {
      const int c0 = 16 * P;
      if (P == 1) chol_gemm<NS, 1>(L, lane);
      if (P == 2) chol_gemm<NS, 2>(L, lane);
      if (P) wave_sync();
      W40_TICK(1)
      R pinv = R(1);
      if (D - c0 >= 16)
        chol_panel<NS, 16>(L, col, inv, c0, rowi, ri, lane, quad, pinv, bad);
      else
        chol_panel<NS, (D % 16 ? D % 16 : 16)>(L, col, inv, c0, rowi, ri, lane, quad, pinv, bad);
      W40_TICK(2)
      logdet += log((double)pinv);
      wave_sync();
      W40_TICK(4)
    }
  }

  // img[c][16 b + r] -= sum_k img[c][k] Lf(16 b + r, k) over the unknowns k already solved: k < 16 b going forward (Lf(i, k) = L[i][k]),
  // k >= 16 (b + 1) going backward (Lf(i, k) = L[k][i], the transposed factor); all D right-hand sides c, r < 16
  // (CTM: mask of the sixteen-row groups of right-hand sides this call updates -- two wavefronts can share one product)
  template <bool FWD, int b, int CTM = (1 << NB) - 1>
  static CDKF_DEV void solve_gemm(R* img, const R* L, const int lane) {
    const int lm = lane & 15, lg = lane >> 4;
    const int ucol = 16 * b + lm;  // the unknown this lane's B operand / accumulator column stands for
    const bool uin = ucol < D;
    V4 acc[NB];
#pragma unroll
    for (int ct = 0; ct < NB; ++ct)
      if ((CTM >> ct) & 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * ct + Tile::row(lg, r);
          acc[ct][r] = (row < D && uin) ? img[row * LDY + ucol] : R(0);
        }
      }
    constexpr int kbeg = FWD ? 0 : 16 * (b + 1), kend = FWD ? 16 * b : D;
    const int boff = W::rs(uin ? ucol : 0);
#pragma unroll
    for (int k0 = kbeg; k0 < kend; k0 += 4) {
      const int kk = k0 + lg;
      const R bv = uin ? (FWD ? L[boff + kk] : L[W::rs(kk) + ucol]) : R(0);
      R av[NB];
#pragma unroll
      for (int ct = 0; ct < NB; ++ct)
        if ((CTM >> ct) & 1) av[ct] = (16 * ct + lm < D) ? -img[(16 * ct + lm) * LDY + kk] : R(0);
#pragma unroll
      for (int ct = 0; ct < NB; ++ct)
        if ((CTM >> ct) & 1) acc[ct] = wg_mfma(av[ct], bv, acc[ct]);
    }
#pragma unroll
    for (int ct = 0; ct < NB; ++ct)
      if ((CTM >> ct) & 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * ct + Tile::row(lg, r);
          if (row < D && uin) img[row * LDY + ucol] = acc[ct][r];
        }
      }
  }

  // the W_ unknowns from c0 of this lane's right-hand side: the triangle on the diagonal (factor entries as LDS broadcasts)
  template <bool FWD, int W_>
  static CDKF_DEV void sub_block(R* mine, const R* L, const R* inv, const R* dotw, const int c0, const bool isrow, R& dot) {
    R u[W_];
#pragma unroll
    for (int r = 0; r < W_; ++r) u[r] = mine[c0 + r];
    if constexpr (FWD) {
#pragma unroll
      for (int r = 0; r < W_; ++r) {
        const R* Lr = L + W::rs(c0 + r) + c0;
#pragma unroll
        for (int q = 0; q < r; ++q) u[r] = rfma(-Lr[q], u[q], u[r]);
        u[r] *= inv[c0 + r];
      }
    } else {
      const R* Lq[W_];
#pragma unroll
      for (int q = 0; q < W_; ++q) Lq[q] = L + W::rs(c0 + q) + c0;
#pragma unroll
      for (int r = W_ - 1; r >= 0; --r) {
#pragma unroll
        for (int q = r + 1; q < W_; ++q) u[r] = rfma(-Lq[q][r], u[q], u[r]);
        u[r] *= inv[c0 + r];
      }
      if (dotw) {
#pragma unroll
        for (int r = 0; r < W_; ++r) dot = rfma(u[r], dotw[c0 + r], dot);
      }
    }
    if (isrow) {
#pragma unroll
      for (int r = 0; r < W_; ++r) mine[c0 + r] = u[r];
    }
  }

  // (L L^T) x = b for the D right-hand sides held as the rows of img (lane c < D: row c, in place).  After the forward pass the image
  // holds Y = L^-1 B (by rows: Y^T), handed to `between` (the filter's rank-d product reads it there); returns this lane's
  // x . dotw (dotw: a D-vector in LDS, or nullptr).
  template <typename Between>
  static CDKF_DEV R solve(R* img, const R* L, const R* inv, const R* dotw, const int lane, Between&& between W40_TICK_ARG) {
    constexpr int WL = D % 16 ? D % 16 : 16;  // width of the last block
    const bool isrow = lane < D;
    R* mine = img + (isrow ? lane : 0) * LDY;
    R dot = R(0);
    for (int b = 0; b < NB; ++b) {  // forward
      if (b == 1) solve_gemm<true, 1>(img, L, lane);
      if (b == 2) solve_gemm<true, 2>(img, L, lane);
      if (b) wave_sync();
      W40_TICK(5)
      if (b < NB - 1)
        sub_block<true, 16>(mine, L, inv, dotw, 16 * b, isrow, dot);
      else
        sub_block<true, WL>(mine, L, inv, dotw, 16 * b, isrow, dot);
      wave_sync();
      W40_TICK(6)
    }
    between();
    W40_TICK(7)
    for (int b = NB - 1; b >= 0; --b) {  // backward
      if (b == 0 && NB > 1) solve_gemm<false, 0>(img, L, lane);
      if (b == 1 && NB > 2) solve_gemm<false, 1>(img, L, lane);
      if (b < NB - 1) wave_sync();
      W40_TICK(8)
      if (b < NB - 1)
        sub_block<false, 16>(mine, L, inv, dotw, 16 * b, isrow, dot);
      else
        sub_block<false, WL>(mine, L, inv, dotw, 16 * b, isrow, dot);
      wave_sync();
      W40_TICK(9)
    }
    return dot;
  }
};

// One Dormand-Prince step of the owned covariance entries and the lane's mean component (slopes scaled by dt as they are formed).
template <typename R, int EPL, typename Rhs>
CDKF_DEV void w40_dopri5(Rhs&& rhs, R (&Pe)[EPL], R& mj, const R dt) {
  using C = Dp5<R>;
  R k1[EPL], k2[EPL], k3[EPL], k4[EPL], k5[EPL], k6[EPL], ys[EPL];
  R m1, m2, m3, m4, m5, m6;
  rhs(Pe, mj, k1, m1);
#pragma unroll
  for (int s = 0; s < EPL; ++s) {
    k1[s] *= dt;
    ys[s] = rfma(C::a21, k1[s], Pe[s]);
  }
  m1 *= dt;
  rhs(ys, rfma(C::a21, m1, mj), k2, m2);
#pragma unroll
  for (int s = 0; s < EPL; ++s) {
    k2[s] *= dt;
    ys[s] = rfma(C::a32, k2[s], rfma(C::a31, k1[s], Pe[s]));
  }
  m2 *= dt;
  rhs(ys, rfma(C::a32, m2, rfma(C::a31, m1, mj)), k3, m3);
#pragma unroll
  for (int s = 0; s < EPL; ++s) {
    k3[s] *= dt;
    ys[s] = rfma(C::a43, k3[s], rfma(C::a42, k2[s], rfma(C::a41, k1[s], Pe[s])));
  }
  m3 *= dt;
  rhs(ys, rfma(C::a43, m3, rfma(C::a42, m2, rfma(C::a41, m1, mj))), k4, m4);
#pragma unroll
  for (int s = 0; s < EPL; ++s) {
    k4[s] *= dt;
    ys[s] = rfma(C::a54, k4[s], rfma(C::a53, k3[s], rfma(C::a52, k2[s], rfma(C::a51, k1[s], Pe[s]))));
  }
  m4 *= dt;
  rhs(ys, rfma(C::a54, m4, rfma(C::a53, m3, rfma(C::a52, m2, rfma(C::a51, m1, mj)))), k5, m5);
#pragma unroll
  for (int s = 0; s < EPL; ++s) {
    k5[s] *= dt;
    ys[s] = rfma(C::a65, k5[s], rfma(C::a64, k4[s], rfma(C::a63, k3[s], rfma(C::a62, k2[s], rfma(C::a61, k1[s], Pe[s])))));
  }
  m5 *= dt;
  rhs(ys, rfma(C::a65, m5, rfma(C::a64, m4, rfma(C::a63, m3, rfma(C::a62, m2, rfma(C::a61, m1, mj))))), k6, m6);
#pragma unroll
  for (int s = 0; s < EPL; ++s) {
    k6[s] *= dt;
    Pe[s] = rfma(C::b6, k6[s], rfma(C::b5, k5[s], rfma(C::b4, k4[s], rfma(C::b3, k3[s], rfma(C::b1, k1[s], Pe[s])))));
  }
  m6 *= dt;
  mj = rfma(C::b6, m6, rfma(C::b5, m5, rfma(C::b4, m4, rfma(C::b3, m3, rfma(C::b1, m1, mj)))));
}

// ONE: R is diagonal (WgArgs::r_diag): a single factorisation.  The log-likelihood wants log det S and v^T S^-1 v of the UN-jittered
// S = Sb - 1e-9 I (TFP's factor in the reference, inference_ekf.py:285-286), the gain psd_solve's Sb = S + 1e-9 I; to first order in
// eps = 1e-9 (the next order is eps^2 |Sb^-1|^2: below 1e-14 of a step's term for R >= 1e-2)
//     log det S = log det Sb - eps tr(Sb^-1),        v^T S^-1 v = v^T Sb^-1 v + eps |Sb^-1 v|^2,
// and with a diagonal R both corrections fall out of what the gain's solve leaves behind: Sb^-1 = (I - X_oo)(R + eps I)^-1 on the
// observed block (X = Sb^-1 H P), so tr(Sb^-1) = sum_c (1 - X_cc) / (R_cc + eps), and P Sb^-1 v = X^T v is the mean's increment, so
// (Sb^-1 v)_c = (v_c - (X^T v)_c) / (R_cc + eps).  The factorisation is 29 % of a d = 40 step with two systems in lockstep, 13 % with one.
template <typename R, int D, bool ONE = false>
__global__ __launch_bounds__(256, 1) void ekf_filter_wave_l96_kernel(const WgArgs<R> a) {
  using W = W40<D>;
  using Tile = W40Tile<R>;
  using Lin = W40Lin<R, D>;
  constexpr int EPL = W::EPL, LDP = W::LDP, LDY = W::LDY;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  R* shQ = reinterpret_cast<R*>(smem_raw);
  R* shR = shQ + 64 * EPL;
  R* Wb = shQ + W::SH + (long)wave * W::o_end;
  static_assert(W::o_L1 % 2 == 0 && W::o_L2 % 2 == 0 && W::o_end % 2 == 0 && W::SH % 2 == 0, "16-byte aligned regions");
  R* buf = static_cast<R*>(__builtin_assume_aligned(Wb + W::o_buf, 16));
  R* L1 = static_cast<R*>(__builtin_assume_aligned(Wb + W::o_L1, 16));
  R* L2 = static_cast<R*>(__builtin_assume_aligned(Wb + W::o_L2, 16));
  R* xs = Wb + W::o_xs;
  R* ca = Wb + W::o_ca;
  R* cb = Wb + W::o_cb;
  R* inv2 = Wb + W::o_i2;
  R* vv = Wb + W::o_v;
  const long n = (long)blockIdx.x * W::kWaves + wave;

  // ---- per-lane constants: the owned entries e = lane + 64 s of the packed upper triangle ----------------------------------------
  // (the entry's constants (L Qc L^T)_ij and R_ij live once per workgroup in LDS, slot-major so that a wavefront reads a
  //  contiguous run; the per-entry addresses other than the stage image's are recomputed from (i, j) where they are used --
  //  registers are what the scheduler needs to keep LDS reads in flight)
  unsigned* tabA = reinterpret_cast<unsigned*>(shQ + W::SHQ);
  unsigned* tabB = tabA + 64 * EPL;
  int* obs = reinterpret_cast<int*>(shQ + W::SHT);
  const int M = a.m;
  R Pe[EPL];
  {
    const R* P0 = a.par + a.o_P0;
    const R* LQL = a.par + a.o_LQL;
    const R* Rm = a.par + a.o_R;
    // The emission selects M <= D state components (every row of H a unit vector, no two alike, no bias: wave40_shape).  The update
    // runs in STATE coordinates on the order-D system  S_full = [P + R on the observed components; the identity on the others]  with
    // zero innovation and zero right-hand-side rows there: its factor is [chol(S); I], its solution [S^-1 H P; 0] -- the same numbers
    // as the M x M system of inference_ekf.py:153-199, with M = D and H = I the system this kernel was written for.
    if (wave == 0) {
      int r_obs = -1;
      if (lane < D) {
        const R* Hm = a.par + a.o_H;
        for (int r = 0; r < M; ++r)
          if (Hm[r * D + lane] != R(0)) r_obs = r;
      }
      obs[lane] = r_obs;
      wave_sync();
    }
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
      const int e = lane + 64 * s;
      const bool own = e < W::NP;
      int i = 0, rs = 0;  // row i starts at rs = i D - i (i - 1) / 2
      while (i + 1 < D && e >= rs + (D - i)) {
        rs += D - i;
        ++i;
      }
      const int j = own ? i + (e - rs) : i;
      Pe[s] = own ? R(0.5) * (P0[i * D + j] + P0[j * D + i]) : R(0);
      if (wave == 0) {
        const int oi = obs[i], oj = obs[j];
        shQ[64 * s + lane] = own ? LQL[i * D + j] : R(0);
        shR[64 * s + lane] = !own ? R(0) : ((oi >= 0 && oj >= 0) ? Rm[oi * M + oj] : (i == j ? R(1) : R(0)));
        tabA[64 * s + lane] = (unsigned)i | (unsigned)j << 8 | (unsigned)(oi >= 0) << 16 | (unsigned)(oj >= 0) << 17;
        tabB[64 * s + lane] = (unsigned)(i * LDY + j) | (unsigned)(j * LDY + i) << 11 | (unsigned)(W::rs(j) + i) << 22;
      }
    }
  }
  __syncthreads();
  if (n >= a.N) return;  // whole wavefront; no workgroup barrier anywhere below
  // the owned entries' indices come from the table where they are used (registers are for the Runge-Kutta slopes and the panels)
  struct Ent { int i, j; };
  auto entry = [&](int s) { const unsigned w = tabA[64 * s + lane]; return Ent{(int)(w & 255u), (int)((w >> 8) & 255u)}; };
  struct Off { int y, yt, l; };
  auto offsets = [&](int s) { const unsigned w = tabB[64 * s + lane]; return Off{(int)(w & 2047u), (int)((w >> 11) & 2047u), (int)(w >> 22)}; };
  const bool isrow = lane < D;
  const int lp1 = (lane + 1 >= D) ? lane + 1 - D : lane + 1, lm1 = (lane == 0) ? D - 1 : lane - 1,
            lm2 = (lane <= 1) ? lane + D - 2 : lane - 2;
  const R forcing = (a.par + a.o_theta)[0];
  R mj = isrow ? (a.par + a.o_m0)[lane] : R(0);
  const int rowi = (lane <= D) ? lane : D;       // row of the augmented system this lane factors (lanes > D shadow row D)
  const int ri = W::rs(rowi);
  LlAcc ll;
  int st = 0;
  bool bad = false;
  // diagnostic build aid (scripts/prof_w40.sh, -DCDKF_W40_PROFILE builds ONLY): a.forecast carries a mask of phases to SKIP
  // (results are then meaningless).  The shipped library compiles the mask to a constant zero: no environment variable can
  // switch a phase off.
#ifdef CDKF_W40_PROFILE
  const int skip = a.forecast;
#else
  constexpr int skip = 0;
#endif
  if ((skip & 32) && wave) return;  // (one wavefront per CU: how much do the four of a workgroup cost each other?)

  // ---- streams -----------------------------------------------------------------------------------------------------------
  const R* tp = a.t + n * a.t_sn;
  const int myobs = isrow ? obs[lane] : -1;  // the emission row that observes this lane's state component (-1: none)
  const R rdd = (myobs >= 0) ? (a.par + a.o_R)[myobs * M + myobs] : R(1);  // this component's diagonal entry of R (state coordinates)
  const R* yp = a.y + n * a.y_sn + (myobs >= 0 ? myobs : 0) * a.y_si;
  R tcur = tp[0];
  R ynext = yp[0];

  // stage image of (mean, covariance) in LDS; returns the slopes of the owned entries and of the lane's mean component
  auto rhs = [&](const R (&Ps)[EPL], const R xm, R (&kP)[EPL], R& kM) {
    int ei[EPL], ej[EPL], offP[EPL];
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
      const Ent e = entry(s);
      ei[s] = e.i;
      ej[s] = e.j;
      offP[s] = (e.i + 2) * LDP + (e.j + 2);
      if (W::owned(s, lane)) {
        buf[offP[s]] = Ps[s];
        buf[(e.j + 2) * LDP + (e.i + 2)] = Ps[s];
      }
    }
    if (isrow) xs[lane] = xm;
    wave_sync();
    // halo of the image: rows -2, -1 <- D-2, D-1; row D <- 0; the same for the columns (corners are never read)
    for (int e = lane; e < 3 * D; e += 64) {
      const int r = (e >= 2 * D) ? 2 : (e >= D ? 1 : 0), c = e - r * D;
      const int src = (r == 2) ? 0 : D - 2 + r, dst = (r == 2) ? D : r - 2;
      buf[(dst + 2) * LDP + (c + 2)] = buf[(src + 2) * LDP + (c + 2)];
      buf[(c + 2) * LDP + (dst + 2)] = buf[(c + 2) * LDP + (src + 2)];
    }
    typedef R Pair __attribute__((ext_vector_type(2)));
    Pair* cab = reinterpret_cast<Pair*>(__builtin_assume_aligned(ca, 16));  // (a_i, b_i) side by side: one 16-byte read per index
    if (isrow) {
      const R xp1 = xs[lp1], xm1 = xs[lm1], xm2 = xs[lm2];
      cab[lane] = Pair{xm1, xp1 - xm2};
      kM = rfma(xp1 - xm2, xm1, forcing - xm);
    } else {
      kM = R(0);
    }
    wave_sync();
    // operands of four entries at a time go into registers before any of their arithmetic (forty LDS reads in flight: a lone
    // wavefront has no other wave to cover the ~100-cycle round trip, and the compiler otherwise waits after every read)
    constexpr int CH = 4;
#pragma unroll
    for (int s0 = 0; s0 < EPL; s0 += CH) {
      R o6[CH][6], c4[CH][4], qv[CH];
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int s = (s0 + u < EPL) ? s0 + u : EPL - 1;
        const R* c = buf + offP[s];
        o6[u][0] = c[-2 * LDP];
        o6[u][1] = c[-LDP];
        o6[u][2] = c[LDP];
        o6[u][3] = c[-2];
        o6[u][4] = c[-1];
        o6[u][5] = c[1];
        const Pair ci = cab[ei[s]], cj = cab[ej[s]];
        c4[u][0] = ci[0];
        c4[u][1] = ci[1];
        c4[u][2] = cj[0];
        c4[u][3] = cj[1];
        qv[u] = shQ[64 * s + lane];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        if (s0 + u < EPL) {
          const int s = s0 + u;
          R k = rfma(R(-2), Ps[s], qv[u]);
          k = rfma(c4[u][0], o6[u][2] - o6[u][0], k);
          k = rfma(c4[u][1], o6[u][1], k);
          k = rfma(c4[u][2], o6[u][5] - o6[u][3], k);
          k = rfma(c4[u][3], o6[u][4], k);
          kP[s] = W::owned(s, lane) ? k : R(0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    wave_sync();  // the image is rewritten by the next stage
  };

  // stream a full d x d matrix / d-vector of this step out of LDS images (row-major image with leading dimension ld)
  auto store_cov = [&](R* dst, long k, const R* img, int ld, int) {
    if (!dst || (skip & 16)) return;
    R* o = dst + n * a.P_sn + k * a.P_sk;
    // five reads in flight, then their stores (all twenty-five at once cost registers the sweep does not have: 57 -> 48 ms)
    constexpr int NE = (D * D + 63) / 64, CHK = 5;
#pragma unroll 1
    for (int q0 = 0; q0 < NE; q0 += CHK) {
      R v[CHK];
#pragma unroll
      for (int u = 0; u < CHK; ++u) {
        const int e = lane + 64 * (q0 + u), r = e / D, c = e - r * D;
        v[u] = (e < D * D) ? img[r * ld + c] : R(0);
      }
#pragma unroll
      for (int u = 0; u < CHK; ++u) {
        const int e = lane + 64 * (q0 + u);
        if (e < D * D) o[(long)e * a.P_si] = v[u];
      }
    }
  };

  W40_TICK_DECL
  for (long k = 0; k < a.T; ++k) {
    const R yk = ynext;
    const R tnext_obs = (k + 1 < a.T) ? tp[(k + 1) * a.t_sk] : tcur;
    if (k + 1 < a.T) ynext = yp[(k + 1) * a.y_sk];

    // =================================== update ===================================================================================
    // S = P + R into both packed systems (row d: the innovation), P itself into the image (rows = columns of the solves)
    int oY[EPL], oYT[EPL];  // the owned entries' two positions in the (symmetric) image, kept for the whole update
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
      const Off f = offsets(s);
      oY[s] = f.y;
      oYT[s] = f.yt;
      if (W::owned(s, lane)) {
        const unsigned wA = tabA[64 * s + lane];
        const bool obs_i = (wA >> 16) & 1u, obs_j = (wA >> 17) & 1u;
        const R sv = (obs_i && obs_j) ? Pe[s] + shR[64 * s + lane] : shR[64 * s + lane];
        if constexpr (!ONE) L1[f.l] = sv;
        L2[f.l] = (f.y == f.yt) ? sv + R(1e-9) : sv;
        buf[f.y] = obs_j ? Pe[s] : R(0);   // row i = right-hand side i = column i of H P in state coordinates: component j of it
        buf[f.yt] = obs_i ? Pe[s] : R(0);
      }
    }
    if (isrow) {
      const R v = (myobs >= 0) ? yk - mj : R(0);
      if constexpr (!ONE) L1[W::rs(D) + lane] = v;
      L2[W::rs(D) + lane] = v;
      vv[lane] = v;
    }
    for (int e = lane; e < (48 - D) * LDY; e += 64) buf[D * LDY + e] = R(0);  // rows d .. 47 of the image: zero operands of the tiles
    wave_sync();
    // both factorisations in lockstep (W40Lin::cholesky): system 0 = S with the innovation as its augmented row (TFP's factor: the
    // log-likelihood's log-determinant and quadratic form), system 1 = S + 1e-9 I (psd_solve's factor: the gain)
    R quad = R(0);
    if (!(skip & 1)) {
      W40_TICK(0)
      if constexpr (ONE) {
        R* const sys[1] = {L2};
        R* const scr[1] = {ca};
        Lin::template cholesky<1>(sys, scr, inv2, rowi, ri, lane, quad, ll.ll, bad W40_TICK_PASS);
      } else {
        R* const sys[2] = {L1, L2};
        R* const scr[2] = {ca, cb};  // (the predict step's coefficient vectors: free during the update)
        Lin::template cholesky<2>(sys, scr, inv2, rowi, ri, lane, quad, ll.ll, bad W40_TICK_PASS);
      }
    }
    ll.ll += -0.5 * (double)w40_readlane(quad, D) - 0.5 * M * 1.8378770664093454835606594728112;

    // gain: lane c < D solves (L2 L2^T) x = P[:, c] in place in its row of the image (W40Lin::solve).  After the forward pass the
    // image holds Y (row c = column c of Y = L_b^-1 P), after the backward pass X: each is consumed by a rank-d product on the
    // matrix cores where it stands -- T = Y^T Y - 1e-9 X^T X; the A- and B-operand layouts of v_mfma_*_16x16x4 coincide, so one
    // read serves both sides.
    typename Tile::V4 acc[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) acc[q] = typename Tile::V4{0, 0, 0, 0};
    const int lm = lane & 15, lg = lane >> 4;
    auto rank_update = [&](const R scaleA) {
#pragma unroll
      for (int ks = 0; ks < (D + 3) / 4; ++ks) {
        const int kk = 4 * ks + lg;
        const bool kin = kk < D;
        R op[3];
#pragma unroll
        for (int tI = 0; tI < 3; ++tI) op[tI] = kin ? buf[(16 * tI + lm) * LDY + kk] : R(0);
        int q = 0;
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
          for (int nt = mt; nt < 3; ++nt) {
            acc[q] = wg_mfma(scaleA * op[mt], op[nt], acc[q]);
            ++q;
          }
      }
    };
    R dotv = R(0);  // X^T v of this lane's column
    if (!(skip & 2))
      dotv = Lin::solve(buf, L2, inv2, vv, lane, [&] {
        if (!(skip & 4)) rank_update(R(1));
        wave_sync();
      } W40_TICK_PASS);
    if constexpr (ONE) {  // the first-order corrections of the log-likelihood's two terms (header): + eps tr(Sb^-1) / 2 - eps |Sb^-1 v|^2 / 2
      R c1 = R(0);
      if (isrow) {
        const R rinv = R(1) / (rdd + R(1e-9));
        const R w = (myobs >= 0) ? (vv[lane] - dotv) * rinv : R(0);  // (an unobserved component: (Sb^-1 v)_c = 0; its X^T v is the gain's increment)
        c1 = rfma(-w, w, (R(1) - buf[lane * LDY + lane]) * rinv);
      }
      cb[lane] = c1;
      wave_sync();
      R s0 = R(0), s1 = R(0), s2 = R(0), s3 = R(0);
#pragma unroll
      for (int c = 0; c < D; c += 4) {
        s0 += cb[c];
        s1 += cb[c + 1];
        s2 += cb[c + 2];
        s3 += cb[c + 3];
      }
      ll.ll += 0.5e-9 * (double)((s0 + s1) + (s2 + s3));
    }
    // m+ = m + X^T v
    if (isrow) {
      mj += dotv;
      if (mj != mj) st |= kStatusNan;
    }
    if (!(skip & 4)) rank_update(R(-1e-9));
    wave_sync();
    // tiles -> image (row-major, leading dimension LDY) -> owners
    {
      int q = 0;
#pragma unroll
      for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int nt = mt; nt < 3; ++nt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * mt + Tile::row(lg, r), colx = 16 * nt + lm;
            if (row < D && colx < D) buf[row * LDY + colx] = acc[q][r];
          }
          ++q;
        }
    }
    wave_sync();
#pragma unroll
    for (int s = 0; s < EPL; ++s)
      if (W::owned(s, lane)) Pe[s] -= buf[oY[s]];
    wave_sync();
    // filtered moments out: full symmetric image first
#pragma unroll
    for (int s = 0; s < EPL; ++s)
      if (W::owned(s, lane)) {
        buf[oY[s]] = Pe[s];
        buf[oYT[s]] = Pe[s];
      }
    if (a.fm && isrow) a.fm[n * a.m_sn + k * a.m_sk + lane * a.m_si] = mj;
    wave_sync();
    store_cov(a.fP, k, buf, LDY, 0);
    wave_sync();

    W40_TICK(10)
    // =================================== predict ==================================================================================
    const R t1 = (k + 1 < a.T) ? tnext_obs : tcur + a.dt_final;
    {
      R tprev = tcur;
      R tnx = rmin(tcur + a.dt0, t1);
      long steps = 0;
      while (tprev < t1 && !(skip & 8)) {
        if (steps >= a.max_steps) {
          st |= kStatusMaxSteps;
          break;
        }
        const R dt = tnx - tprev;
        w40_dopri5<R, EPL>(rhs, Pe, mj, dt);
        tprev = rmin(tnx, t1);
        const R tn = tnx + a.dt0;
        tnx = (tn > t1 - Tol<R>::v) ? t1 : tn;
        ++steps;
      }
    }
    W40_TICK(11)
    // predicted moments out
    if (a.pm && isrow) a.pm[n * a.m_sn + k * a.m_sk + lane * a.m_si] = mj;
    if (a.pP) {
#pragma unroll
      for (int s = 0; s < EPL; ++s)
        if (W::owned(s, lane)) {
          const Off f = offsets(s);
          buf[f.y] = Pe[s];
          buf[f.yt] = Pe[s];
        }
      wave_sync();
      store_cov(a.pP, k, buf, LDY, 0);
      wave_sync();
    }
    tcur = tnext_obs;
  }
  if (bad) st |= kStatusNotPd;
  if (lane == 0) {
    a.ll[n] = (R)ll.ll;
    if (a.status) a.status[n] = st;
  }
#ifdef CDKF_W40_PROFILE
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    printf("w40 cycles/step:");
    for (int i = 0; i < 12; ++i) printf(" [%d] %lld", i, w40_prof[i] / a.T);
    printf("\n");
  }
#endif
}

// ---- backward sweep of the EKF (RTS) smoother, same model and ownership (inference_ekf.py:363-448, 503-531) ------------------------
// Per interval [t_k, t_k+1] the filtered (m_f, P_f) at t_k are constants: G = F(m_f) + psd_solve(P_f, L Qc L^T)^T and f(m_f) are
// formed once, then  dm = -[f(m_f) + G (m_s - m_f)],  dP = -[G P_s + (G P_s)^T - L Qc L^T]  is integrated over [0, t_k+1 - t_k]
// from the smoothed moments at t_k+1.  One wavefront per trajectory:
//  * P_f streams from HBM into an LDS image; sym(P_f) + 1e-9 I is factored and (L L^T) X = L Qc L^T solved by W40Lin (16-wide blocks,
//    the products on the matrix cores); the solve leaves X^T where G is wanted, the four non-zeros of F's row i are added by lane i;
//  * a stage writes the symmetric image of P_s with (m_s - m_f) as row D, multiplies G by it on the matrix cores (3 x 3 tiles of
//    16 x 16, ten steps of four: 90 v_mfma_*_16x16x4 -- the mean's G (m_s - m_f) is column D of the same product), sends the tiles
//    back through LDS and the owners pick up A_ij + A_ji.  G is dense (the solve fills it), so this is 2 D^3 flops per stage: the
//    matrix cores' time (90 x 64 cycles) is the floor of a stage.
template <int D>
struct W40S {
  using W = W40<D>;
  static constexpr int IMG = ((D + 1) * W::LDY + 1) & ~1;  // rows 0 .. D (row D: the mean's column of the stage product)
  static_assert(IMG >= W::LPK && IMG >= D * W::LDY, "the second region also holds the packed system and the staged P_f");
  static constexpr int o_G = 0, o_B = IMG, o_inv = 2 * IMG, o_xs = o_inv + 64, o_col = o_xs + 64, o_end = o_col + 64;
  static constexpr int SHQ = 64 * W::EPL, SH = SHQ + 2 * 64 * W::EPL;  // L Qc L^T in ownership order, then the index table
  static constexpr int kWaves = 4;
};
template <int D>
__host__ __device__ constexpr long wave40_smoother_lds_reals() { return (long)W40S<D>::SH + (long)W40S<D>::kWaves * W40S<D>::o_end; }

template <typename R, int D>
__global__ __launch_bounds__(256, 1) void ekf_smoother_wave_l96_kernel(const WgArgs<R> a) {
  using W = W40<D>;
  using S = W40S<D>;
  using Tile = W40Tile<R>;
  using Lin = W40Lin<R, D>;
  using V4 = typename Tile::V4;
  constexpr int EPL = W::EPL, LDY = W::LDY;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  R* shQ = reinterpret_cast<R*>(smem_raw);
  unsigned* tabA = reinterpret_cast<unsigned*>(shQ + S::SHQ);
  unsigned* tabB = tabA + 64 * EPL;
  R* Wb = shQ + S::SH + (long)wave * S::o_end;
  static_assert(S::SH % 2 == 0 && S::o_end % 2 == 0 && S::o_B % 2 == 0, "16-byte aligned regions");
  R* Gm = static_cast<R*>(__builtin_assume_aligned(Wb + S::o_G, 16));  // staged P_f -> right-hand sides -> X^T -> G
  R* Bm = static_cast<R*>(__builtin_assume_aligned(Wb + S::o_B, 16));  // packed system / stage image / product
  R* inv = Wb + S::o_inv;
  R* xs = Wb + S::o_xs;
  R* col = Wb + S::o_col;
  const long n = (long)blockIdx.x * S::kWaves + wave;
  if (wave == 0) {
    const R* LQL = a.par + a.o_LQL;
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
      const int e = lane + 64 * s;
      const bool own = e < W::NP;
      int i = 0, rs = 0;
      while (i + 1 < D && e >= rs + (D - i)) {
        rs += D - i;
        ++i;
      }
      const int j = own ? i + (e - rs) : i;
      shQ[64 * s + lane] = own ? LQL[i * D + j] : R(0);
      tabA[64 * s + lane] = (unsigned)i | (unsigned)j << 8;
      tabB[64 * s + lane] = (unsigned)(i * LDY + j) | (unsigned)(j * LDY + i) << 11 | (unsigned)(W::rs(j) + i) << 22;
    }
  }
  __syncthreads();
  if (n >= a.N) return;
  const bool isrow = lane < D;
  const int lp1 = (lane + 1 >= D) ? lane + 1 - D : lane + 1, lm1 = (lane == 0) ? D - 1 : lane - 1,
            lm2 = (lane <= 1) ? lane + D - 2 : lane - 2;
  const R forcing = (a.par + a.o_theta)[0];
  const int rowi = (lane <= D) ? lane : D, ri = W::rs(rowi);
  const int lm = lane & 15, lg = lane >> 4;
  const R* tp = a.t + n * a.t_sn;
  const long mo = n * a.m_sn + (isrow ? lane : 0) * a.m_si;
  const R* fPn = a.fP + n * a.P_sn;
  int st = 0;
  bool bad = false;
#ifdef CDKF_W40_PROFILE
  const int skip = a.forecast;  // diagnostic mask of phases to skip (scripts/prof_w40.sh; profiling builds only)
#else
  constexpr int skip = 0;
#endif

  int oY[EPL], oYT[EPL];  // the owned entries' two positions in a symmetric image
#pragma unroll
  for (int s = 0; s < EPL; ++s) {
    const unsigned w = tabB[64 * s + lane];
    oY[s] = (int)(w & 2047u);
    oYT[s] = (int)((w >> 11) & 2047u);
  }
  auto store_cov = [&](long k) {  // the symmetric image in Bm -> sP[k]
    R* o = a.sP + n * a.P_sn + k * a.P_sk;
    constexpr int NE = (D * D + 63) / 64, CHK = 5;
#pragma unroll 1
    for (int q0 = 0; q0 < NE; q0 += CHK) {
      R v[CHK];
#pragma unroll
      for (int u = 0; u < CHK; ++u) {
        const int e = lane + 64 * (q0 + u), r = e / D, c = e - r * D;
        v[u] = (e < D * D) ? Bm[r * LDY + c] : R(0);
      }
#pragma unroll
      for (int u = 0; u < CHK; ++u) {
        const int e = lane + 64 * (q0 + u);
        if (e < D * D) o[(long)e * a.P_si] = v[u];
      }
    }
  };

  // smoothed moments at the last time = the filtered ones
  R Ps[EPL];
  R ms = isrow ? a.fm[mo + (a.T - 1) * a.m_sk] : R(0);
  {
    const R* src = fPn + (a.T - 1) * a.P_sk;
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
      const unsigned w = tabA[64 * s + lane];
      const int i = (int)(w & 255u), j = (int)(w >> 8);
      Ps[s] = W::owned(s, lane) ? src[(long)(i * D + j) * a.P_si] : R(0);
      if (W::owned(s, lane)) {
        Bm[oY[s]] = Ps[s];
        Bm[oYT[s]] = Ps[s];
      }
    }
    if (isrow) a.sm[mo + (a.T - 1) * a.m_sk] = ms;
    wave_sync();
    store_cov(a.T - 1);
    wave_sync();
  }

  R t1 = tp[(a.T - 1) * a.t_sk];
  for (long k = a.T - 2; k >= 0; --k) {
    const R t0 = tp[k * a.t_sk];
    const R mf = isrow ? a.fm[mo + k * a.m_sk] : R(0);
    {  // P_f -> image (row-major)
      const R* src = fPn + k * a.P_sk;
      constexpr int NE = (D * D + 63) / 64;
      R v[NE];
#pragma unroll
      for (int q = 0; q < NE; ++q) {
        const int e = lane + 64 * q;
        v[q] = (e < D * D) ? src[(long)e * a.P_si] : R(0);
      }
#pragma unroll
      for (int q = 0; q < NE; ++q) {
        const int e = lane + 64 * q, r = e / D, c = e - r * D;
        if (e < D * D) Gm[r * LDY + c] = v[q];
      }
    }
    if (isrow) xs[lane] = mf;
    wave_sync();
    // S = sym(P_f) + 1e-9 I in packed lower storage (row D: zeros -- W40Lin factors D + 1 rows); f(m_f) and the Jacobian's row
#pragma unroll
    for (int s = 0; s < EPL; ++s)
      if (W::owned(s, lane)) {
        const unsigned w = tabB[64 * s + lane];
        const R sv = R(0.5) * (Gm[oY[s]] + Gm[oYT[s]]);
        Bm[(int)(w >> 22)] = (oY[s] == oYT[s]) ? sv + R(1e-9) : sv;
      }
    if (isrow) Bm[W::rs(D) + lane] = R(0);
    R fa = R(0), fb = R(0), fmf = R(0);  // a_i = x_{i-1}, b_i = x_{i+1} - x_{i-2}: F_i,i-2 = -a_i, F_i,i-1 = b_i, F_ii = -1, F_i,i+1 = a_i
    if (isrow) {
      const R xp1 = xs[lp1], xm1 = xs[lm1], xm2 = xs[lm2];
      fa = xm1;
      fb = xp1 - xm2;
      fmf = rfma(fb, fa, forcing - mf);
    }
    wave_sync();
    // right-hand sides L Qc L^T (row c of the image = column c)
#pragma unroll
    for (int s = 0; s < EPL; ++s)
      if (W::owned(s, lane)) {
        const R q = shQ[64 * s + lane];
        Gm[oY[s]] = q;
        Gm[oYT[s]] = q;
      }
    wave_sync();
    {
      R* const sys[1] = {Bm};
      R* const scr[1] = {col};
      R quad = R(0);
      double logdet = 0.0;
      long long w40_last = 0;
      (void)w40_last;
      if (!(skip & 1)) Lin::template cholesky<1>(sys, scr, inv, rowi, ri, lane, quad, logdet, bad W40_TICK_PASS);
      if (!(skip & 2)) Lin::solve(Gm, Bm, inv, (const R*)nullptr, lane, [] {} W40_TICK_PASS);
    }
    // G = F(m_f) + X^T: the image holds X^T already
    if (isrow) {
      R* g = Gm + lane * LDY;
      g[lm2] -= fa;
      g[lm1] += fb;
      g[lane] -= R(1);
      g[lp1] += fa;
    }
    wave_sync();

    auto rhs = [&](const R (&Pst)[EPL], const R xm, R (&kP)[EPL], R& kM) {
#pragma unroll
      for (int s = 0; s < EPL; ++s)
        if (W::owned(s, lane)) {
          Bm[oY[s]] = Pst[s];
          Bm[oYT[s]] = Pst[s];
        }
      if (isrow) Bm[D * LDY + lane] = xm - mf;
      wave_sync();
      V4 acc[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) acc[q] = V4{0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < D / 4; ++ks) {
        const int kk = 4 * ks + lg;
        R og[3], op[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const int row = 16 * t + lm;
          og[t] = (row < D) ? Gm[(row < D ? row : 0) * LDY + kk] : R(0);
          op[t] = (row <= D) ? Bm[(row <= D ? row : 0) * LDY + kk] : R(0);
        }
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
          for (int nt = 0; nt < 3; ++nt) acc[3 * mt + nt] = wg_mfma(og[mt], op[nt], acc[3 * mt + nt]);
      }
      wave_sync();
#pragma unroll
      for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * mt + Tile::row(lg, r), colx = 16 * nt + lm;
            if (row < D && colx <= D) Bm[row * LDY + colx] = acc[3 * mt + nt][r];
          }
      wave_sync();
#pragma unroll
      for (int s = 0; s < EPL; ++s)
        kP[s] = W::owned(s, lane) ? shQ[64 * s + lane] - (Bm[oY[s]] + Bm[oYT[s]]) : R(0);
      kM = isrow ? -(fmf + Bm[lane * LDY + D]) : R(0);
      wave_sync();
    };
    {
      const R tend = t1 - t0;
      R tprev = R(0);
      R tnx = rmin(a.dt0, tend);
      long steps = 0;
      while (tprev < tend && !(skip & 8)) {
        if (steps >= a.max_steps) {
          st |= kStatusMaxSteps;
          break;
        }
        w40_dopri5<R, EPL>(rhs, Ps, ms, tnx - tprev);
        tprev = rmin(tnx, tend);
        const R tn = tnx + a.dt0;
        tnx = (tn > tend - Tol<R>::v) ? tend : tn;
        ++steps;
      }
    }
#pragma unroll
    for (int s = 0; s < EPL; ++s)
      if (W::owned(s, lane)) {
        Bm[oY[s]] = Ps[s];
        Bm[oYT[s]] = Ps[s];
      }
    if (isrow) {
      a.sm[mo + k * a.m_sk] = ms;
      if (ms != ms) st |= kStatusNan;
    }
    wave_sync();
    if (!(skip & 16)) store_cov(k);
    wave_sync();
    t1 = t0;
  }
  if (bad) st |= kStatusNotPd;
  if (lane == 0 && a.status && st) atomicOr(&a.status[n], st);
}

}  // namespace cdkf
