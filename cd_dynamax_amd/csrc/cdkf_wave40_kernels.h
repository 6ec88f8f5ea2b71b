// cdkf_wave40_kernels.h -- wavefront-per-trajectory EKF filter sweep for the Lorenz-96 model at state dimensions 9 .. 48
// (BASELINE config 4: d = m = 40), emission = the identity on the state (H = I, any symmetric R).
//
// The workgroup-per-trajectory kernel (cdkf_wg2_kernels.h) spends a d = 40 observation step in ~35 barrier-separated phases with a
// few dozen flops per thread in each: 512 threads wait on each other most of the time (SQ_WAIT_ANY 69 % of the wave cycles,
// profiles/r02_c_config4_counters.json).  Here ONE wavefront owns a trajectory for the whole scan and nothing inside the time loop
// waits on another wavefront; four wavefronts (four trajectories) share a workgroup only to fill the CU's four SIMDs.
//
//  * State: the packed upper triangle of P (820 entries at d = 40) dealt round-robin to the 64 lanes -- 13 entries per lane, with
//    their six Dormand-Prince slopes, in registers; lanes < d own the mean.
//  * Predict (inference_ekf.py:76-123): the stage covariance is written to an LDS image with a two-element halo (the Lorenz-96
//    neighbourhood i-2, i-1, i+1 wraps around), so an entry's right-hand side
//        dP_ij = a_i (P_{i+1,j} - P_{i-2,j}) + b_i P_{i-1,j} + a_j (P_{i,j+1} - P_{i,j-2}) + b_j P_{i,j-1} - 2 P_ij + (L Qc L^T)_ij,
//        a_i = x_{i-1},  b_i = x_{i+1} - x_{i-2}                      (the four non-zeros of the Jacobian's row i)
//    is six LDS reads at constant offsets from ONE per-entry address, four coefficient reads and eight flops; two wavefront-scope
//    synchronisations per stage.
//  * Update (inference_ekf.py:153-199, 285-286; H = I so H P = P, S = P + R):
//      - both Cholesky factors (TFP's of S for the log-likelihood, psd_solve's of S + 1e-9 I for the gain) left-looking in packed
//        lower storage by panels of eight columns, lane i = row i, the two recurrences in lockstep; the innovation rides along as
//        row d of the un-jittered system, which makes its forward substitution (the log-likelihood's quadratic form) part of the
//        factorisation;
//      - X = (S + 1e-9 I)^-1 P by forward and backward substitution with lane c = right-hand side c, the forty unknowns of a column in
//        registers (fully unrolled; the factor's entries arrive as LDS broadcasts at compile-time offsets);
//      - m+ = m + X^T (y - m) (lane c: its own column against the broadcast innovation);
//      - P+ = P - X^T S X as P - Y^T Y + 1e-9 X^T X with Y = L_b^-1 P the forward solve's result (S = L_b L_b^T - 1e-9 I): two
//        symmetric rank-d products on the matrix cores (v_mfma_*_16x16x4, six upper 16 x 16 tiles, operands straight from the LDS
//        image of Y / X, whose A- and B-operand layouts coincide), the -1e-9 folded into the A operand.  Needs R symmetric.
//  * Outputs stream from the LDS images (full d x d rows, coalesced).
//
// Scope: drift Lorenz-96, H = I (m = d), num_iter = 1, state_order first / second (the same for this drift: grad(div f) = 0), fixed-
// step Dormand-Prince, no forecast.  Everything else stays on cdkf_wg2_kernels.h; CDKF_NO_WAVE40=1 forces that (A/B, tests).
#pragma once
#include "cdkf_wave8_kernels.h"

namespace cdkf {

template <int D>
struct W40 {
  static_assert(D > 8 && D <= 48, "three 16-wide tiles");
  static constexpr int NP = D * (D + 1) / 2;        // packed upper triangle
  static constexpr int EPL = (NP + 63) / 64;        // entries per lane
  static constexpr int LDP = D + 4;                 // stage image: rows / columns -2 .. D (halo), leading dimension
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
  static constexpr int SH = 2 * 64 * EPL;  // per workgroup: (L Qc L^T) and R entries in ownership order [s][lane]
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
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
CDKF_DEV float w40_readlane(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }

template <typename R, int D>
__global__ __launch_bounds__(256, 1) void ekf_filter_wave_l96_kernel(const WgArgs<R> a) {
  using W = W40<D>;
  using Tile = W40Tile<R>;
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
  int offP[EPL], ei[EPL], ej[EPL];
  bool own[EPL];
  R Pe[EPL];
  {
    const R* P0 = a.par + a.o_P0;
    const R* LQL = a.par + a.o_LQL;
    const R* Rm = a.par + a.o_R;
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
      const int e = lane + 64 * s;
      own[s] = e < W::NP;
      int i = 0, rs = 0;  // row i starts at rs = i D - i (i - 1) / 2
      while (i + 1 < D && e >= rs + (D - i)) {
        rs += D - i;
        ++i;
      }
      const int j = own[s] ? i + (e - rs) : i;
      ei[s] = i;
      ej[s] = j;
      offP[s] = (i + 2) * LDP + (j + 2);
      Pe[s] = own[s] ? R(0.5) * (P0[i * D + j] + P0[j * D + i]) : R(0);
      if (wave == 0) {
        shQ[64 * s + lane] = own[s] ? LQL[i * D + j] : R(0);
        shR[64 * s + lane] = own[s] ? Rm[i * D + j] : R(0);
      }
    }
  }
  __syncthreads();
  if (n >= a.N) return;  // whole wavefront; no workgroup barrier anywhere below
  auto offT = [&](int s) { return (ej[s] + 2) * LDP + (ei[s] + 2); };   // transposed position in the stage image
  auto offY = [&](int s) { return ei[s] * LDY + ej[s]; };               // update image, row-major
  auto offYT = [&](int s) { return ej[s] * LDY + ei[s]; };
  auto offL = [&](int s) { return W::rs(ej[s]) + ei[s]; };              // packed lower: row j, column i (i <= j)
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
  // diagnostic build aid (scripts/prof_w40.sh): a.forecast carries a mask of phases to SKIP (results are then meaningless); the
  // launcher passes 0 unless CDKF_W40_ABLATE is set.  Uniform run-time branches: the shipped path pays nothing for them.
  const int skip = a.forecast;
  if ((skip & 32) && wave) return;  // (one wavefront per CU: how much do the four of a workgroup cost each other?)

  // ---- streams -----------------------------------------------------------------------------------------------------------
  const R* tp = a.t + n * a.t_sn;
  const R* yp = a.y + n * a.y_sn + (isrow ? lane : 0) * a.y_si;
  R tcur = tp[0];
  R ynext = yp[0];

  // stage image of (mean, covariance) in LDS; returns the slopes of the owned entries and of the lane's mean component
  auto rhs = [&](const R (&Ps)[EPL], const R xm, R (&kP)[EPL], R& kM) {
#pragma unroll
    for (int s = 0; s < EPL; ++s)
      if (own[s]) {
        buf[offP[s]] = Ps[s];
        buf[offT(s)] = Ps[s];
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
    if (isrow) {
      const R xp1 = xs[lp1], xm1 = xs[lm1], xm2 = xs[lm2];
      ca[lane] = xm1;
      cb[lane] = xp1 - xm2;
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
        c4[u][0] = ca[ei[s]];
        c4[u][1] = cb[ei[s]];
        c4[u][2] = ca[ej[s]];
        c4[u][3] = cb[ej[s]];
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
          kP[s] = own[s] ? k : R(0);
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
    for (int e = lane; e < D * D; e += 64) {
      const int r = fdiv(e, D), c = e - r * D;
      o[(long)e * a.P_si] = img[r * ld + c];
    }
  };

  for (long k = 0; k < a.T; ++k) {
    const R yk = ynext;
    const R tnext_obs = (k + 1 < a.T) ? tp[(k + 1) * a.t_sk] : tcur;
    if (k + 1 < a.T) ynext = yp[(k + 1) * a.y_sk];

    // =================================== update ===================================================================================
    // S = P + R into both packed systems (row d: the innovation), P itself into the image (rows = columns of the solves)
#pragma unroll
    for (int s = 0; s < EPL; ++s)
      if (own[s]) {
        const R sv = Pe[s] + shR[64 * s + lane];
        L1[offL(s)] = sv;
        L2[offL(s)] = (ei[s] == ej[s]) ? sv + R(1e-9) : sv;
        buf[offY(s)] = Pe[s];
        buf[offYT(s)] = Pe[s];
      }
    if (isrow) {
      const R v = yk - mj;
      L1[W::rs(D) + lane] = v;
      L2[W::rs(D) + lane] = v;
      vv[lane] = v;
    }
    for (int e = lane; e < (48 - D) * LDY; e += 64) buf[D * LDY + e] = R(0);  // rows d .. 47 of the image: zero operands of the tiles
    wave_sync();
    // left-looking factorisation of both systems by panels of eight columns, lane i = row i (row d: forward substitution of the
    // innovation).  A panel's eight entries of the lane's row sit in registers for both systems (sixteen independent chains): the
    // contributions of the finished columns stream in from LDS (the lane's own row + broadcasts of the panel's rows), inside the
    // panel the multipliers L[8p+q][c] are other lanes' registers and arrive through v_readlane -- no LDS round trip, one
    // wavefront synchronisation per panel.
    R quad = R(0);
    for (int p = 0; p < ((skip & 1) ? 0 : D / 8); ++p) {
      const int c0 = 8 * p;
      R u1[8], u2[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        u1[r] = L1[ri + c0 + r];
        u2[r] = L2[ri + c0 + r];
      }
      for (int t0 = 0; t0 < c0; t0 += 2) {  // two finished columns at a time: 36 LDS reads in flight, then 32 multiply-adds
        R a1[2], a2[2], b1[2][8], b2[2][8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          a1[q] = L1[ri + t0 + q];
          a2[q] = L2[ri + t0 + q];
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const int rr = W::rs(c0 + r) + t0 + q;
            b1[q][r] = L1[rr];
            b2[q][r] = L2[rr];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            u1[r] = rfma(-a1[q], b1[q][r], u1[r]);
            u2[r] = rfma(-a2[q], b2[q][r], u2[r]);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      R pinv = R(1);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int c = c0 + r;
        const R p1 = w40_readlane(u1[r], c), p2 = w40_readlane(u2[r], c);
        bad = bad || !(p1 > R(0)) || !(p2 > R(0));
        const R r1 = w40_rsqrt(p1), r2 = w40_rsqrt(p2);
        u1[r] *= r1;  // L[i][c] for the rows below the pivot
        u2[r] *= r2;
        if (lane == D) quad = rfma(u1[r], u1[r], quad);  // z_c of the log-likelihood's forward substitution
        if (lane == 0) inv2[c] = r2;
        pinv *= r1;
#pragma unroll
        for (int q = r + 1; q < 8; ++q) {
          const R b1 = w40_readlane(u1[r], c0 + q), b2 = w40_readlane(u2[r], c0 + q);  // L[c0+q][c]
          u1[q] = rfma(-u1[r], b1, u1[q]);
          u2[q] = rfma(-u2[r], b2, u2[q]);
        }
      }
      ll.ll += log((double)pinv);  // sum of the logs of the reciprocal pivots, eight at a time
#pragma unroll
      for (int r = 0; r < 8; ++r)
        if (rowi > c0 + r) {
          L1[ri + c0 + r] = u1[r];
          L2[ri + c0 + r] = u2[r];
        }
      wave_sync();
    }
    ll.ll += -0.5 * (double)w40_readlane(quad, D) - 0.5 * D * 1.8378770664093454835606594728112;

    // gain: lane c < D solves (L2 L2^T) x = P[:, c] in place in its row of the image, eight unknowns at a time in registers: the
    // contributions of the rows already solved stream through (their unknowns from the lane's own row, the factor's entries as
    // broadcasts), the 8 x 8 triangle inside a block is unrolled.  After the forward pass the image holds Y (row c = column c of
    // Y = L_b^-1 P), after the backward pass X: each is consumed by a rank-d product on the matrix cores where it stands --
    // T = Y^T Y - 1e-9 X^T X; the A- and B-operand layouts of v_mfma_*_16x16x4 coincide, so one read serves both sides.
    static_assert(D % 8 == 0, "blocks of eight unknowns");
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
    R* mine = buf + (isrow ? lane : 0) * LDY;
    for (int b = 0; b < ((skip & 2) ? 0 : D / 8); ++b) {  // forward: L2 y = p
      R u[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) u[r] = mine[8 * b + r];
      for (int t0 = 0; t0 < 8 * b; t0 += 4) {  // four solved unknowns at a time: 36 LDS reads in flight, then 32 multiply-adds
        R yt[4], lv[4][8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          yt[q] = mine[t0 + q];
#pragma unroll
          for (int r = 0; r < 8; ++r) lv[q][r] = L2[W::rs(8 * b + r) + t0 + q];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int r = 0; r < 8; ++r) u[r] = rfma(-lv[q][r], yt[q], u[r]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
#pragma unroll
        for (int q = 0; q < r; ++q) u[r] = rfma(-L2[W::rs(8 * b + r) + 8 * b + q], u[q], u[r]);
        u[r] *= inv2[8 * b + r];
      }
      if (isrow) {
#pragma unroll
        for (int r = 0; r < 8; ++r) mine[8 * b + r] = u[r];
      }
    }
    wave_sync();
    if (!(skip & 4)) rank_update(R(1));
    wave_sync();
    R dotv = R(0);  // X^T v of this lane's column
    for (int b = ((skip & 2) ? -1 : D / 8 - 1); b >= 0; --b) {  // backward: L2^T x = y
      R u[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) u[r] = mine[8 * b + r];
      for (int t0 = D - 4; t0 >= 8 * b + 8; t0 -= 4) {
        R xt[4], lv[4][8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          xt[q] = mine[t0 + q];
          const R* Lt = L2 + W::rs(t0 + q) + 8 * b;
#pragma unroll
          for (int r = 0; r < 8; ++r) lv[q][r] = Lt[r];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int r = 0; r < 8; ++r) u[r] = rfma(-lv[q][r], xt[q], u[r]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 7; r >= 0; --r) {
#pragma unroll
        for (int q = r + 1; q < 8; ++q) u[r] = rfma(-L2[W::rs(8 * b + q) + 8 * b + r], u[q], u[r]);
        u[r] *= inv2[8 * b + r];
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) dotv = rfma(u[r], vv[8 * b + r], dotv);
      if (isrow) {
#pragma unroll
        for (int r = 0; r < 8; ++r) mine[8 * b + r] = u[r];
      }
    }
    // m+ = m + X^T v
    if (isrow) {
      mj += dotv;
      if (mj != mj) st |= kStatusNan;
    }
    wave_sync();
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
      if (own[s]) Pe[s] -= buf[offY(s)];
    wave_sync();
    // filtered moments out: full symmetric image first
#pragma unroll
    for (int s = 0; s < EPL; ++s)
      if (own[s]) {
        buf[offY(s)] = Pe[s];
        buf[offYT(s)] = Pe[s];
      }
    if (a.fm && isrow) a.fm[n * a.m_sn + k * a.m_sk + lane * a.m_si] = mj;
    wave_sync();
    store_cov(a.fP, k, buf, LDY, 0);
    wave_sync();

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
        tprev = rmin(tnx, t1);
        const R tn = tnx + a.dt0;
        tnx = (tn > t1 - Tol<R>::v) ? t1 : tn;
        ++steps;
      }
    }
    // predicted moments out
    if (a.pm && isrow) a.pm[n * a.m_sn + k * a.m_sk + lane * a.m_si] = mj;
    if (a.pP) {
#pragma unroll
      for (int s = 0; s < EPL; ++s)
        if (own[s]) {
          buf[offY(s)] = Pe[s];
          buf[offYT(s)] = Pe[s];
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
}

}  // namespace cdkf
