// cdkf_wave8s_kernels.h -- the MLP-drift EKF sweep for state_dim <= 8 with ONE TRAJECTORY SPREAD OVER NW WAVEFRONTS (NW = 2 or 4).
//
// Why ("config 5: fix occupancy, not loops"): at 1024 trajectories per GPU the
// wavefront-per-trajectory sweep of cdkf_wave8_kernels.h puts one wavefront on each of the 1024 SIMDs.  Its matrix-core loops run at
// the pipe's own rate, but between them sit the LDS-synchronised phases of a right-hand side (layer 1 + tanh, layer 2's tanh, k_P, g,
// the stage combination) with nothing to overlap: the pipe is busy 42 % of the time.  Here a workgroup of NW wavefronts owns ONE
// trajectory -- each wavefront takes 4 / NW of the four sixteen-row tiles of W2 (tangent product) and of the four sixteen-column tiles
// of the transposed product, its slices of W2 pinned in registers in both operand layouts (W2 is not in LDS at all: 31 KB per workgroup)
// -- and 4 such workgroups share a compute unit, so every SIMD interleaves the wavefronts of NW different trajectories: one
// trajectory's latency phases run under another's products.  The chain of a right-hand side shrinks by (1 - 1 / NW) of its
// matrix-core time; three workgroup barriers per right-hand side replace three wavefront fences.
//
// Division of labour inside a workgroup (wavefront 0 = "lead"):
//   every wavefront, redundantly (cheaper than a barrier): layer 1 and its tanh, layer 2's tanh, s, s2 -- lane = hidden unit, results in
//       the wavefront's OWN LDS copy, so no wavefront waits for another's activations;
//   every wavefront, its share: 64 / NW products of [T | z2] = W2 [D1 W1 | a1], 16 / NW chained products of [F | f] = W3 [D2 T | a2]
//       (a partial sum over its rows of T -- they are its accumulators), 64 / NW products of [E1 | tc]^T = [diag(d2) W3^T | s2]^T W2;
//   the lead alone: the 8 x 8 work (stage combinations, k_P = F P + P F^T + L Qc L^T, g and 0.5 P g, the measurement update, the stores,
//       the checkpoints of the reverse sweep).  The others wait at the next barrier and cost their SIMD nothing.
// Barriers per right-hand side: B0 stage value in LDS -> B1 tangent tiles in LDS -> B2 partial Jacobians / E1 in LDS.
//
// Arithmetic: the same sums as the wavefront-per-trajectory sweep except that [F | f] is the sum of NW partial chains (each in the
// k order of its tiles) instead of one chain of sixteen -- 1 ulp apart; parity is against the oracle (tests/test_gpu_wg.py).
// Reference functions restated: extended_kalman_filter and helpers, inference_ekf.py:46-148, 153-199, 202-326.
#pragma once
#include "cdkf_wave8_kernels.h"

namespace cdkf {

template <int NW>
struct W8sOff {
  // 8 x 8 tiles and vectors of the lead: the same offsets as W8Off (w8_measurement_update works on them)
  static constexpr int base_end = W8Off::base_end;  // 608
  static constexpr int s2z = base_end;      // column 8 of the tangent product: z2 - b2   [64]
  static constexpr int U = s2z + 64;        // [64][9]: the rows of T (read by every wavefront's layer-2 phase), later w1 * tq (lead)
  static constexpr int E = U + 576;         // [64][9]: [E1 | tc] from the transposed product
  static constexpr int F2 = E + 576;        // NW partial sums of [F | f], [8][9] each (80 apart)
  static constexpr int rk = F2 + 80 * NW;   // Dormand-Prince a[sg][jj] as a 6 x 6 table
  static constexpr int pw = rk + 36;        // per-wavefront copies of a1, d1, a2, d2, s2 (64 each)
  static constexpr int pw_size = 320;
  static constexpr int a1 = 0, d1 = 64, a2 = 128, d2 = 192, s2 = 256;
  static constexpr int W1 = pw + NW * pw_size;  // weights, zero-padded: W1 [64][9] (a lane's row and the B-operand reads conflict-free)
  static constexpr int b1 = W1 + 576, b2 = b1 + 64, W3 = b2 + 64, b3 = W3 + 9 * 65;  // (W3: a ninth row of zeros for the lanes lm >= 8)
  static constexpr int lql = b3 + 8;  // L Qc L^T on the lane grid [64]
  static constexpr int end = lql + 64;
};
template <int NW>
__host__ __device__ inline long wave8s_lds_reals() {
  return W8sOff<NW>::end;
}

// pick element [w] of a small register array by a wavefront-uniform index (a run-time register index would go through scratch)
template <typename R, int K>
CDKF_DEV R w8s_pick(const R (&v)[K], int w) {
  R r = v[0];
#pragma unroll
  for (int q = 1; q < K; ++q) r = (w == q) ? v[q] : r;
  return r;
}

template <typename R, int NW, bool SECOND>
__global__ __launch_bounds__(64 * NW, NW == 2 ? 2 : 4) void ekf_filter_wave8s_kernel(const WgArgs<R> a) {
  static_assert(NW == 2 || NW == 4, "two or four wavefronts per trajectory");
  constexpr int MT = 4 / NW;  // sixteen-row / sixteen-column tiles per wavefront
  // Register budget: 512 / (wavefronts per SIMD) = 256 (NW = 2) or 128 (NW = 4) per lane, and the two W2 slices alone are 64 reals.  In
  // fp64 the small weights (rows of W1, columns of W3, their operand-layout copies) therefore come from LDS at each use -- conflict-free
  // layouts, requested ahead of the products that need them -- instead of sitting in registers for the whole sweep.
  constexpr bool SMALL_IN_REGS = sizeof(R) == 4 && NW == 2;
#ifndef CDKF_W8S_LEADACT
#define CDKF_W8S_LEADACT 1
#endif
  // LEADACT: the two activation phases (layer 1 + tanh; layer 2's tanh, s, s2) on the lead alone, behind one more barrier, instead of
  // redundantly on every wavefront: an MFMA holds its SIMD's vector issue for all of its cycles on this part (scripts/mb/mb_overlap.hip), so
  // with two wavefronts per SIMD every redundant vector instruction is paid in full -- a barrier's wait is what the partner can hide
  constexpr bool LEADACT = CDKF_W8S_LEADACT != 0;
  constexpr bool LEAN = !SMALL_IN_REGS;  // registers are short: addresses recomputed per right-hand side, the tangent re-read from LDS
  using O = W8sOff<NW>;
  using MTile = W8Tile<R>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  R* W = reinterpret_cast<R*>(smem_raw);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  const bool lead = wave == 0;
  int i = lane >> 3, j = lane & 7;
  const int d = a.d, m = a.m;
  const int h1 = a.h1, h2 = a.h2;
  const R* th = a.par + a.o_theta;
  const R* gW1 = th;
  const R* gb1 = gW1 + h1 * d;
  const R* gW2 = gb1 + h1;
  const R* gb2 = gW2 + h2 * h1;
  const R* gW3 = gb2 + h2;
  const R* gb3 = gW3 + d * h2;
  R* PW = W + O::pw + (LEADACT ? 0 : wave) * O::pw_size;  // (LEADACT: one copy, the lead's)
  bool inP = (i < d) && (j < d);

  // ---- the small weights (zero-padded) in LDS, the tableau table, zeros in the slope tiles ---------------------------------------------
  for (int e = threadIdx.x; e < O::end; e += blockDim.x) W[e] = 0;
  __syncthreads();
  for (int e = threadIdx.x; e < h1 * d; e += blockDim.x) W[O::W1 + fdiv(e, d) * 9 + (e - fdiv(e, d) * d)] = gW1[e];
  for (int e = threadIdx.x; e < h1; e += blockDim.x) W[O::b1 + e] = gb1[e];
  for (int e = threadIdx.x; e < h2; e += blockDim.x) W[O::b2 + e] = gb2[e];
  for (int e = threadIdx.x; e < d * h2; e += blockDim.x) W[O::W3 + fdiv(e, h2) * 65 + (e - fdiv(e, h2) * h2)] = gW3[e];
  for (int e = threadIdx.x; e < d; e += blockDim.x) W[O::b3 + e] = gb3[e];
  if (lead && inP) W[O::lql + lane] = (a.par + a.o_LQL)[i * d + j];
  if (lead && lane < 36) {
    using TBi = Dp5T<R>;
    const int r = lane / 6, c = lane - 6 * r;
    R v = 0;
#pragma unroll
    for (int rr = 1; rr < 6; ++rr)
#pragma unroll
      for (int cc = 0; cc < 5; ++cc)
        if (rr == r && cc == c && cc < rr) v = TBi::a[rr][cc];
    W[O::rk + lane] = v;
  }
  __syncthreads();
  const long n = blockIdx.x;  // one trajectory per workgroup

  // ---- per-lane constants -----------------------------------------------------------------------------------------------------------
  const bool hsel = a.hsel != 0;
  // Matrix-core operand layouts as in cdkf_wave8_kernels.h (A[m = lane & 15][k = lane >> 4], B[k][n = lane & 15]); this wavefront's
  // tiles are mt = MT wave + local index:
  //   w2A[mt][ks] = W2[16 (MT wave + mt) + lm][4 ks + lg]               A operand of the tangent product, its rows
  //   w2B[it][nt] = W2[prow(it)][16 (MT wave + nt) + lm]                B operand of the transposed product, its columns
  //   w1B[ks]     = W1[4 ks + lg][lm] (lm < 8),  w3A[mt][r] = W3[lm][16 mt + row(lg, r)] (lm < 8; all four tiles: the transposed
  //                 product's A operand runs over every hidden unit)
  int lm = lane & 15, lg = lane >> 4;
  auto prow = [&](int it) __attribute__((always_inline)) { return 16 * (it >> 2) + MTile::row(lg, it & 3); };
  const R* gW2p = a.par + a.o_w2pad;  // W2 zero-padded to [64][64] (launch_wg.hip: wg_prepare)
  auto w2 = [&](int p, int q) __attribute__((always_inline)) { return gW2p[p * 64 + q]; };
  int lmc = lm < 8 ? lm : 8;  // (column 8 of W1's rows and row 8 of W3 are zeros)
  R w2A[MT][16], w2B[SECOND ? 16 : 1][MT];
  R w1row_r[SMALL_IN_REGS ? kW8 : 1], w3col_r[SMALL_IN_REGS ? kW8 : 1], w1B_r[SMALL_IN_REGS ? 16 : 1], w3A_r[SMALL_IN_REGS ? 16 : 1];
  auto w1row = [&](int k) __attribute__((always_inline)) {
    if constexpr (SMALL_IN_REGS) return w1row_r[k]; else return W[O::W1 + lane * 9 + k];
  };
  auto w3col = [&](int k) __attribute__((always_inline)) {
    if constexpr (SMALL_IN_REGS) return w3col_r[k]; else return W[O::W3 + k * 65 + lane];
  };
  auto w1B = [&](int ks) __attribute__((always_inline)) {
    if constexpr (SMALL_IN_REGS) return w1B_r[ks]; else return W[O::W1 + (4 * ks + lg) * 9 + lmc];
  };
  auto w3A = [&](int mt, int r) __attribute__((always_inline)) {
    if constexpr (SMALL_IN_REGS) return w3A_r[4 * mt + r]; else return W[O::W3 + lmc * 65 + 16 * mt + MTile::row(lg, r)];
  };
  if constexpr (SMALL_IN_REGS) {
#pragma unroll
    for (int jj = 0; jj < kW8; ++jj) {
      w1row_r[jj] = pin(W[O::W1 + lane * 9 + jj]);
      w3col_r[jj] = pin(W[O::W3 + jj * 65 + lane]);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) w3A_r[4 * mt + r] = pin(W[O::W3 + lmc * 65 + 16 * mt + MTile::row(lg, r)]);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) w1B_r[ks] = pin(W[O::W1 + (4 * ks + lg) * 9 + lmc]);
  }
  // fp64: the slices are (re)loaded after every measurement update instead of living through it -- the update's triangular solves
  // need the registers, and 64 KB per wavefront and observation from L2 is 2 % of a step's time; fp32 has room to keep them throughout.
#ifndef CDKF_W8S_W2_PER_STEP
#define CDKF_W8S_W2_PER_STEP 0
#endif
  constexpr bool W2_PER_STEP = CDKF_W8S_W2_PER_STEP && sizeof(R) == 8;
  auto load_w2 = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) w2A[mt][ks] = pin(w2(16 * (MT * wave + mt) + lm, 4 * ks + lg));
    if constexpr (SECOND) {
#pragma unroll
      for (int it = 0; it < 16; ++it)
#pragma unroll
        for (int nt = 0; nt < MT; ++nt) w2B[it][nt] = pin(w2(prow(it), 16 * (MT * wave + nt) + lm));
    }
  };
  if constexpr (!W2_PER_STEP) load_w2();
  const R b1r = W[O::b1 + lane], b2r = W[O::b2 + lane], lqlr = W[O::lql + lane];  // (held in registers where there is room: !LEAN)
  R e8 = (lm == 8) ? R(1) : R(0), ne8 = (lm == 8) ? R(0) : R(1);
  int sc_off = (lm == 8) ? O::a2 : O::d2;  // column 8 of the second product carries a2 (the drift itself)
  int off2 = (lm == 8) ? O::s2 : O::d2;
  // The lane index is laundered through an empty asm at the top of every right-hand side (as cdkf_adjoint_kernels.h does): what derives
  // from it -- some forty LDS addresses of tiles, images and operand slices -- is then recomputed where it is used instead of being
  // hoisted out of the sweep's loops, where each address held a register for the whole sweep
  auto fresh = [&]() __attribute__((always_inline)) {
    if constexpr (!LEAN) return;
    asm volatile("" : "+v"(lane));
    i = lane >> 3;
    j = lane & 7;
    inP = (i < d) && (j < d);
    lm = lane & 15;
    lg = lane >> 4;
    lmc = lm < 8 ? lm : 8;
    e8 = (lm == 8) ? R(1) : R(0);
    ne8 = (lm == 8) ? R(0) : R(1);
    sc_off = (lm == 8) ? O::a2 : O::d2;
    off2 = (lm == 8) ? O::s2 : O::d2;
  };
  constexpr bool second = SECOND;  // (state_order 'second': the transposed product and its W2 slice exist in this instantiation only)
  const bool zeroth = a.order == 0;

  // ---- state (lead) -----------------------------------------------------------------------------------------------------------------
  R Pij = inP ? R(0.5) * ((a.par + a.o_P0)[i * d + j] + (a.par + a.o_P0)[j * d + i]) : R(0);
  R mj = (lane < d) ? (a.par + a.o_m0)[lane] : R(0);
  double ll = 0.0;
  int st = 0;
  bool bad = false;

  R* mck = nullptr;  // MLP stage checkpoint of the right-hand side in hand (reverse sweep's forward pass only; uniform; lead writes)
  bool want_rows = false;
  // right-hand side of the moment ODEs for the stage value (lead: xs = mean on lanes < d, Ps = this lane's covariance entry)
  auto rhs = [&](R xs, R Ps, R& kM, R& kP) __attribute__((always_inline)) {
    fresh();
    if (lead) {
      W[W8Off::P + lane] = Ps;
      if (lane < kW8) W[W8Off::x + lane] = xs;
    }
    if constexpr (!LEADACT) __syncthreads();  // B0: the stage value
    if (!LEADACT || lead) {
      if constexpr (LEADACT) wave_sync();
      R xk[kW8];
#pragma unroll
      for (int k = 0; k < kW8; ++k) xk[k] = W[W8Off::x + k];
      // ---- layer 1: lane = hidden unit q (!LEADACT: every wavefront, into its own copy) ------------------------------------------------
      R z1 = LEAN ? W[O::b1 + lane] : b1r;
#pragma unroll
      for (int k = 0; k < kW8; ++k) z1 = rfma(w1row(k), xk[k], z1);
      const R a1 = rtanh_fast(z1);
      const R d1 = R(1) - a1 * a1;
      PW[O::a1 + lane] = a1;
      PW[O::d1 + lane] = d1;
      if (lead && mck) mck[kMlpCkA1 * 64 + lane] = a1;
    }
    if constexpr (LEADACT) __syncthreads();  // B0: layer 1's activations
    else wave_sync();
    // ---- this wavefront's tiles of [T | z2 - b2] = W2 [D1 W1 | a1] -------------------------------------------------------------------
    typename MTile::V4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = typename MTile::V4{0, 0, 0, 0};
    {
      constexpr int LA = sizeof(R) == 4 ? 8 : 2;  // k-steps of look-ahead for the two activations of a k-step
      R dq[16], aq[16], wq[16];
#pragma unroll
      for (int ks = 0; ks < LA; ++ks) {
        dq[ks] = PW[O::d1 + 4 * ks + lg];
        aq[ks] = PW[O::a1 + 4 * ks + lg];
        wq[ks] = w1B(ks);
      }
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        R bv = rfma(dq[ks], wq[ks], aq[ks] * e8);
        if (ks + LA < 16) {
          dq[ks + LA] = PW[O::d1 + 4 * (ks + LA) + lg];
          aq[ks + LA] = PW[O::a1 + 4 * (ks + LA) + lg];
          wq[ks + LA] = w1B(ks + LA);
        }
        asm volatile("" : "+v"(bv) : : "memory");
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = wg_mfma(w2A[mt][ks], bv, acc[mt]);
      }
    }
    if (lm == 8) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) W[O::s2z + 16 * (MT * wave + mt) + MTile::row(lg, r)] = acc[mt][r];
    }
    if ((LEAN || want_rows) && lm < kW8) {  // the rows of T: grad(div f) and the checkpoint read them in lane = hidden-unit order, layer 3 (LEAN) in tile order
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) W[O::U + (16 * (MT * wave + mt) + MTile::row(lg, r)) * 9 + lm] = acc[mt][r];
    }
    __syncthreads();  // B1: the tangent tiles of all wavefronts
    // ---- layer 2 activations, s_p = sum_i W3[i][p] T[p][i], s2 = -2 a2 d2 s (!LEADACT: every wavefront, into its own copy) ---------------
    if (!LEADACT || lead) {
      const R z2 = W[O::s2z + lane] + (LEAN ? W[O::b2 + lane] : b2r);
      const R a2 = rtanh_fast(z2);
      const R d2 = R(1) - a2 * a2;
      PW[O::a2 + lane] = a2;
      PW[O::d2 + lane] = d2;
      if (want_rows) {
        R sdiv = 0;
#pragma unroll
        for (int k = 0; k < kW8; ++k) {
          const R tk = W[O::U + lane * 9 + k];
          sdiv = rfma(w3col(k), tk, sdiv);
          if (lead && mck) mck[(kMlpCkT + k) * 64 + lane] = tk;
        }
        if (second) PW[O::s2 + lane] = R(-2) * a2 * d2 * sdiv;
        if (lead && mck) {
          mck[kMlpCkA2 * 64 + lane] = a2;
          if (second) mck[kMlpCkS * 64 + lane] = sdiv;
        }
      }
    }
    if constexpr (LEADACT) __syncthreads();  // B1': layer 2's activations
    else wave_sync();
    // ---- this wavefront's share of layer 3 (a partial sum over its rows of T) and its column tiles of the transposed product ---------------
    typename MTile::V4 acc3{0, 0, 0, 0};
    typename MTile::V4 cacc[MT];
#pragma unroll
    for (int nt = 0; nt < MT; ++nt) cacc[nt] = typename MTile::V4{0, 0, 0, 0};
    {
      auto sc_of = [&](int q) __attribute__((always_inline)) { return PW[sc_off + 16 * (MT * wave + (q >> 2)) + MTile::row(lg, q & 3)]; };
      // this wavefront's rows of T, back from their LDS image (the accumulators that held them are free from the barrier on; column 8 of
      // the operand is a2 itself: lmc = 8 reads the image's unused ninth column, which the e8 blend below ignores)
      auto t_of = [&](int q) __attribute__((always_inline)) { return W[O::U + (16 * (MT * wave + (q >> 2)) + MTile::row(lg, q & 3)) * 9 + lmc]; };
      auto w3o_of = [&](int q) __attribute__((always_inline)) {  // W3 slice of this wavefront's own tile (q >> 2), k-step q & 3
        if constexpr (SMALL_IN_REGS) {
          R cand[NW];
#pragma unroll
          for (int w = 0; w < NW; ++w) cand[w] = w3A(MT * w + (q >> 2), q & 3);
          return w8s_pick<R, NW>(cand, wave);
        } else {
          return W[O::W3 + lmc * 65 + 16 * (MT * wave + (q >> 2)) + MTile::row(lg, q & 3)];
        }
      };
      if constexpr (SECOND) {
        constexpr int LA = sizeof(R) == 4 ? 8 : 2;
        R x2q[16], w3q[16], scq[4 * MT], w3oq[4 * MT], tq_[4 * MT];
#pragma unroll
        for (int it = 0; it < LA; ++it) {
          x2q[it] = PW[off2 + prow(it)];
          w3q[it] = w3A(it >> 2, it & 3);
          if (it < 4 * MT) {
            scq[it] = sc_of(it);
            w3oq[it] = w3o_of(it);
            if constexpr (LEAN) tq_[it] = t_of(it);
          }
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
          R av = x2q[it] * (w3q[it] + e8);  // lm < 8: d2_p W3[lm][p];  lm = 8: s2_p;  else 0
          if (it + LA < 16) {
            x2q[it + LA] = PW[off2 + prow(it + LA)];
            w3q[it + LA] = w3A((it + LA) >> 2, (it + LA) & 3);
            if (it + LA < 4 * MT) {
              scq[it + LA] = sc_of(it + LA);
              w3oq[it + LA] = w3o_of(it + LA);
              if constexpr (LEAN) tq_[it + LA] = t_of(it + LA);
            }
          }
          if (it < 4 * MT) {
            R b3v = scq[it] * rfma(LEAN ? tq_[it] : acc[it >> 2][it & 3], ne8, e8);
            R w3v = w3oq[it];
            asm volatile("" : "+v"(av), "+v"(b3v), "+v"(w3v) : : "memory");
            acc3 = wg_mfma(w3v, b3v, acc3);
          } else {
            asm volatile("" : "+v"(av) : : "memory");
          }
#pragma unroll
          for (int nt = 0; nt < MT; ++nt) cacc[nt] = wg_mfma(av, w2B[it][nt], cacc[nt]);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4 * MT; ++q) acc3 = wg_mfma(w3o_of(q), sc_of(q) * rfma(LEAN ? t_of(q) : acc[q >> 2][q & 3], ne8, e8), acc3);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = MTile::row(lg, r);
      if (row < kW8 && lm < 9) W[O::F2 + 80 * wave + row * 9 + lm] = acc3[r];
    }
    if (second) {
#pragma unroll
      for (int nt = 0; nt < MT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = MTile::row(lg, r);
          if (row < 9) W[O::E + (16 * (MT * wave + nt) + lm) * 9 + row] = cacc[nt][r];
        }
    }
    __syncthreads();  // B2: the partial Jacobians and E1 | tc of all wavefronts
    if (!lead) return;
    // ---- the lead: F, f, k_P = F P + P F^T + L Qc L^T, 0.5 P grad(div f) -------------------------------------------------------------------
    {
      R fs = 0, fv = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        fs += W[O::F2 + 80 * w + i * 9 + j];
        if (lane < kW8) fv += W[O::F2 + 80 * w + lane * 9 + 8];
      }
      W[W8Off::F + lane] = fs;
      if (lane < kW8) kM = (lane < d) ? fv + W[O::b3 + lane] : R(0);
      if (mck) mck[kMlpCkF * 64 + lane] = inP ? fs : R(0);
    }
    wave_sync();
    if (!zeroth) {
      // (F P)_ij and (F P)_ji -- the second sum is, term by term and in the same order, what lane (j, i) forms as its first:
      // k_P stays exactly symmetric without waiting for the partner's value
      R sa = 0, sb = 0;
#pragma unroll
      for (int k = 0; k < kW8; ++k) {
        sa = rfma(W[W8Off::F + i * kW8 + k], W[W8Off::P + k * kW8 + j], sa);
        sb = rfma(W[W8Off::F + j * kW8 + k], W[W8Off::P + k * kW8 + i], sb);
      }
      kP = (sa + sb) + (LEAN ? W[O::lql + lane] : lqlr);
    }
    if (second) {
      // tq, then g_i = sum_q W1[q][i] tq_q: lane (i, j) adds the hidden units q = 8 c + j, the eight partial sums of a grid row meet by
      // DPP; 0.5 (P g)_j = 0.5 sum_i P_ij g_i over the grid rows by the half-row rotation and the two swaps -- no LDS round trip
      R td = 0, w1r[kW8];
#pragma unroll
      for (int k = 0; k < kW8; ++k) w1r[k] = w1row(k);
#pragma unroll
      for (int k = 0; k < kW8; ++k) {
        const R ek = W[O::E + lane * 9 + k];
        td = rfma(w1r[k], ek, td);
        if (mck) mck[(kMlpCkE1 + k) * 64 + lane] = ek;
      }
      const R a1l = PW[O::a1 + lane], d1l = PW[O::d1 + lane];  // (the lead's own copies; a1 / d1 above went out of registers long ago)
      const R tqv = d1l * rfma(R(-2) * a1l, td, W[O::E + lane * 9 + 8]);
      if (mck) {
        mck[kMlpCkTd * 64 + lane] = td;
        mck[kMlpCkTq * 64 + lane] = tqv;
      }
#pragma unroll
      for (int k = 0; k < kW8; ++k) W[O::U + lane * 8 + k] = w1r[k] * tqv;  // (the rows of T were consumed before B2)
      wave_sync();
      R part = 0;
#pragma unroll
      for (int c = 0; c < 8; ++c) part += W[O::U + (8 * c + j) * 8 + i];
      const R gi = w8_sum_j(part);  // g_i in the lanes (i, *)
      if (mck && j == 0) mck[kMlpCkG * 64 + i] = (i < d) ? gi : R(0);
      if (!zeroth) {
        const R pg = w8_sum_i(Ps * gi);  // (P g)_j in the lanes (*, j)
        if (lane < kW8) kM = rfma(R(0.5), pg, kM);
      }
    }
  };

  const R* tp = a.t + n * a.t_sn;
  const R* yp = a.y + n * a.y_sn;
  using C = Dp5<R>;
  for (long k = 0; k < a.T; ++k) {
    if (lead) {
      // ---------------- update (inference_ekf.py:153-199, 285-286) and the filtered stores ----------------
      // (the update's per-lane constants are fetched here, once per observation, instead of occupying registers through the sweep)
      const R Hij = (i < m && j < d) ? (a.par + a.o_H)[i * d + j] : R(0);
      const R Rij = (i < m && j < m) ? (a.par + a.o_R)[i * m + j] : R(0);
      const R hbj = (lane < m) ? (a.par + a.o_hb)[lane] : R(0);
      const R yl = (lane < m) ? yp[k * a.y_sk + lane * a.y_si] : R(0);
      w8_measurement_update<R>(W, lane, i, j, d, m, inP, hsel, Hij, Rij, hbj, yl, a.num_iter, a.forecast, Pij, mj, ll, bad);
      if (mj != mj) st |= kStatusNan;
      if (a.fm && lane < d) a.fm[n * a.m_sn + k * a.m_sk + lane * a.m_si] = mj;
      if (a.fP && inP) a.fP[n * a.P_sn + k * a.P_sk + (i * d + j) * a.P_si] = Pij;
    }
    // ---------------- predict (every wavefront walks the same steps; the lead holds the state) ----------------
    if constexpr (W2_PER_STEP) load_w2();
    const R t0 = tp[k * a.t_sk];
    const R t1 = (k + 1 < a.T) ? tp[(k + 1) * a.t_sk] : t0 + a.dt_final;
    {
      R tprev = t0;
      R tnext = rmin(t0 + a.dt0, t1);
      long steps = 0;
      while (tprev < t1) {  // uniform over the workgroup
        if (steps >= a.max_steps) {
          st |= kStatusMaxSteps;
          break;
        }
        const R dt = tnext - tprev;
        R* ckp = (a.ck && k + 1 < a.T && steps < a.ck_smax) ? a.ck + ((n * (a.T - 1) + k) * a.ck_smax + steps) * kCkStep : nullptr;
        R* mckp = (a.ckm && k + 1 < a.T && steps == 0) ? a.ckm + (n * (a.T - 1) + k) * (6L * a.ckm_nf * 64) : nullptr;
        want_rows = second || mckp;
        R kM6 = 0, kP6 = 0;
#pragma unroll 1
        for (int sg = 0; sg < 6; ++sg) {
          R sm = 0, sp = 0;
          if (lead) {
#pragma unroll
            for (int jj = 0; jj < 5; ++jj) {  // (jj >= sg: zero coefficient -- the same sums as the guarded form, without its branches)
              const R c = W[O::rk + 6 * sg + jj];
              sm = rfma(c, W[W8Off::km + 8 * jj + (lane & 7)], sm);
              sp = rfma(c, W[W8Off::X + 64 * jj + lane], sp);
            }
          }
          R kM = 0, kP = 0;
          mck = mckp ? mckp + (long)sg * a.ckm_nf * 64 : nullptr;
          rhs(rfma(dt, sm, mj), rfma(dt, sp, Pij), kM, kP);
          if (lead) {
            if (ckp) {  // slopes for the reverse sweep (uniform branch; nullptr outside cdkf_ekf_loglik_grad_all)
              ckp[sg * 72 + lane] = kP;
              if (lane < kW8) ckp[sg * 72 + 64 + lane] = kM;
            }
            if (sg < 5) {
              W[W8Off::X + 64 * sg + lane] = kP;
              if (lane < kW8) W[W8Off::km + 8 * sg + lane] = kM;
            } else {
              kM6 = kM;
              kP6 = kP;
            }
          }
        }
        if (lead) {
          auto KM = [&](int q) __attribute__((always_inline)) { return W[W8Off::km + 8 * q + (lane & 7)]; };
          auto KP = [&](int q) __attribute__((always_inline)) { return W[W8Off::X + 64 * q + lane]; };
          const R sm = rfma(C::b6, kM6, rfma(C::b5, KM(4), rfma(C::b4, KM(3), rfma(C::b3, KM(2), C::b1 * KM(0)))));
          mj = (lane < kW8) ? rfma(dt, sm, mj) : mj;
          if (!zeroth) Pij = rfma(dt, rfma(C::b6, kP6, rfma(C::b5, KP(4), rfma(C::b4, KP(3), rfma(C::b3, KP(2), C::b1 * KP(0))))), Pij);
        }
        tprev = rmin(tnext, t1);
        const R tn = tnext + a.dt0;
        tnext = (tn > t1 - Tol<R>::v) ? t1 : tn;
        ++steps;
      }
    }
    if (lead) {
      if (zeroth) Pij = rfma(rsqrt_(t1 - t0), inP ? (a.par + a.o_LQLz)[i * d + j] : R(0), Pij);
      if (a.pm && lane < d) a.pm[n * a.m_sn + k * a.m_sk + lane * a.m_si] = mj;
      if (a.pP && inP) a.pP[n * a.P_sn + k * a.P_sk + (i * d + j) * a.P_si] = Pij;
    }
  }
  if (lead && lane == 0) {
    if (bad) st |= kStatusNotPd;
    if (ll != ll) st |= kStatusNan;
    a.ll[n] = (R)ll;
    if (a.status) a.status[n] = st;
  }
}

}  // namespace cdkf
