// cdkf_adjoint_w40_kernels.h -- reverse sweep (value-and-gradient of the marginal log-likelihood w.r.t. the forcing and every leaf of
// the model block) of the Lorenz-96 model at state dimensions 12 .. 40 on ONE WAVEFRONT PER TRAJECTORY: the counterpart of
// ekf_filter_wave_l96_kernel (cdkf_wave40_kernels.h) for the discrete adjoint that ekf_adjoint_wg_kernel runs with a workgroup per
// trajectory.  Same scope as the forward kernel: emission = a selection of state components (H rows of the identity, no bias), any
// symmetric R, num_iter = 1, fixed-step Dormand-Prince.  Oracle: cdkf_oracle.ekf_loglik_grad_adjoint, line by line.
//
// Why: the workgroup kernel spends a d = 40 observation step in ~60 barrier-separated phases of a few dozen flops per thread (375 k
// cycles, one trajectory per CU); the factorisation and the substitutions do not parallelise over 256 threads.  Here a wavefront owns
// the trajectory, nothing inside the time loop waits on another wavefront, and two trajectories share a CU (fp32: four).
//
//  * Update, reversed (inference_ekf.py:153-199 backwards), in STATE coordinates as the forward kernel runs it: the order-D system
//    S_full = [P + R on the observed components; the identity on the others], zero innovation / right-hand-side rows there.
//      - ONE factorisation and ONE substitution sweep: W2 = (sym S + 1e-9 I)^-1 from the identity (W40Lin); psd_solve's two
//        applications are products, X = W2 (E P), Ub = W2 Kb, and the log-likelihood's S^-1 = W2 + 1e-9 W2 W2 (+ O(1e-18 |W2|^3))
//        needs no factor of its own;
//      - ten dense products on the matrix cores (v_mfma_*_16x16x4, 3 x 3 tiles of 16 x 16, operands straight from four LDS images
//        that are reused as their contents die): X, W2 W2, X Pb, (E P) Pb, (X Pb) X^T, W2 Kb, X Ub^T, Ub X^T, Sbar (E P), Ub P;
//        S X Pb is formed as (E P - 1e-9 X) Pb (S X = E P - 1e-9 X by the definition of X), so S itself is never stored;
//      - Sbar, Kb, Ub stay in accumulator registers between the products that form and consume them.
//  * Predict, reversed: per Runge-Kutta step the six stages forward (slopes of the owned entries of the packed upper triangle in
//    registers, as the forward kernel) and backward: the stage cotangent Lam as a symmetric LDS image with a wrap-around halo,
//    Ybar_P = Lam F + (Lam F)^T as a six-point stencil on it (the Jacobian's four non-zeros per row), the state's cotangent from
//    three rows of Lam against four rows of the stage covariance, the weighted sums of the stage cotangents in lane-owned LDS arrays.
//    Intervals of several steps are replayed from the filtered moments in chunks of `cap` step starts (global scratch).
//  * The model block accumulates in global memory (H, R, bias: read-modify-write by the lane that owns the tile entry in every step;
//    the first step writes), LQL / P0 / m0 / the forcing in registers until the end.
#pragma once
#include "cdkf_wave40_kernels.h"

namespace cdkf {

#ifdef CDKF_W40A_PROFILE  // local diagnostic build: cycles per phase (s_memtime), printed by trajectory 0 (scripts/prof_w40a.sh)
__device__ long long w40a_prof[24];
#define W40A_TICK(i)                                                               \
  {                                                                                \
    const long long w40a_now = clock64();                                          \
    if (threadIdx.x == 0 && blockIdx.x == 0) w40a_prof[i] += w40a_now - w40a_last; \
    w40a_last = clock64();                                                         \
  }
#else
#define W40A_TICK(i)
#endif

// NW: wavefronts per trajectory.  1: a wavefront owns a trajectory, two trajectories per workgroup (they share the index tables and
// the entry constants in LDS).  2: a workgroup of two wavefronts owns ONE trajectory -- each wavefront takes every other owned slot of
// the packed triangle and every other 16 x 16 tile of the products; the factorisation, the substitution sweep and the vectors stay on
// the first wavefront -- and two workgroups share a CU (the entry constants move to registers to make the LDS fit twice).
template <typename R, int D, int NW = 1>
struct W40A {
  using W = W40<D>;
  static constexpr int EPL = W::EPL, LDY = W::LDY, LDP = W::LDP;
  static constexpr int NB = (D + 15) / 16, NT = NB * NB;
  static constexpr int IMG = D * LDY;          // an update image: D rows, leading dimension LDY (even)
  static constexpr int SIMG = (D + 4) * LDP;   // a stage image: rows / columns -2 .. D + 1 (wrap-around halo)
  static constexpr int LPK = W::LPK;
  static constexpr int ACC = 64 * EPL;         // a lane-owned array of the owned entries
  static constexpr int NACC = 5;               // weighted sums of the stage cotangents, stages 0 .. 4
  static constexpr int upd = 4 * IMG + LPK, prd = SIMG + IMG + NACC * ACC;
  static constexpr int body = ((upd > prd ? upd : prd) + 1) & ~1;
  static constexpr int NV = 17;
  static constexpr int o_end = body + NV * 64;
  static constexpr int kWaves = 2;                 // wavefronts per workgroup
  static constexpr int kTraj = NW == 1 ? 2 : 1;    // trajectories per workgroup
  // per workgroup: (NW = 1) the (L Qc L^T) and R entries in ownership order; the index tables (two 32-bit words per entry), obs[64]
  static constexpr int TABW = 2 * 64 * EPL + 64;  // 32-bit words
  static constexpr int SHC = NW == 1 ? 2 * 64 * EPL : 0;
  static constexpr int SH = (SHC + (TABW * 4 + (int)sizeof(R) - 1) / (int)sizeof(R) + 1) & ~1;
  static constexpr long lds_reals = (long)SH + (long)kTraj * o_end;
  // global scratch per trajectory: `cap` step starts (owned entries lane-major + the mean)
  __host__ __device__ static constexpr long start_reals() { return 64L * EPL + 64; }
};

// which owned slots / product tiles wavefront H of NW takes
template <int NW, int H>
__host__ __device__ constexpr bool w40a_mine(int q) { return NW == 1 || (q & 1) == H; }
// ... of a product with a SYMMETRIC result: the tiles on and above the diagonal only (the others are their mirrors), dealt in turn
template <int NW, int H, int NB>
__host__ __device__ constexpr bool w40a_mine_sym(int mt, int nt) {
  return mt <= nt && (NW == 1 || ((mt * NB - mt * (mt - 1) / 2 + (nt - mt)) & 1) == H);
}

// acc[mt * NB + nt] += sum_k A(16 mt + (lane & 15), k) B(k, 16 nt + (lane & 15)) over k < D (a multiple of four), the operands'
// addresses formed once: pa[t] walks row 16 t + (lane & 15) of the A image along k (a row that is
// not there points at a row of zeros); pb[t] walks column 16 t + (lane & 15) of the B image down the rows (BROW = false; a column that
// is not there is read at column 0 and dropped by blast_ok) or, for a transposed factor, row 16 t + (lane & 15) along k (BROW = true,
// zeros as for A).  SYM: the result is symmetric -- only the tiles on and above the diagonal are formed.  SA: A is scaled by 1, -1, -1/2, 2 (codes 0 .. 3).  MASKK: rows k of B whose state component is not observed
// count as zero.  k is fully unrolled: every read is an immediate offset from its pointer.
template <typename R, int D, bool BROW, bool MASKK, int SA, int NW, int H, bool SYM = false, typename Acc>
CDKF_DEV void w40a_mmp(Acc& acc, const R* const (&pa)[W40A<R, D>::NB], const R* const (&pb)[W40A<R, D>::NB], const bool blast_ok,
                       const unsigned long long obsmask, const int lg) {
  constexpr int NB = W40A<R, D>::NB, LDY = W40A<R, D>::LDY;
  // (tiles of the other wavefront: neither their products nor -- where a whole tile row / column is the other's -- their operands)
  bool need_a[NB], need_b[NB];
#pragma unroll
  for (int t = 0; t < NB; ++t) need_a[t] = need_b[t] = false;
#pragma unroll
  for (int mt = 0; mt < NB; ++mt)
#pragma unroll
    for (int nt = 0; nt < NB; ++nt)
      if (SYM ? w40a_mine_sym<NW, H, NB>(mt, nt) : w40a_mine<NW, H>(mt * NB + nt)) need_a[mt] = need_b[nt] = true;
  constexpr R sa = SA == 1 ? R(-1) : (SA == 2 ? R(-0.5) : (SA == 3 ? R(2) : R(1)));
#pragma unroll
  for (int ks = 0; ks < D / 4; ++ks) {
    R av[NB], bv[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      av[t] = need_a[t] ? pa[t][4 * ks] : R(0);
      bv[t] = need_b[t] ? (BROW ? pb[t][4 * ks] : pb[t][4 * ks * LDY]) : R(0);
    }
    if constexpr (!BROW && D % 16 != 0) bv[NB - 1] = blast_ok ? bv[NB - 1] : R(0);
    if constexpr (MASKK) {
      const bool ok = (obsmask >> (4 * ks + lg)) & 1ull;
#pragma unroll
      for (int t = 0; t < NB; ++t) bv[t] = ok ? bv[t] : R(0);
    }
    if constexpr (SA != 0) {
#pragma unroll
      for (int t = 0; t < NB; ++t) av[t] *= sa;
    }
#pragma unroll
    for (int mt = 0; mt < NB; ++mt)
#pragma unroll
      for (int nt = 0; nt < NB; ++nt)
        if (SYM ? w40a_mine_sym<NW, H, NB>(mt, nt) : w40a_mine<NW, H>(mt * NB + nt))
          acc[mt * NB + nt] = wg_mfma(av[mt], bv[nt], acc[mt * NB + nt]);
  }
}

template <typename R, int D, int NW, int H>
CDKF_DEV void w40a_sweep(const WgArgs<R>& a, R* __restrict__ grad, R* __restrict__ grad_model, R* __restrict__ ws, const long ws_stride,
                         const int cap, unsigned char* smem_raw) {
  using W = W40<D>;
  using A = W40A<R, D, NW>;
  constexpr bool LEAD = NW == 1 || H == 0;  // the wavefront that factors, substitutes and owns the vectors (lanes = components)
  auto mine = [](int q) constexpr { return w40a_mine<NW, H>(q); };
  // synchronisation of everything that shares the trajectory's LDS: the wavefront itself, or the workgroup's two
  auto sync = [&]() __attribute__((always_inline)) {
    if constexpr (NW == 1)
      wave_sync();
    else
      __syncthreads();
  };
  using Tile = W40Tile<R>;
  using Lin = W40Lin<R, D>;
  using V4 = typename Tile::V4;
  using T = Dp5T<R>;
  constexpr int EPL = W::EPL, LDP = W::LDP, LDY = W::LDY, IMG = A::IMG, NB = A::NB, NT = A::NT;
  static_assert(A::SH % 2 == 0 && A::o_end % 2 == 0 && IMG % 2 == 0 && A::SIMG % 2 == 0 && A::LPK % 2 == 0, "16-byte aligned regions");
  const int wave = threadIdx.x >> 6;
  int lane = threadIdx.x & 63;
  // The lane index is laundered through an empty asm where a phase begins: the index tables are read-only words, so the compiler would
  // otherwise hoist all 2 EPL table reads -- and the dozens of addresses derived from them -- out of the time loop and spill them
  R* shQ = reinterpret_cast<R*>(smem_raw);  // (NW = 1 only: with NW = 2 the entry constants live in registers, Qe / Re below)
  R* shR = shQ + 64 * EPL;
  unsigned* tabA = reinterpret_cast<unsigned*>(shQ + A::SHC);
  unsigned* tabB = tabA + 64 * EPL;
  int* obs = reinterpret_cast<int*>(tabB + 64 * EPL);
  R* Wb = shQ + A::SH + (NW == 1 ? (long)wave * A::o_end : 0L);
  // update: four images and the packed factor; predict: the stage image (halo), the stage covariance, the cotangent sums
  R* I0 = static_cast<R*>(__builtin_assume_aligned(Wb, 16));
  R* I1 = I0 + IMG;
  R* I2 = I1 + IMG;
  R* I3 = I2 + IMG;
  R* Lp = static_cast<R*>(__builtin_assume_aligned(I3 + IMG, 16));
  R* S0 = I0;
  R* PsI = S0 + A::SIMG;
  R* AccL = PsI + IMG;  // weighted sums of the later stages' cotangents (stages 0 .. 4), lane-major
  R* vec = Wb + A::body;
  R *v_xs = vec, *v_ca = vec + 64 /* pairs: 128 */, *v_c3 = vec + 192 /* triples: 192 */, *v_lam = vec + 384, *v_mb = vec + 448,
    *v_v = vec + 512, *v_w = vec + 576, *v_vb = vec + 640, *v_u = vec + 704, *v_m = vec + 768, *v_inv = vec + 832, *v_dt = vec + 896, *v_dummy = vec + 960, *v_zero = vec + 1024;
  const long n = NW == 1 ? (long)blockIdx.x * A::kWaves + wave : (long)blockIdx.x;
  const int M = a.m;

  // ---- tables: the owned entries e = lane + 64 s of the packed upper triangle (as the forward kernel) ------------------------------
  {
    const R* LQL = a.par + a.o_LQL;
    const R* Rm = a.par + a.o_R;
    if (wave == 0) {
      int r_obs = -1;
      if (lane < D) {
        const R* Hm = a.par + a.o_H;
        for (int r = 0; r < M; ++r)
          if (Hm[r * D + lane] != R(0)) r_obs = r;
      }
      obs[lane] = r_obs;
      wave_sync();
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
        const int oi = obs[i], oj = obs[j];
        if constexpr (NW == 1) {
          shQ[64 * s + lane] = own ? LQL[i * D + j] : R(0);
          shR[64 * s + lane] = !own ? R(0) : ((oi >= 0 && oj >= 0) ? Rm[oi * M + oj] : (i == j ? R(1) : R(0)));
        }
        tabA[64 * s + lane] = (unsigned)i | (unsigned)j << 8 | (unsigned)(oi >= 0) << 16 | (unsigned)(oj >= 0) << 17;
        tabB[64 * s + lane] = (unsigned)(i * LDY + j) | (unsigned)(j * LDY + i) << 11 | (unsigned)(W::rs(j) + i) << 22;
      }
    }
  }
  __syncthreads();
  if (n >= a.N) return;  // NW = 1: a whole wavefront, and no workgroup barrier anywhere below; NW = 2: the whole workgroup
  struct Ent { int i, j; };
  auto entry = [&](int s) { const unsigned w = tabA[64 * s + lane]; return Ent{(int)(w & 255u), (int)((w >> 8) & 255u)}; };
  struct Off { int y, yt, l; };
  auto offsets = [&](int s) { const unsigned w = tabB[64 * s + lane]; return Off{(int)(w & 2047u), (int)((w >> 11) & 2047u), (int)(w >> 22)}; };
  bool isrow, blast_ok;
  int lp1, lp2, lm1, lm2, rowi, ri, lm, lg;
  auto fresh = [&]() __attribute__((always_inline)) {
    asm volatile("" : "+v"(lane));
    __builtin_assume(lane >= 0 && lane < 64);
    isrow = lane < D;
    lp1 = (lane + 1 >= D) ? lane + 1 - D : lane + 1;
    lp2 = (lane + 2 >= D) ? lane + 2 - D : lane + 2;
    lm1 = (lane == 0) ? D - 1 : lane - 1;
    lm2 = (lane <= 1) ? lane + D - 2 : lane - 2;
    rowi = (lane <= D) ? lane : D;  // row of the (augmented) system this lane factors
    ri = W::rs(rowi);
    lm = lane & 15;
    lg = lane >> 4;
    blast_ok = 16 * (NB - 1) + lm < D;
  };
  fresh();
  const int myobs = isrow ? obs[lane] : -1;
  const unsigned long long obsmask = __ballot(myobs >= 0);
  auto observed = [&](int i) { return (bool)((obsmask >> i) & 1ull); };
  // the owned entries' constants (L Qc L^T)_ij and R_ij in state coordinates: LDS (NW = 1) or registers (NW = 2)
  R Qe[EPL], Re[EPL];
  if constexpr (NW != 1) {
    const R* LQL = a.par + a.o_LQL;
    const R* Rm = a.par + a.o_R;
#pragma unroll
    for (int s = 0; s < EPL; ++s)
      if (mine(s)) {
        const unsigned w = tabA[64 * s + lane];
        const int i = (int)(w & 255u), j = (int)((w >> 8) & 255u);
        const bool own = W::owned(s, lane);
        const int oi = obs[i], oj = obs[j];
        Qe[s] = own ? LQL[i * D + j] : R(0);
        Re[s] = !own ? R(0) : ((oi >= 0 && oj >= 0) ? Rm[oi * M + oj] : (i == j ? R(1) : R(0)));
      }
  }
  auto q_of = [&](int s) {
    if constexpr (NW == 1)
      return shQ[64 * s + lane];
    else
      return Qe[s];
  };
  auto r_of = [&](int s) {
    if constexpr (NW == 1)
      return shR[64 * s + lane];
    else
      return Re[s];
  };
  const R forcing = (a.par + a.o_theta)[0];
  const R* tp = a.t + n * a.t_sn;
  const R* yp = a.y + n * a.y_sn + (myobs >= 0 ? myobs : 0) * a.y_si;
  R* g = grad + n;  // (the forcing: one drift parameter)
  R* gm = grad_model ? grad_model + n * ((long)D + 2L * D * D + (long)M * D + M + (long)M * M) : nullptr;
  R* gP0 = gm ? gm + D : nullptr;
  R* gQ = gm ? gm + D + (long)D * D : nullptr;
  R* gH = gm ? gm + D + 2L * D * D : nullptr;
  R* gBias = gm ? gH + (long)M * D : nullptr;
  R* gR = gm ? gBias + M : nullptr;
  R* wsb = ws + n * ws_stride;
  R* PsG = wsb + (long)cap * A::start_reals();  // the step's stage inputs 1 .. 5 of the owned entries, lane-major (L2-resident: rewritten every step)
  // Parked in the same scratch (lane-major, 64 EPL reals each): Pbar while the update's adjoint runs (its registers are then free for the
  // products' tiles: held across the update it was spilled for the whole sweep, one reload per use), and d ll / d (L Qc L^T), which is
  // touched once per Runge-Kutta step
  R* PbG = PsG + 5L * 64 * EPL;

  R Pb[EPL];  // cotangent of the covariance: owned entries
#pragma unroll
  for (int s = 0; s < EPL; ++s)
    if (mine(s)) Pb[s] = R(0);
  // Two wavefronts per trajectory (7 owned entries per lane at D = 40): Pbar stays in its registers across the update and d ll / d (L Qc L^T),
  // touched once per Runge-Kutta step, accumulates in registers for the whole sweep (round 5: 230 of 256 accumulator registers, no spill;
  // both were parked in the scratch: 26 KB of traffic per trajectory-step).  One wavefront per trajectory (13 entries per lane) has no room:
  // held there they push the fp64 build from 1 116 to 1 204 B of scratch and 553 spilled registers -- and the -O3 build goes WRONG (one more
  // member of NOTES R5.1's family, caught by test_lorenz96_d40_value_and_gradient) -- so that mapping keeps the parking.
  constexpr bool kPark = NW == 1;
  R* gQG = PbG + 64L * EPL;
  R gQacc[EPL];
#pragma unroll
  for (int s = 0; s < EPL; ++s) {
    gQacc[s] = R(0);
    if constexpr (kPark)
      if (mine(s)) gQG[64 * s + lane] = R(0);
  }
  R mb = R(0), gF = R(0);
  int st = 0;
  bool bad = false;

  // tiles <-> images
  auto tiles_zero = [&](V4 (&acc)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (mine(t)) acc[t] = V4{0, 0, 0, 0};
  };
  auto tiles_store = [&](R* img, const V4 (&acc)[NT]) {
#pragma unroll
    for (int mt = 0; mt < NB; ++mt)
#pragma unroll
      for (int nt = 0; nt < NB; ++nt)
        if (mine(mt * NB + nt)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * mt + Tile::row(lg, r), col = 16 * nt + lm;
            R* p = (row < D && col < D) ? img + row * LDY + col : v_dummy + lane;  // (no predicated stores: a select of the address)
            *p = acc[mt * NB + nt][r];
          }
        }
  };
  auto tiles_store_all = [&](R* img, const V4 (&acc)[NT]) {  // (every tile: a product one wavefront formed alone)
#pragma unroll
    for (int mt = 0; mt < NB; ++mt)
#pragma unroll
      for (int nt = 0; nt < NB; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * mt + Tile::row(lg, r), col = 16 * nt + lm;
          R* p = (row < D && col < D) ? img + row * LDY + col : v_dummy + lane;
          *p = acc[mt * NB + nt][r];
        }
  };
  // ... of a symmetric matrix held as its tiles on and above the diagonal (w40a_mine_sym): a tile off the diagonal is stored twice
  auto mineS = [](int mt, int nt) constexpr { return w40a_mine_sym<NW, H, NB>(mt, nt); };
  auto tilesS_zero = [&](V4 (&acc)[NT]) {
#pragma unroll
    for (int mt = 0; mt < NB; ++mt)
#pragma unroll
      for (int nt = 0; nt < NB; ++nt)
        if (mineS(mt, nt)) acc[mt * NB + nt] = V4{0, 0, 0, 0};
  };
  auto tilesS_store = [&](R* img, const V4 (&acc)[NT]) {
#pragma unroll
    for (int mt = 0; mt < NB; ++mt)
#pragma unroll
      for (int nt = 0; nt < NB; ++nt)
        if (mineS(mt, nt)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * mt + Tile::row(lg, r), col = 16 * nt + lm;
            const bool in = row < D && col < D;
            R* p = in ? img + row * LDY + col : v_dummy + lane;
            *p = acc[mt * NB + nt][r];
            if (mt != nt) {
              R* q = in ? img + col * LDY + row : v_dummy + lane;
              *q = acc[mt * NB + nt][r];
            }
          }
        }
  };
  auto img_at = [&](const R* img, int i, int k) { return i < D ? img[i * LDY + k] : R(0); };  // (k < D by construction)
  // operand walks of the products (w40a_mmp)
  v_zero[lane] = R(0);
  struct Ptr3 { const R* p[NB]; };
  auto rows_of = [&](const R* img, const bool observed_only) {  // row 16 t + lm along k; rows that are not there: zeros
    Ptr3 q;
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      const int i = 16 * t + lm;
      q.p[t] = (i < D && (!observed_only || observed(i))) ? img + i * LDY + lg : v_zero + lg;
    }
    return q;
  };
  auto cols_of = [&](const R* img) {  // column 16 t + lm down the rows
    Ptr3 q;
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      const int j = 16 * t + lm;
      q.p[t] = img + lg * LDY + (j < D ? j : 0);
    }
    return q;
  };
  // acc's entries added to global memory at addr(row, col) (< 0: not there); the old values are all in flight before the first store
  auto tiles_accumulate = [&](R* base, const V4 (&acc)[NT], const bool first_, auto&& addr) {
    int off[NT][4];
    R old[NT][4];
#pragma unroll
    for (int mt = 0; mt < NB; ++mt)
#pragma unroll
      for (int nt = 0; nt < NB; ++nt)
        if (mine(mt * NB + nt)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * mt + Tile::row(lg, r), col = 16 * nt + lm;
            off[mt * NB + nt][r] = (row < D && col < D) ? addr(row, col) : -1;
          }
        }
    if (!first_) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (mine(t)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) old[t][r] = base[off[t][r] >= 0 ? off[t][r] : 0];
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (mine(t)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // (entries that are not there go to a scratch word of this lane: no predicated stores)
          R* p = off[t][r] >= 0 ? base + off[t][r] : wsb + 64 * H + lane;
          *p = first_ ? acc[t][r] : old[t][r] + acc[t][r];
        }
      }
  };

  // ... of a symmetric matrix held as its tiles on and above the diagonal: an entry off the diagonal tiles goes to both of its places
  auto tilesS_accumulate = [&](R* base, const V4 (&acc)[NT], const bool first_, auto&& addr) {
    int off[NT][4], offm[NT][4];
    R old[NT][4], oldm[NT][4];
#pragma unroll
    for (int mt = 0; mt < NB; ++mt)
#pragma unroll
      for (int nt = 0; nt < NB; ++nt)
        if (mineS(mt, nt)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * mt + Tile::row(lg, r), col = 16 * nt + lm;
            const bool in = row < D && col < D;
            off[mt * NB + nt][r] = in ? addr(row, col) : -1;
            offm[mt * NB + nt][r] = (in && mt != nt) ? addr(col, row) : -1;
          }
        }
    if (!first_) {
#pragma unroll
      for (int mt = 0; mt < NB; ++mt)
#pragma unroll
        for (int nt = 0; nt < NB; ++nt)
          if (mineS(mt, nt)) {
            const int t = mt * NB + nt;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              old[t][r] = base[off[t][r] >= 0 ? off[t][r] : 0];
              if (mt != nt) oldm[t][r] = base[offm[t][r] >= 0 ? offm[t][r] : 0];
            }
          }
    }
#pragma unroll
    for (int mt = 0; mt < NB; ++mt)
#pragma unroll
      for (int nt = 0; nt < NB; ++nt)
        if (mineS(mt, nt)) {
          const int t = mt * NB + nt;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            R* p = off[t][r] >= 0 ? base + off[t][r] : wsb + 64 * H + lane;
            *p = first_ ? acc[t][r] : old[t][r] + acc[t][r];
            if (mt != nt) {
              R* q = offm[t][r] >= 0 ? base + offm[t][r] : wsb + 64 * H + lane;
              *q = first_ ? acc[t][r] : oldm[t][r] + acc[t][r];
            }
          }
        }
  };

  // ---- one right-hand side of the moment equations on the stage image (as the forward kernel's) --------------------------------------
  auto rhs = [&](const R (&Ps)[EPL], const R xm, R (&kP)[EPL], R& kM) {
    fresh();
    unsigned wa[EPL];  // the owned entries' (i, j): one batch of table reads per right-hand side
#pragma unroll
    for (int s = 0; s < EPL; ++s)
      if (mine(s)) wa[s] = tabA[64 * s + lane];
#pragma unroll
    for (int s = 0; s < EPL; ++s)
      if (mine(s)) {
        const int ei = (int)(wa[s] & 255u), ej = (int)((wa[s] >> 8) & 255u);
        const bool own = W::owned(s, lane);
        R* l1 = own ? S0 + (ei + 2) * LDP + (ej + 2) : v_dummy + lane;
        R* l2 = own ? S0 + (ej + 2) * LDP + (ei + 2) : v_dummy + lane;
        *l1 = Ps[s];
        *l2 = Ps[s];
      }
    if (LEAD && isrow) v_xs[lane] = xm;
    sync();
    typedef R Pair __attribute__((ext_vector_type(2)));
    Pair* cab = reinterpret_cast<Pair*>(__builtin_assume_aligned(v_ca, 16));
    kM = R(0);
    if constexpr (NW == 1 || H == 1) {  // (two wavefronts: the second copies the halo while the first forms the coefficients)
      for (int e = lane; e < 3 * D; e += 64) {  // halo: rows -2, -1 <- D-2, D-1; row D <- 0; the same for the columns
        const int r = (e >= 2 * D) ? 2 : (e >= D ? 1 : 0), c = e - r * D;
        const int src = (r == 2) ? 0 : D - 2 + r, dst = (r == 2) ? D : r - 2;
        S0[(dst + 2) * LDP + (c + 2)] = S0[(src + 2) * LDP + (c + 2)];
        S0[(c + 2) * LDP + (dst + 2)] = S0[(c + 2) * LDP + (src + 2)];
      }
    }
    if constexpr (LEAD) {
      if (isrow) {
        const R xp1 = v_xs[lp1], xm1 = v_xs[lm1], xm2 = v_xs[lm2];
        cab[lane] = Pair{xm1, xp1 - xm2};
        kM = rfma(xp1 - xm2, xm1, forcing - xm);
      }
    }
    sync();
    constexpr int CH = NW == 1 ? 4 : 8;  // (slots in flight together: four of this wavefront's)
#pragma unroll
    for (int s0 = 0; s0 < EPL; s0 += CH) {
      R o6[CH][6], c4[CH][4], qv[CH];
#pragma unroll
      for (int u = 0; u < CH; ++u)
        if (s0 + u < EPL && mine(s0 + u)) {
        const int s = s0 + u;
        const Ent e{(int)(wa[s] & 255u), (int)((wa[s] >> 8) & 255u)};
        const R* c = S0 + (e.i + 2) * LDP + (e.j + 2);
        o6[u][0] = c[-2 * LDP];
        o6[u][1] = c[-LDP];
        o6[u][2] = c[LDP];
        o6[u][3] = c[-2];
        o6[u][4] = c[-1];
        o6[u][5] = c[1];
        const Pair ci = cab[e.i], cj = cab[e.j];
        c4[u][0] = ci[0];
        c4[u][1] = ci[1];
        c4[u][2] = cj[0];
        c4[u][3] = cj[1];
        qv[u] = q_of(s);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        if (s0 + u < EPL && mine(s0 + u)) {
          const int s = s0 + u;
          R kk = rfma(R(-2), Ps[s], qv[u]);
          kk = rfma(c4[u][0], o6[u][2] - o6[u][0], kk);
          kk = rfma(c4[u][1], o6[u][1], kk);
          kk = rfma(c4[u][2], o6[u][5] - o6[u][3], kk);
          kk = rfma(c4[u][3], o6[u][4], kk);
          kP[s] = W::owned(s, lane) ? kk : R(0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    sync();  // the image is rewritten by the next stage
  };

  // One Dormand-Prince step's stages forward from (P0, x0).  keep: what the reverse pass needs of them are the stage INPUTS -- the
  // covariance parts go to lane-owned arrays in the trajectory's global scratch (stages 1 .. 5; stage 0's is the step's start), the
  // mean's to xin; the slopes die with the call and the sixth right-hand side is not evaluated (nothing reads its slope).  Otherwise
  // (the replay of an interval's earlier steps) the step is completed in place.
  R xin[6];
  auto stages_forward = [&](R (&P0)[EPL], R& x0, const R dt, auto keep_tag) {
    constexpr bool KEEP = decltype(keep_tag)::value;
    R kP[6][EPL], km[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      R Ps[EPL], xs = x0;
#pragma unroll
      for (int u = 0; u < EPL; ++u)
        if (mine(u)) Ps[u] = P0[u];
#pragma unroll
      for (int j = 0; j < 5; ++j)
        if (j < i) {
#pragma unroll
          for (int u = 0; u < EPL; ++u)
            if (mine(u)) Ps[u] = rfma(T::a[i][j], kP[j][u], Ps[u]);
          xs = rfma(T::a[i][j], km[j], xs);
        }
      if constexpr (KEEP) {
        xin[i] = xs;
        if (i > 0) {
#pragma unroll
          for (int u = 0; u < EPL; ++u)
            if (mine(u)) PsG[((i - 1) * EPL + u) * 64 + lane] = Ps[u];
        }
      }
      if (!KEEP || i < 5) {
        rhs(Ps, xs, kP[i], km[i]);
#pragma unroll
        for (int u = 0; u < EPL; ++u)
          if (mine(u)) kP[i][u] *= dt;
        km[i] *= dt;
      }
    }
    if constexpr (!KEEP) {
#pragma unroll
      for (int i = 0; i < 6; ++i)
        if (T::b[i] != R(0)) {
#pragma unroll
          for (int u = 0; u < EPL; ++u)
            if (mine(u)) P0[u] = rfma(T::b[i], kP[i][u], P0[u]);
          x0 = rfma(T::b[i], km[i], x0);
        }
    }
  };

#ifdef CDKF_W40A_PROFILE
  long long w40a_last = clock64();
#endif
  // The model block's two matrix leaves accumulate in REGISTERS over the sweep (round 5): dR = sum_k Sbar_k and dH = sum_k (2 Sbar (E P) -
  // vbar m^T + Ub P)_k are formed as accumulator tiles anyway, and the 102 accumulator registers this kernel leaves unused hold the
  // running sums of the tiles a wavefront owns (3 symmetric + 5 general tiles x 4 values) -- until round 4 every step read-modify-wrote
  // both matrices in global memory: 2 x (m^2 + m d) reals per trajectory-step, 52 KB at d = m = 40 in fp64, a third of the sweep's HBM
  // traffic (profiles/r04_o_config4_value_and_grad_2048x500_counters.json: 168 KB per trajectory-step against 26 KB algorithmic).
  V4 totS[NT], totH[NT];
  R totB = R(0);
  if (gm) {
    tilesS_zero(totS);
    tiles_zero(totH);
  }
  for (long k = a.T - 1; k >= 0; --k) {
    if constexpr (NW != 1) sync();  // (the other wavefront may still be reading the step's cotangent sums, which the images below overwrite)
    fresh();
    W40A_TICK(0)
    // =================================== (1) the measurement update at k, reversed =====================================================
    // predicted moments (P0 / m0 at k = 0) into I0, the identity into I2, Pbar into I1
    {
      const R* P0p = a.par + a.o_P0;
      const R* src = a.pP + n * a.P_sn + (k > 0 ? k - 1 : 0) * a.P_sk;
      constexpr int NE = (D * D + 63) / 64, CHK = (NE + 1) / 2;
#pragma unroll
      for (int q0 = 0; q0 < NE; q0 += CHK) {
        R v[CHK];
        if (k > 0) {
#pragma unroll
          for (int u = 0; u < CHK; ++u)
            if (mine(q0 + u)) {
              const int e = lane + 64 * (q0 + u);
              v[u] = src[(long)(e < D * D ? e : 0) * a.P_si];
            }
        } else {
#pragma unroll
          for (int u = 0; u < CHK; ++u)
            if (mine(q0 + u)) {
              const int e = lane + 64 * (q0 + u), ec = e < D * D ? e : 0, r = ec / D, c = ec - r * D;
              v[u] = R(0.5) * (P0p[r * D + c] + P0p[c * D + r]);
            }
        }
#pragma unroll
        for (int u = 0; u < CHK; ++u)  // (slots past the matrix rewrite entry (0, 0) with its own value: no predicated stores)
          if (mine(q0 + u)) {
            const int e = lane + 64 * (q0 + u), ec = e < D * D ? e : 0, r = ec / D, c = ec - r * D;
            I0[r * LDY + c] = v[u];
            I2[r * LDY + c] = (r == c) ? R(1) : R(0);
          }
      }
    }
    if constexpr (LEAD) {
      R mpred = R(0), vk = R(0);
      if (isrow) {
        mpred = (k == 0) ? (a.par + a.o_m0)[lane] : a.pm[n * a.m_sn + (k - 1) * a.m_sk + lane * a.m_si];
        vk = (myobs >= 0) ? yp[k * a.y_sk] - mpred : R(0);
      }
      v_v[lane] = vk;
      v_mb[lane] = mb;
      v_m[lane] = mpred;
    }
#pragma unroll
    for (int s = 0; s < EPL; ++s)
      if (mine(s) && W::owned(s, lane)) {
        const Off f = offsets(s);
        I1[f.y] = Pb[s];
        I1[f.yt] = Pb[s];
        if constexpr (kPark) PbG[64 * s + lane] = Pb[s];
      }
    sync();
    fresh();
    // sym(S) + 1e-9 I, packed lower (state coordinates: the identity on the unobserved components); the augmented row is not used
#pragma unroll
    for (int s = 0; s < EPL; ++s)
      if (mine(s) && W::owned(s, lane)) {
        const Off f = offsets(s);
        const unsigned wA = tabA[64 * s + lane];
        const bool obs_i = (wA >> 16) & 1u, obs_j = (wA >> 17) & 1u;
        const R sv = (obs_i && obs_j) ? I0[f.y] + r_of(s) : r_of(s);
        Lp[f.l] = (f.y == f.yt) ? sv + R(1e-9) : sv;
      }
    if (LEAD && isrow) Lp[W::rs(D) + lane] = R(0);
    sync();
    W40A_TICK(1)  // loads, images, packed system
    if constexpr (LEAD) {
      {
        R quad = R(0);
        double logdet = 0.0;
        R* const sys[1] = {Lp};
        R* const scr[1] = {v_ca};
        Lin::template cholesky<1>(sys, scr, v_inv, rowi, ri, lane, quad, logdet, bad);
      }
      W40A_TICK(2)  // factorisation
      // W2 = (sym S + 1e-9 I)^-1 = L^-T L^-1: the FORWARD substitution of the D columns of the identity, in place in I2 (row c of the
      // image = column c of Y = L^-1, which is lower triangular: the blocks' products skip the tiles of right-hand sides that are still
      // zero) -- the backward substitution is replaced by the product Y^T Y below, which both wavefronts share
      {
        constexpr int WL = D % 16 ? D % 16 : 16;
        R* myrow = I2 + (isrow ? lane : 0) * LDY;
        R dot = R(0);
        if constexpr (Lin::NB > 1) {
          Lin::template sub_block<true, 16>(myrow, Lp, v_inv, (const R*)nullptr, 0, isrow, dot);
          wave_sync();
          Lin::template solve_gemm<true, 1, 1>(I2, Lp, lane);
          wave_sync();
          if constexpr (Lin::NB > 2) {
            Lin::template sub_block<true, 16>(myrow, Lp, v_inv, (const R*)nullptr, 16, isrow, dot);
            wave_sync();
            Lin::template solve_gemm<true, 2, 3>(I2, Lp, lane);
            wave_sync();
            Lin::template sub_block<true, WL>(myrow, Lp, v_inv, (const R*)nullptr, 32, isrow, dot);
          } else {
            Lin::template sub_block<true, WL>(myrow, Lp, v_inv, (const R*)nullptr, 16, isrow, dot);
          }
        } else {
          Lin::template sub_block<true, WL>(myrow, Lp, v_inv, (const R*)nullptr, 0, isrow, dot);
        }
        wave_sync();
      }
    }
    // G = (E P) Pbar -> I3, ALL of it on one wavefront: with two per trajectory the second forms it while the first factors and
    // substitutes (nothing else is ready for it), and S X Pbar = G - 1e-9 X Pbar with X Pbar = W2 G needs neither X nor Pbar's image
    // any more -- the product (E P) Pbar leaves the shared part of the step
    if constexpr (NW == 1 || H == 1) {
      V4 accG[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) accG[t] = V4{0, 0, 0, 0};
      w40a_mmp<R, D, false, false, 0, 1, 0>(accG, rows_of(I0, true).p, cols_of(I1).p, blast_ok, obsmask, lg);
      tiles_store_all(I3, accG);
    }
    sync();
    {
      V4 acc[NT];  // (symmetric: the tiles on and above the diagonal, each stored with its mirror)
      tilesS_zero(acc);
      w40a_mmp<R, D, true, false, 0, NW, H, true>(acc, rows_of(I2, false).p, rows_of(I2, false).p, blast_ok, obsmask, lg);
      sync();  // (every read of Y is done)
      tilesS_store(I2, acc);
    }
    sync();
    W40A_TICK(3)  // W2
    // X = W2 (E P) -> I1 (Pbar's image is dead: G is formed)
    {
      V4 acc[NT];
      tiles_zero(acc);
      w40a_mmp<R, D, false, true, 0, NW, H>(acc, rows_of(I2, false).p, cols_of(I0).p, blast_ok, obsmask, lg);
      tiles_store(I1, acc);
    }
    sync();
    W40A_TICK(4)  // X
    // w = S^-1 v = u1 + 1e-9 W2 u1, u1 = W2 v;  vbar = X mbar - w
    R wv = R(0), vb = R(0);
    if constexpr (LEAD) {
      R u1 = R(0), tx = R(0);
      if (isrow) {
        const R* w2r = I2 + lane * LDY;
        const R* xr = I1 + lane * LDY;
#pragma unroll 4
        for (int c = 0; c < D; ++c) {
          u1 = rfma(w2r[c], v_v[c], u1);
          tx = rfma(xr[c], v_mb[c], tx);
        }
      }
      v_u[lane] = u1;
      wave_sync();
      R u2 = R(0);
      if (isrow) {
        const R* w2r = I2 + lane * LDY;
#pragma unroll 4
        for (int c = 0; c < D; ++c) u2 = rfma(w2r[c], v_u[c], u2);
      }
      wv = rfma(R(1e-9), u2, u1);
      vb = tx - wv;
      v_w[lane] = wv;
      v_vb[lane] = vb;
    }
    W40A_TICK(5)  // w, vbar
    // X Pb = W2 G;  Kb = v mbar^T - 2 S X Pb,  S X Pb = G - 1e-9 X Pb
    V4 accK[NT], accS[NT];
    {
      V4 accX[NT];
      tiles_zero(accX);
      w40a_mmp<R, D, false, false, 0, NW, H>(accX, rows_of(I2, false).p, cols_of(I3).p, blast_ok, obsmask, lg);
#pragma unroll
      for (int mt = 0; mt < NB; ++mt)
#pragma unroll
        for (int nt = 0; nt < NB; ++nt)
          if (mine(mt * NB + nt)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = 16 * mt + Tile::row(lg, r), col = 16 * nt + lm;
              const int t = mt * NB + nt;
              const R gel = (row < D && col < D) ? I3[row * LDY + col] : R(0);
              accK[t][r] = rfma(R(-2), rfma(R(-1e-9), accX[t][r], gel), v_v[row] * v_mb[col]);
            }
          }
      sync();  // (every read of G is done; v_w / v_vb are visible)
      tiles_store(I3, accX);  // X Pb
    }
    W40A_TICK(6)  // X Pb, (E P) Pb, Kb
    // Sbar = -(X Pb) X^T + w w^T / 2 - S^-1 / 2 - sym(X Ub^T);  S^-1 = W2 + 1e-9 W2 W2
    // (Sbar is symmetric, and so is every term of it: its tiles on and above the diagonal only -- w40a_mine_sym)
    tilesS_zero(accS);
    w40a_mmp<R, D, true, false, 0, NW, H, true>(accS, rows_of(I2, false).p, rows_of(I2, false).p, blast_ok, obsmask, lg);  // (W2 is symmetric)
#pragma unroll
    for (int mt = 0; mt < NB; ++mt)
#pragma unroll
      for (int nt = 0; nt < NB; ++nt)
        if (mineS(mt, nt)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * mt + Tile::row(lg, r), col = 16 * nt + lm;
            const int t = mt * NB + nt;
            const R w2 = (row < D && col < D) ? I2[row * LDY + col] : R(0);
            accS[t][r] = R(0.5) * (v_w[row] * v_w[col] - rfma(R(1e-9), accS[t][r], w2));
          }
        }
    sync();  // (X Pb is in I3)
    w40a_mmp<R, D, true, false, 1, NW, H, true>(accS, rows_of(I3, false).p, rows_of(I1, false).p, blast_ok, obsmask, lg);
    sync();
    tiles_store(I3, accK);  // Kb
    sync();
    W40A_TICK(7)  // W2 W2, (X Pb) X^T
    // Ub = W2 Kb -> I2 (over the then dead W2)
    {
      V4 accU[NT];
      tiles_zero(accU);
      w40a_mmp<R, D, false, false, 0, NW, H>(accU, rows_of(I2, false).p, cols_of(I3).p, blast_ok, obsmask, lg);
      sync();
      tiles_store(I2, accU);
    }
    sync();
    W40A_TICK(8)  // Ub
    w40a_mmp<R, D, true, false, 2, NW, H, true>(accS, rows_of(I1, false).p, rows_of(I2, false).p, blast_ok, obsmask, lg);
    w40a_mmp<R, D, true, false, 2, NW, H, true>(accS, rows_of(I2, false).p, rows_of(I1, false).p, blast_ok, obsmask, lg);
    // model block: dR += Sbar (the observed pairs)
    if (gm) {
#pragma unroll
      for (int mt = 0; mt < NB; ++mt)
#pragma unroll
        for (int nt = 0; nt < NB; ++nt)
          if (mineS(mt, nt)) totS[mt * NB + nt] += accS[mt * NB + nt];
    }
    tilesS_store(I3, accS);  // Sbar (Kb is dead: Ub was formed behind a synchronisation)
    sync();
    W40A_TICK(9)  // X Ub^T, Ub X^T, dR
    if (gm) {  // dH += 2 Sbar (E P) - vbar m^T + Ub P; dbias -= vbar
      V4 accH[NT];
      tiles_zero(accH);
      w40a_mmp<R, D, false, true, 3, NW, H>(accH, rows_of(I3, false).p, cols_of(I0).p, blast_ok, obsmask, lg);
      w40a_mmp<R, D, false, false, 0, NW, H>(accH, rows_of(I2, false).p, cols_of(I0).p, blast_ok, obsmask, lg);
#pragma unroll
      for (int mt = 0; mt < NB; ++mt)
#pragma unroll
        for (int nt = 0; nt < NB; ++nt)
          if (mine(mt * NB + nt)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = 16 * mt + Tile::row(lg, r), col = 16 * nt + lm;
              accH[mt * NB + nt][r] = rfma(-v_vb[row], v_m[col], accH[mt * NB + nt][r]);
            }
          }
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (mine(t)) totH[t] += accH[t];
      if (LEAD && myobs >= 0) totB -= vb;
    }
    fresh();
    // Pbar <- Pbar + sym(Ub^T H) + H^T Sbar H;  mbar <- mbar - H^T vbar
#pragma unroll
    for (int s = 0; s < EPL; ++s)
      if (mine(s) && W::owned(s, lane)) {
        const Off f = offsets(s);
        const unsigned wA = tabA[64 * s + lane];
        const bool obs_i = (wA >> 16) & 1u, obs_j = (wA >> 17) & 1u;
        const R uij = obs_j ? I2[f.yt] : R(0), uji = obs_i ? I2[f.y] : R(0);
        const R pb0 = kPark ? PbG[64 * s + lane] : Pb[s];
        Pb[s] = pb0 + rfma(R(0.5), uij + uji, (obs_i && obs_j) ? I3[f.y] : R(0));
      }
    mb -= vb;  // (zero on the second wavefront)
    sync();
    W40A_TICK(10)  // dH, Pbar
    if (k == 0) break;

    // =================================== (2) predict k-1 -> k: the Runge-Kutta steps of the interval, reversed =========================
    const R t0 = tp[(k - 1) * a.t_sk], t1 = tp[k * a.t_sk];
    long S = 0;
    {
      R tprev = t0, tnx = rmin(t0 + a.dt0, t1);
      while (tprev < t1 && S < a.max_steps) {
        ++S;
        tprev = rmin(tnx, t1);
        const R tn = tnx + a.dt0;
        tnx = (tn > t1 - Tol<R>::v) ? t1 : tn;
      }
      if (tprev < t1) st |= kStatusMaxSteps;
    }
    for (long cs = S > 0 ? ((S - 1) / cap) * cap : -1; cs >= 0; cs -= cap) {
      const long ce = (cs + cap < S) ? cs + cap : S;
      fresh();
      // replay from the filtered moments at k-1 up to the chunk's last step start, keeping the chunk's starts
      R P0[EPL], x0;
#pragma unroll
      for (int s = 0; s < EPL; ++s)
        if (mine(s)) {
          const Ent e = entry(s);
          P0[s] = a.fP[n * a.P_sn + (k - 1) * a.P_sk + (long)(e.i * D + e.j) * a.P_si];  // (slots past the triangle: entry (0, 0), masked below)
        }
      x0 = (LEAD && isrow) ? a.fm[n * a.m_sn + (k - 1) * a.m_sk + lane * a.m_si] : R(0);
#pragma unroll
      for (int s = 0; s < EPL; ++s)
        if (mine(s)) P0[s] = W::owned(s, lane) ? P0[s] : R(0);
      {
        R tprev = t0, tnx = rmin(t0 + a.dt0, t1);
        for (long s = 0; s < ce; ++s) {
          const R dt = tnx - tprev;
          if (s >= cs) {
            R* sv = wsb + (s - cs) * A::start_reals();  // (the covariance part is read back by the reverse pass's last stage as well)
#pragma unroll
            for (int u = 0; u < EPL; ++u)
              if (mine(u)) sv[64 * u + lane] = P0[u];
            if constexpr (LEAD) sv[64 * EPL + lane] = x0;
            v_dt[s - cs] = dt;  // (the step sizes of the chunk: cap <= 64)
          }
          if (s + 1 < ce) stages_forward(P0, x0, dt, std::false_type{});
          tprev = rmin(tnx, t1);
          const R tn = tnx + a.dt0;
          tnx = (tn > t1 - Tol<R>::v) ? t1 : tn;
        }
      }
      sync();
      for (long s = ce - 1; s >= cs; --s) {
        if (s + 1 < ce) {
          const R* sv = wsb + (s - cs) * A::start_reals();
#pragma unroll
          for (int u = 0; u < EPL; ++u)
            if (mine(u)) P0[u] = sv[64 * u + lane];
          if constexpr (LEAD) x0 = sv[64 * EPL + lane];
        }
        const R dt = v_dt[s - cs];
        W40A_TICK(11)  // replay, step start
        stages_forward(P0, x0, dt, std::true_type{});  // (the stage inputs parked for the reverse pass)
        W40A_TICK(12)  // stages forward
        // ---- ... and backward ----------------------------------------------------------------------------------------------------
        R Pn[EPL], mn = mb, accm[5];
#pragma unroll
        for (int u = 0; u < EPL; ++u)
          if (mine(u)) Pn[u] = Pb[u];
#pragma unroll
        for (int j = 0; j < 5; ++j) accm[j] = R(0);
#pragma unroll
        for (int i = 5; i >= 0; --i) {
          fresh();
          // stage cotangent  Lam = dt (b_i Pbar + sum_{j > i} a_ji Ybar_j)  and the stage's input, entry by entry into the two images
          // (neither is kept in registers: the stencil reads Lam back from its image)
          R lam, xs = x0;
          R pin_[EPL];  // the stage's input: in flight from the scratch while the cotangent is formed
#pragma unroll
          for (int u = 0; u < EPL; ++u)
            if (mine(u))
              pin_[u] = (i > 0) ? PsG[((i > 0 ? i - 1 : 0) * EPL + u) * 64 + lane] : (wsb + (s - cs) * A::start_reals())[64 * u + lane];
          unsigned wa[EPL];  // the owned entries' (i, j), kept for the stage (one batch of table reads)
#pragma unroll
          for (int u = 0; u < EPL; ++u)
            if (mine(u)) wa[u] = tabA[64 * u + lane];
          {
            constexpr int CP = NW == 1 ? 5 : 10;  // (slots in flight together: five of this wavefront's)
#pragma unroll
            for (int u0 = 0; u0 < EPL; u0 += CP) {
              R ac[CP];
#pragma unroll
              for (int c = 0; c < CP; ++c)
                if (u0 + c < EPL && mine(u0 + c)) {
                  const int u = u0 + c;
                  ac[c] = (i < 5) ? AccL[(i * EPL + u) * 64 + lane] : R(0);
                }
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int c = 0; c < CP; ++c)
                if (u0 + c < EPL && mine(u0 + c)) {
                  const int u = u0 + c;
                  const R Lm = dt * rfma(T::b[i], Pb[u], ac[c]);
                  const R Psu = pin_[u];
                  const int ei = (int)(wa[u] & 255u), ej = (int)((wa[u] >> 8) & 255u);
                  const bool own = W::owned(u, lane);  // (the last slot of the upper lanes: a scratch word instead of a predicated store)
                  R* l1 = own ? S0 + (ei + 2) * LDP + (ej + 2) : v_dummy + lane;
                  R* l2 = own ? S0 + (ej + 2) * LDP + (ei + 2) : v_dummy + lane;
                  R* q1 = own ? PsI + ei * LDY + ej : v_dummy + lane;
                  R* q2 = own ? PsI + ej * LDY + ei : v_dummy + lane;
                  *l1 = Lm;
                  *l2 = Lm;
                  *q1 = Psu;
                  *q2 = Psu;
                }
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          lam = dt * rfma(T::b[i], mb, (i < 5) ? accm[i < 5 ? i : 0] : R(0));
          if (LEAD && isrow) gF += lam;
          xs = xin[i];
          if (LEAD && isrow) {
            v_xs[lane] = xs;
            v_lam[lane] = lam;
          }
          W40A_TICK(18)  // (stage: cotangent, input, images)
          sync();
          if constexpr (NW == 1 || H == 1) {
            for (int e = lane; e < 3 * D; e += 64) {  // halo of Lam: rows / columns -1 <- D-1, D <- 0, D+1 <- 1
              const int r = (e >= 2 * D) ? 2 : (e >= D ? 1 : 0), c = e - r * D;
              const int src = (r == 0) ? D - 1 : r - 1, dst = (r == 0) ? -1 : D + r - 1;
              S0[(dst + 2) * LDP + (c + 2)] = S0[(src + 2) * LDP + (c + 2)];
              S0[(c + 2) * LDP + (dst + 2)] = S0[(c + 2) * LDP + (src + 2)];
            }
          }
          R xbar = R(0);
          if (LEAD && isrow) {  // coefficients of column j of the Jacobian: F[j-1][j] = x[j-2], F[j+2][j] = -x[j+1], F[j+1][j] = x[j+2] - x[j-1]
            const R xm2 = v_xs[lm2], xm1 = v_xs[lm1], xp1 = v_xs[lp1], xp2 = v_xs[lp2];
            v_c3[3 * lane] = xm2;
            v_c3[3 * lane + 1] = xp1;
            v_c3[3 * lane + 2] = xp2 - xm1;
            // F^T lam
            xbar = rfma(v_lam[lm1], xm2, rfma(-v_lam[lp2], xp1, rfma(v_lam[lp1], xp2 - xm1, -lam)));
          }
          sync();
          W40A_TICK(14)  // (stage: halo, coefficients)
          fresh();
          // Ybar_P = Lam F + (Lam F)^T of the owned entries, added to the step's input cotangent and to the earlier stages' sums as formed
          constexpr int CH = NW == 1 ? 4 : 8;
#pragma unroll
          for (int s0 = 0; s0 < EPL; s0 += CH) {
            R o7[CH][7], c6[CH][6], old[CH][5];
#pragma unroll
            for (int u = 0; u < CH; ++u)
              if (s0 + u < EPL && mine(s0 + u)) {
              const int s = s0 + u;
              const Ent e{(int)(wa[s] & 255u), (int)((wa[s] >> 8) & 255u)};
              const R* c = S0 + (e.i + 2) * LDP + (e.j + 2);
              o7[u][0] = c[-1];
              o7[u][1] = c[2];
              o7[u][2] = c[1];
              o7[u][3] = c[-LDP];
              o7[u][4] = c[2 * LDP];
              o7[u][5] = c[LDP];
              o7[u][6] = c[0];
#pragma unroll
              for (int q = 0; q < 3; ++q) {
                c6[u][q] = v_c3[3 * e.j + q];
                c6[u][3 + q] = v_c3[3 * e.i + q];
              }
#pragma unroll
              for (int j = 0; j < 5; ++j) old[u][j] = (j < i && i < 5) ? AccL[(j * EPL + s) * 64 + lane] : R(0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < CH; ++u)
              if (s0 + u < EPL && mine(s0 + u)) {
                const int s = s0 + u;
                R y = R(-2) * o7[u][6];
                y = rfma(c6[u][0], o7[u][0], y);
                y = rfma(-c6[u][1], o7[u][1], y);
                y = rfma(c6[u][2], o7[u][2], y);
                y = rfma(c6[u][3], o7[u][3], y);
                y = rfma(-c6[u][4], o7[u][4], y);
                y = rfma(c6[u][5], o7[u][5], y);
                y = W::owned(s, lane) ? y : R(0);
                Pn[s] += y;
#pragma unroll
                for (int j = 0; j < 5; ++j)
                  if (j < i) {
                    AccL[(j * EPL + s) * 64 + lane] = rfma(T::a[i][j], y, old[u][j]);
                  }
              }
            __builtin_amdgcn_sched_barrier(0);
          }
          W40A_TICK(15)  // (stage: stencil)
          // the state's cotangent: sum_ij (2 Lam Ps)_ij dF_ij/dx_k = 2 (M[k+1][k+2] - M[k+1][k-1] + M[k-1][k-2] - M[k+2][k+1]),  M = Lam Ps.
          // Lane i forms the three entries of ROW i of M that this needs from any lane -- b1 = M[i][i+1], b2 = M[i][i-2], b3 = M[i][i-1]:
          // one row of Lam against three rows of the (symmetric) stage covariance, 4 D reads per lane where three rows of Lam against
          // four of Ps for the four entries of lane k's own sum took 7 D; the LDS pipe, which the four wavefronts of a CU share, is what
          // bounds this phase -- and the lanes exchange the entries through vectors behind the stage's last synchronisation.
          // (NW = 2: each wavefront takes half of the columns; the partial entries of both meet in the same exchange)
          {
            R b1 = R(0), b2 = R(0), b3 = R(0);
            if (isrow) {
              const R* lr = S0 + (lane + 2) * LDP + 2;  // row i of Lam (odd leading dimension: one column per read)
              // (the stage covariance's rows start on 16-byte boundaries -- LDY is even: two columns per read)
              typedef R Pair __attribute__((ext_vector_type(2)));
              const Pair* q1 = reinterpret_cast<const Pair*>(__builtin_assume_aligned(PsI + lp1 * LDY, 16));
              const Pair* q2 = reinterpret_cast<const Pair*>(__builtin_assume_aligned(PsI + lm2 * LDY, 16));
              const Pair* q3 = reinterpret_cast<const Pair*>(__builtin_assume_aligned(PsI + lm1 * LDY, 16));
              R e1 = R(0), e2 = R(0), e3 = R(0);
              constexpr int c_lo = (NW == 1 || H == 0) ? 0 : D / 4, c_hi = (NW == 1 || H == 1) ? D / 2 : D / 4;  // (pairs of columns)
#pragma unroll 2
              for (int c = c_lo; c < c_hi; ++c) {
                const Pair a1 = q1[c], a2 = q2[c], a3 = q3[c];
                const R la = lr[2 * c], lb = lr[2 * c + 1];
                b1 = rfma(la, a1[0], b1);
                e1 = rfma(lb, a1[1], e1);
                b2 = rfma(la, a2[0], b2);
                e2 = rfma(lb, a2[1], e2);
                b3 = rfma(la, a3[0], b3);
                e3 = rfma(lb, a3[1], e3);
              }
              b1 += e1;
              b2 += e2;
              b3 += e3;
            }
            // (vectors of the update, free during the predict: v_v / v_w / v_vb the first wavefront's entries, v_u / v_m / v_inv the second's)
            (LEAD ? v_v : v_u)[lane] = b1;
            (LEAD ? v_w : v_m)[lane] = b2;
            (LEAD ? v_vb : v_inv)[lane] = b3;
          }
          W40A_TICK(16)  // (stage: dot products)
          sync();  // the images are rewritten by the next stage; the entries of M are visible
          if constexpr (LEAD) {  // the mean's part of the sums
            if (isrow) {
              R m12 = v_v[lp1] - v_w[lp1], m3 = v_vb[lm1] - v_vb[lp2];
              if constexpr (NW != 1) {
                m12 += v_u[lp1] - v_m[lp1];
                m3 += v_inv[lm1] - v_inv[lp2];
              }
              xbar = rfma(R(2), m12 + m3, xbar);
            }
#pragma unroll
            for (int j = 0; j < 5; ++j)
              if (j < i) accm[j] = rfma(T::a[i][j], xbar, accm[j]);
            mn += xbar;
          }
          W40A_TICK(17)  // (stage: sums)
        }
        // d ll / d (L Qc L^T) += the sum of the six stage cotangents = dt (Pbar + sum_j Acc_j)  (sum_i b_i = 1; the weighted sums are all
        // complete and still in their arrays)
#pragma unroll
        for (int u = 0; u < EPL; ++u)
          if (mine(u)) {
            R sacc = Pb[u];
#pragma unroll
            for (int j = 0; j < 5; ++j) sacc += AccL[(j * EPL + u) * 64 + lane];
            if constexpr (kPark) gQG[64 * u + lane] = rfma(dt, sacc, gQG[64 * u + lane]);
            else gQacc[u] = rfma(dt, sacc, gQacc[u]);
            Pb[u] = Pn[u];
          }
        mb = mn;
        W40A_TICK(13)  // stages reversed
      }
    }
  }

  fresh();
  // ---- results --------------------------------------------------------------------------------------------------------------------
  if (gm) {
    tilesS_accumulate(gR, totS, true, [&](int row, int col) {  // (written once: `first_` = plain stores)
      const int orow = obs[row], ocol = obs[col];
      return (orow >= 0 && ocol >= 0) ? orow * M + ocol : -1;
    });
    tiles_accumulate(gH, totH, true, [&](int row, int col) {
      const int orow = obs[row];
      return orow >= 0 ? orow * D + col : -1;
    });
    if (LEAD && myobs >= 0) gBias[myobs] = totB;
    if (LEAD && isrow) gm[lane] = mb;
#pragma unroll
    for (int s = 0; s < EPL; ++s)
      if (mine(s) && W::owned(s, lane)) {
        const Ent e = entry(s);
        gP0[e.i * D + e.j] = Pb[s];
        gP0[e.j * D + e.i] = Pb[s];
        const R gq = kPark ? gQG[64 * s + lane] : gQacc[s];
        gQ[e.i * D + e.j] = gq;
        gQ[e.j * D + e.i] = gq;
      }
  }
  if constexpr (LEAD) {
    wave_sync();
    v_u[lane] = isrow ? gF : R(0);
    wave_sync();
    if (lane == 0) {
      R sum = R(0);
      for (int c = 0; c < D; ++c) sum += v_u[c];
      g[0] = sum;
    }
    if (bad) st |= kStatusNotPd;
    if (st && lane == 0 && a.status) atomicOr(&a.status[n], st);
  }
#ifdef CDKF_W40A_PROFILE
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    printf("w40a cycles/obs-step (sizeof real %d, d %d):", (int)sizeof(R), D);
    for (int q2 = 0; q2 < 19; ++q2) printf(" [%d] %lld", q2, w40a_prof[q2] / a.T);
    printf("\n");
    for (int q2 = 0; q2 < 24; ++q2) w40a_prof[q2] = 0;
  }
#endif
}

// one wavefront per trajectory, two trajectories per workgroup
template <typename R, int D>
__global__ __launch_bounds__(128) void ekf_adjoint_wave_l96_kernel(const WgArgs<R> a, R* __restrict__ grad, R* __restrict__ grad_model,
                                                                    R* __restrict__ ws, const long ws_stride, const int cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  w40a_sweep<R, D, 1, 0>(a, grad, grad_model, ws, ws_stride, cap, smem_raw);
}
// two wavefronts per trajectory, one trajectory per workgroup (two workgroups per CU in fp64; fp32: the LDS admits more, and the
// instantiation is held to 256 registers -- two wavefronts per SIMD -- which it needs by a handful only)
template <typename R, int D>
__global__ __launch_bounds__(128, (sizeof(R) == 4 ? 2 : 1)) void ekf_adjoint_wave2_l96_kernel(const WgArgs<R> a, R* __restrict__ grad, R* __restrict__ grad_model,
                                                                     R* __restrict__ ws, const long ws_stride, const int cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (threadIdx.x < 64)
    w40a_sweep<R, D, 2, 0>(a, grad, grad_model, ws, ws_stride, cap, smem_raw);
  else
    w40a_sweep<R, D, 2, 1>(a, grad, grad_model, ws, ws_stride, cap, smem_raw);
}

}  // namespace cdkf
