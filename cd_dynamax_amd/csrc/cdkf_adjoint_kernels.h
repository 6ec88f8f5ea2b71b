// cdkf_adjoint_kernels.h -- reverse sweep (discrete adjoint) of the EKF log-likelihood, state_dim <= 8:
// d ll / d theta for ALL drift parameters (MLP: h1 d + h1 + h2 h1 + h2 + d h2 + d of them) and, on request, for every
// other parameter of the model (m0, P0, L Qc L^T, H, bias, R) in one backward pass.
//
// The reference gets this from jax.value_and_grad through the filter (ssm_temissions.py:550-568; reverse mode
// through diffrax with RecursiveCheckpointAdjoint, diffrax_utils.py:49).  Same quantity here, written out:
//   forward  : the wave8 filter sweep stores the predicted and filtered moments at every observation (these
//              are the adjoint's checkpoints -- the filter outputs the reference returns anyway);
//   backward : k = T-1 .. 0:  (1) adjoint of the measurement update + log-likelihood term at k,
//                             (2) adjoint of the Dormand-Prince steps over [t_{k-1}, t_k], re-integrating the interval
//                                 from the filtered moments at k-1 (step starts kept in LDS, stages recomputed),
//              each right-hand-side adjoint back-propagates through f AND through the Jacobian F = W3 D2 W2 D1 W1
//              (second-order backprop: reverse over the forward tangent pass).
// state_order 'second' (the reference's default): the MLP's mean term 0.5 P grad(div f) (inference_ekf.py:108-116) is evaluated
// and reversed too -- third derivatives of the drift, two more 64 x 64 x 9 products and one more rank-9 weight update per
// right-hand-side adjoint (mlp_second_fwd and the block after it in rhs_adj; oracle: divgrad_vjp).
//
// Mapping: one wavefront per trajectory as in cdkf_wave8_kernels.h -- lane (i, j) owns entry (i, j) of every 8 x 8 tile
// (P, its adjoint, the stage slopes and their adjoints stay in registers), lane p is hidden unit p in the MLP passes and
// accumulates row p of dW1, column p of dW3 and its bias entries for the whole sweep, dW2 lives in sixteen matrix-core
// accumulator tiles: the parameter gradient never leaves registers until the final store.  The three 64 x 64 x 9 products of a
// right-hand-side adjoint (tangent W2 [D1 W1 | a1], weight update [zt2 | z2b] [D1 W1 | a1]^T, transposed [zt2 | z2b]^T W2) run as
// v_mfma_*_16x16x4; W2 sits in LDS once with rows padded to 65, which serves it along p (A operand of the first) and along q (B
// operand of the last) without bank conflicts.
#pragma once
#include "cdkf_wave8_kernels.h"

namespace cdkf {

struct AdjSh {  // per workgroup, in reals; hidden sizes padded to 64, state to 8
  static constexpr int W1 = 0;             // [64][8]
  static constexpr int b1 = 512;           // [64]
  static constexpr int W2 = 576;           // [64][65]   W2[p][q], rows padded to 65: as a matrix-core operand it is read along p (A of
                                           //            the tangent product) AND along q (B of the transposed one), conflict-free both ways
  static constexpr int b2 = 576 + 64 * 65; // [64]
  static constexpr int W3 = b2 + 64;       // [8][65]
  static constexpr int b3 = W3 + 520;      // [8]
  static constexpr int end = b3 + 8;
};
struct AdjOff {  // per wavefront, in reals
  static constexpr int P = 0, F = 64, Lam = 128, G = 192, A = 256, B = 320, X = 384, S = 448, S1 = 512, S2 = 576, HP = 640,
                       H = 704, Pb = 768, Kb = 832, Ub = 896, Si = 960, XP = 1024;
  static constexpr int x = 1088, lam = 1096, f = 1104, v = 1112, w = 1120, mb = 1128, vb = 1136;
  // MLP passes: three [64][9] images (stride 9: a lane's row is conflict-free) that are the matrix-core operands / results --
  //   UC = [D1 W1 | a1]  (B of the tangent product, B of the dW2 update),   ZC = [zt2 | z2b]  (A of the dW2 update and of the
  //   transposed product),   TC = [T | z2 - b2] (result of the tangent product, a row per hidden unit); once its rows are in
  //   registers the same image takes CC = [c1 | s1] (result of the transposed product) and then the partial sums RED
  static constexpr int a1 = 1168, a2 = 1232, d2 = 1296, UC = 1360, ZC = 1936, TC = 2512, CC = 2512, RED = 2512;
  static constexpr int ck = 3088;  // kAdjCk x (64 P + 8 m), then kAdjCk step sizes
  static constexpr int km = 3088 + kAdjCk * 72 + kAdjCk, ym = km + 48;  // mean parts of the step's slopes / stage cotangents [6][8]
  // the Runge-Kutta tableau as the rolled stage loops want it: rka[r][c] (6 x 6, zero where the method has no entry: c >= r, or a
  // stage it does not have) and rkb[6] -- uniform LDS reads in ONE basic block instead of a scalar load and a branch per entry
  static constexpr int rka = ym + 48, rkb = rka + 36;
  static constexpr int end = rkb + 6;
};
template <typename R, bool MLP>
constexpr int adj_waves() {
  return 4;  // one wavefront per SIMD (150 KB of LDS per workgroup in fp64 with the MLP's weights)
}
template <typename R, bool MLP>
constexpr size_t adj_lds_bytes() {
  return sizeof(R) * (size_t)((MLP ? AdjSh::end : 0) + adj_waves<R, MLP>() * AdjOff::end) + 64;
}
// layout of the optional model-gradient block (per trajectory): m0 [d] | P0 [d,d] | LQL [d,d] | H [m,d] | bias [m] | R [m,m]
__host__ __device__ inline long adj_model_grad_size(int d, int m) { return (long)d + 2L * d * d + (long)m * d + m + (long)m * m; }

// SMOOTH: the same wavefront machinery runs the EKF (RTS) smoother's backward sweep instead of the adjoint (see below)
template <typename R, bool MLP, bool SMOOTH = false>
__global__ __launch_bounds__((64 * adj_waves<R, MLP>()), 1) void ekf_adjoint_wave8_kernel(const WgArgs<R> a, R* __restrict__ grad,
                                                                                       R* __restrict__ grad_model) {
  constexpr int WAVES = adj_waves<R, MLP>();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  R* Sh = reinterpret_cast<R*>(smem_raw);
  const int wave = threadIdx.x >> 6;
  int lane = threadIdx.x & 63;
  int i = lane >> 3, j = lane & 7;
  const int d = a.d, m = a.m, h1 = a.h1, h2 = a.h2;
  R* W = Sh + (MLP ? AdjSh::end : 0) + wave * AdjOff::end;
  const R* th = a.par + a.o_theta;
  const long oW1 = 0, ob1 = oW1 + (long)h1 * d, oW2 = ob1 + h1, ob2 = oW2 + (long)h2 * h1, oW3 = ob2 + h2, ob3 = oW3 + (long)d * h2;

  // ---- shared weights, zero-padded -------------------------------------------------------------------------------
  if constexpr (MLP) {
  for (int e = threadIdx.x; e < AdjSh::end; e += blockDim.x) Sh[e] = 0;
  __syncthreads();
  for (int e = threadIdx.x; e < h1 * d; e += blockDim.x) {
    const int q = fdiv(e, d);
    Sh[AdjSh::W1 + q * 8 + (e - q * d)] = th[oW1 + e];
  }
  for (int e = threadIdx.x; e < h1; e += blockDim.x) Sh[AdjSh::b1 + e] = th[ob1 + e];
  for (int e = threadIdx.x; e < h2 * h1; e += blockDim.x) {
    const int p = fdiv(e, h1), q = e - p * h1;
    Sh[AdjSh::W2 + p * 65 + q] = th[oW2 + e];
  }
  for (int e = threadIdx.x; e < h2; e += blockDim.x) Sh[AdjSh::b2 + e] = th[ob2 + e];
  for (int e = threadIdx.x; e < d * h2; e += blockDim.x) {
    const int r = fdiv(e, h2);
    Sh[AdjSh::W3 + r * 65 + (e - r * h2)] = th[oW3 + e];
  }
  for (int e = threadIdx.x; e < d; e += blockDim.x) Sh[AdjSh::b3 + e] = th[ob3 + e];
  __syncthreads();
  }
  const long n = (long)blockIdx.x * WAVES + wave;
  if (n >= a.N) return;  // whole wavefront; no workgroup barrier follows

  // ---- per-lane constants ---------------------------------------------------------------------------------------
  const bool inP = (i < d) && (j < d);
  const R lql = inP ? (a.par + a.o_LQL)[i * d + j] : R(0);
  if constexpr (!SMOOTH) {
    // tableau table (see AdjOff::rka); the slope / cotangent tiles start from zeros: the branch-free combinations below multiply
    // entries a stage does not use by a zero coefficient, which must not meet the NaN bit patterns uninitialised LDS may hold
    // (afterwards the tiles only ever hold this trajectory's own finite numbers: slopes, cotangents, the update's intermediates)
    if (lane < 36) {
      const int r = fdiv(lane, 6), c = lane - 6 * r;
      W[AdjOff::rka + lane] = (c < r && r < a.rk.stages && c < 5) ? a.rk.a[r][c] : R(0);
    }
    if (lane < 6) W[AdjOff::rkb + lane] = (lane < a.rk.stages) ? a.rk.b[lane] : R(0);
#pragma unroll
    for (int q = 0; q < 12; ++q) W[AdjOff::B + 64 * q + lane] = R(0);
    if (lane < 48) {
      W[AdjOff::km + lane] = R(0);
      W[AdjOff::ym + lane] = R(0);
    }
    wave_sync();
  }
  const R Hij = (i < m && j < d) ? (a.par + a.o_H)[i * d + j] : R(0);  // lane (r=i, k=j) holds H[r][k]
  const R Rij = (i < m && j < m) ? (a.par + a.o_R)[i * m + j] : R(0);
  const R hbj = (lane < m) ? (a.par + a.o_hb)[lane] : R(0);
  // this lane's row of W1 / column of W3 / biases come from the shared LDS copy where they are used: held in registers for the whole
  // sweep they were the first thing the allocator parked in scratch (30 reloads each per step)
  auto w1row = [&](int k) __attribute__((always_inline)) { return Sh[AdjSh::W1 + lane * 8 + k]; };
  auto w3col = [&](int k) __attribute__((always_inline)) { return Sh[AdjSh::W3 + k * 65 + lane]; };
  auto b1l = [&]() __attribute__((always_inline)) { return Sh[AdjSh::b1 + lane]; };
  auto b2l = [&]() __attribute__((always_inline)) { return Sh[AdjSh::b2 + lane]; };
  const R Wlin = (a.kind == kDriftLinear && inP) ? th[i * d + j] : R(0);  // linear drift: lane (i,k) holds W[i][k]
  const R blin = (a.kind == kDriftLinear && lane < d) ? th[d * d + lane] : R(0);

  // ---- parameter-gradient accumulators (registers for the whole sweep) -------------------------------------------
  // d ll / d W2 accumulates in matrix-core accumulator tiles for the whole sweep: gW2t[mt][nt][r] is entry
  // (16 mt + row(lg, r), 16 nt + lm) -- the rank-9 update of every right-hand-side adjoint is 48 v_mfma_*_16x16x4
  using MTile = W8Tile<R>;
  int lm = lane & 15, lg = lane >> 4;
  const R e8 = (lm == 8) ? R(1) : R(0);
  int sc_off = (lm == 8) ? AdjOff::a2 : AdjOff::d2;
  // The lane index is laundered through an empty asm at the top of every right-hand side: what is derived from it (LDS addresses of
  // the tiles and images) is then recomputed where it is used instead of being hoisted out of the sweep's loops and parked in scratch
  auto fresh = [&]() __attribute__((always_inline)) {
    asm volatile("" : "+v"(lane));
    i = lane >> 3;
    j = lane & 7;
    lm = lane & 15;
    lg = lane >> 4;
    sc_off = (lm == 8) ? AdjOff::a2 : AdjOff::d2;
  };
  typename MTile::V4 gW2t[MLP ? 4 : 1][MLP ? 4 : 1];
  R gW1[8], gW3[8];
  R gb1 = 0, gb2 = 0, gb3 = 0;
#pragma unroll
  for (int mt = 0; mt < (MLP ? 4 : 1); ++mt)
#pragma unroll
    for (int nt = 0; nt < (MLP ? 4 : 1); ++nt) gW2t[mt][nt] = typename MTile::V4{0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 8; ++k) gW1[k] = gW3[k] = 0;
  R gTile = 0, gVec = 0;                 // non-MLP drifts: linear dW[i][j] on lane (i, j); bias / Lorenz parameters on lanes < 8
  R gQ = 0, gR = 0, gH = 0, gBias = 0;   // model block: d/d(L Qc L^T), d/dR, d/dH on lane (i, j); d/d bias on lanes < 8

  // ---- 8 x 8 tile products: lane (i, j) gets one entry ----------------------------------------------------------
  auto mm = [&](int TA, int TB) __attribute__((always_inline)) {  // (A B)_ij
    R s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s = rfma(W[TA + i * 8 + k], W[TB + k * 8 + j], s);
    return s;
  };
  auto mm_tn = [&](int TA, int TB) __attribute__((always_inline)) {  // (A^T B)_ij
    R s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s = rfma(W[TA + k * 8 + i], W[TB + k * 8 + j], s);
    return s;
  };
  auto mm_nt = [&](int TA, int TB) __attribute__((always_inline)) {  // (A B^T)_ij
    R s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s = rfma(W[TA + i * 8 + k], W[TB + j * 8 + k], s);
    return s;
  };

  // ---- MLP forward with the tangent of the 8 unit directions; needs W[x] = stage mean (synced) ---------------------
  // lane = hidden unit: returns a1, d1 (layer 1), a2, d2, T = W2 (D1 W1) row (layer 2); leaves a1, U = D1 W1, a2,
  // V = D2 T in LDS and the Jacobian entry F_ij / the drift f in registers of lane (i, j) / LDS vector f.
  auto mlp_fwd = [&](R& a1, R& d1, R& a2, R& d2, R (&T)[8], R& Fij) __attribute__((always_inline)) {
    R z1 = b1l();
#pragma unroll
    for (int k = 0; k < 8; ++k) z1 = rfma(w1row(k), W[AdjOff::x + k], z1);
    a1 = rtanh_fast(z1);
    d1 = R(1) - a1 * a1;
    W[AdjOff::a1 + lane] = a1;
#pragma unroll
    for (int k = 0; k < 8; ++k) W[AdjOff::UC + lane * 9 + k] = d1 * w1row(k);
    W[AdjOff::UC + lane * 9 + 8] = a1;
    wave_sync();
    // [T | z2 - b2] = W2 [D1 W1 | a1] on the matrix cores (A operand: W2T rows, lanes along p; B operand: UC rows)
    typename MTile::V4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = typename MTile::V4{0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const R bv = (lm < 9) ? W[AdjOff::UC + (4 * ks + lg) * 9 + lm] : R(0);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt] = wg_mfma(Sh[AdjSh::W2 + (16 * mt + lm) * 65 + 4 * ks + lg], bv, acc[mt]);
      if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // operands of four k-steps in flight, not of all sixteen (registers)
    }
    if (lm < 9) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) W[AdjOff::TC + (16 * mt + MTile::row(lg, r)) * 9 + lm] = acc[mt][r];
    }
    wave_sync();
#pragma unroll
    for (int k = 0; k < 8; ++k) T[k] = W[AdjOff::TC + lane * 9 + k];
    a2 = rtanh_fast(W[AdjOff::TC + lane * 9 + 8] + b2l());
    d2 = R(1) - a2 * a2;
    W[AdjOff::a2 + lane] = a2;
    W[AdjOff::d2 + lane] = d2;
    wave_sync();
    // [F | f - b3] = W3 [D2 T | a2]: the accumulator rows of the first product are this one's k index
    typename MTile::V4 acc3{0, 0, 0, 0};
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // (the tangent comes back from its image rather than from the live accumulators: registers are what this kernel is
        //  short of -- every value kept across the tanh evaluations above was a spill)
        const int pr = 16 * mt + MTile::row(lg, r);
        const R sc = W[sc_off + pr];
        const R w3 = (lm < 8) ? Sh[AdjSh::W3 + lm * 65 + pr] : R(0);
        const R tv = (lm < 8) ? W[AdjOff::TC + pr * 9 + lm] : e8;
        acc3 = wg_mfma(w3, sc * tv, acc3);
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = MTile::row(lg, r);
      if (row < 8 && lm < 8) W[AdjOff::F + row * 8 + lm] = acc3[r];
      if (row < 8 && lm == 8) W[AdjOff::f + row] = (row < d) ? acc3[r] + Sh[AdjSh::b3 + row] : R(0);
    }
    wave_sync();
    Fij = inP ? W[AdjOff::F + lane] : R(0);
    wave_sync();  // (the callers overwrite the F tile)
  };

  // ---- state_order 'second' for the MLP: the mean also moves with 0.5 P g,  g = grad(div f)  (inference_ekf.py:108-116) -------------
  // With M = (W1 W3)^T and G = M * W2 (oracle/cdkf_oracle.py MLPDrift.divgrad):  g = W1^T tq,  tq = d1 (-2 a1 td + tc),
  //   td_q = sum_p d2_p G_pq = sum_i W1[q][i] E1[q][i],   E1 = W2^T A1,  A1 = diag(d2) W3^T           (G never formed: rank 8 through W3, W1)
  //   tc   = W2^T s2,   s2 = -2 a2 d2 s,   s_p = sum_q G_pq d1_q = sum_i W3[i][p] T[p][i]               (T: the tangent mlp_fwd leaves)
  // so [E1 | tc] = W2^T [A1 | s2] is ONE more 64 x 64 x 9 product on the matrix cores per right-hand side.
  const bool second = MLP && a.order == 2;
  // OUT[p][c] = sum_q W2[p][q] IN[q][c]  (images [64][9]; the caller synchronises before; OUT may be IN)
  // (between: runs after the product's operands have been read and before its result overwrites OUT -- work that still needs IN)
  auto w2_times = [&](int IN, int OUT, auto&& between) __attribute__((always_inline)) {
    typename MTile::V4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = typename MTile::V4{0, 0, 0, 0};
    {  // software pipeline: the operands of k-step ks + LA are requested before the four products of k-step ks are issued (the fence the
       // products' B operand passes through keeps the loads above it and the products below).  LA: one k-step in fp64 (four products =
       // 256 cycles cover an LDS round trip), four in fp32 (128 cycles do not: scripts/mb/mb_mfma16.hip)
      constexpr int LA = sizeof(R) == 4 ? 4 : 1;
      R bq[16], aq[16][4];
      auto request = [&](int ks) __attribute__((always_inline)) {
        bq[ks] = (lm < 9) ? W[IN + (4 * ks + lg) * 9 + lm] : R(0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) aq[ks][mt] = Sh[AdjSh::W2 + (16 * mt + lm) * 65 + 4 * ks + lg];
      };
#pragma unroll
      for (int ks = 0; ks < LA; ++ks) request(ks);
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        R bv = bq[ks];
        const R a0 = aq[ks][0], a1_ = aq[ks][1], a2_ = aq[ks][2], a3_ = aq[ks][3];
        if (ks + LA < 16) request(ks + LA);
        asm volatile("" : "+v"(bv) : : "memory");
        acc[0] = wg_mfma(a0, bv, acc[0]);
        acc[1] = wg_mfma(a1_, bv, acc[1]);
        acc[2] = wg_mfma(a2_, bv, acc[2]);
        acc[3] = wg_mfma(a3_, bv, acc[3]);
      }
    }
    between();
    wave_sync();
    if (lm < 9) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) W[OUT + (16 * mt + MTile::row(lg, r)) * 9 + lm] = acc[mt][r];
    }
    wave_sync();
  };
  // cacc[nt][r] = sum_p IN[p][row(lg, r)] W2[p][16 nt + lm]  (software-pipelined like w2_times)
  auto w2t_product = [&](int IN, typename MTile::V4 (&cacc)[4]) __attribute__((always_inline)) {
    constexpr int LA = sizeof(R) == 4 ? 4 : 1;  // (look-ahead in k-steps, as in w2_times)
    R aq[16], bq[16][4];
    auto request = [&](int ks) __attribute__((always_inline)) {
      aq[ks] = (lm < 9) ? W[IN + (4 * ks + lg) * 9 + lm] : R(0);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) bq[ks][nt] = Sh[AdjSh::W2 + (4 * ks + lg) * 65 + 16 * nt + lm];
    };
#pragma unroll
    for (int ks = 0; ks < LA; ++ks) request(ks);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      R av = aq[ks];
      const R b0 = bq[ks][0], b1_ = bq[ks][1], b2_ = bq[ks][2], b3_ = bq[ks][3];
      if (ks + LA < 16) request(ks + LA);
      asm volatile("" : "+v"(av) : : "memory");
      cacc[0] = wg_mfma(av, b0, cacc[0]);
      cacc[1] = wg_mfma(av, b1_, cacc[1]);
      cacc[2] = wg_mfma(av, b2_, cacc[2]);
      cacc[3] = wg_mfma(av, b3_, cacc[3]);
    }
  };
  // OUT[q][c] = sum_p IN[p][c] W2[p][q]
  auto w2t_times = [&](int IN, int OUT) __attribute__((always_inline)) {
    typename MTile::V4 cacc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) cacc[nt] = typename MTile::V4{0, 0, 0, 0};
    w2t_product(IN, cacc);
    wave_sync();
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = MTile::row(lg, r);
        if (row < 9) W[OUT + (16 * nt + lm) * 9 + row] = cacc[nt][r];
      }
    wave_sync();
  };
  // dW2 += A B^T for two [64][9] images (three k-steps into the sixteen accumulator tiles)
  auto dw2_update = [&](int AI, int BI) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      const int jj = 4 * ks + lg;
      const bool jin = jj < 9;
      R av[4], bv[4];
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
        av[t4] = jin ? W[AI + (16 * t4 + lm) * 9 + jj] : R(0);
        bv[t4] = jin ? W[BI + (16 * t4 + lm) * 9 + jj] : R(0);
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) gW2t[mt][nt] = wg_mfma(av[mt], bv[nt], gW2t[mt][nt]);
    }
  };
  // g into the LDS vector AdjOff::v (lanes < 8); leaves [A1 | s2] in the ZC image; returns what the reverse pass needs
  auto mlp_second_fwd = [&](R a1, R d1, R a2, R d2, const R (&T)[8], R (&E1)[8], R& sdiv, R& s2, R& td, R& tq) __attribute__((always_inline)) {
    sdiv = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) sdiv = rfma(w3col(k), T[k], sdiv);
    s2 = R(-2) * a2 * d2 * sdiv;
#pragma unroll
    for (int k = 0; k < 8; ++k) W[AdjOff::ZC + lane * 9 + k] = d2 * w3col(k);
    W[AdjOff::ZC + lane * 9 + 8] = s2;
    wave_sync();
    w2t_times(AdjOff::ZC, AdjOff::CC);
#pragma unroll
    for (int k = 0; k < 8; ++k) E1[k] = W[AdjOff::CC + lane * 9 + k];
    const R tc = W[AdjOff::CC + lane * 9 + 8];
    wave_sync();  // (RED below shares the CC image)
    td = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) td = rfma(w1row(k), E1[k], td);
    tq = d1 * rfma(R(-2) * a1, td, tc);
#pragma unroll
    for (int k = 0; k < 8; ++k) W[AdjOff::RED + lane * 8 + k] = w1row(k) * tq;
    wave_sync();
    R part = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) part += W[AdjOff::RED + (8 * c + i) * 8 + j];
    W[AdjOff::A + lane] = part;
    wave_sync();
    if (lane < 8) {
      R sg = 0;
#pragma unroll
      for (int r = 0; r < 8; ++r) sg += W[AdjOff::A + r * 8 + lane];
      W[AdjOff::v + lane] = (lane < d) ? sg : R(0);
    }
    wave_sync();
  };

  // ---- registry drifts other than the MLP: Jacobian entry on lane (i, j), f in the LDS vector; needs W[x] (synced) ----
  auto drift_fwd = [&](R& Fij) __attribute__((always_inline)) {
    R xk[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) xk[k] = W[AdjOff::x + k];
    Fij = 0;
    R fi = 0;
    if (a.kind == kDriftLinear) {
      Fij = Wlin;
      if (lane < 8) {
        fi = blin;
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (k < d) fi = rfma(th[lane < d ? lane * d + k : 0], xk[k], fi);
      }
    } else if (a.kind == kDriftLorenz63) {
      const R sg = th[0], rho = th[1], bt = th[2];
      if (i == 0) Fij = (j == 0) ? -sg : (j == 1 ? sg : R(0));
      if (i == 1) Fij = (j == 0) ? rho - xk[2] : (j == 1 ? R(-1) : (j == 2 ? -xk[0] : R(0)));
      if (i == 2) Fij = (j == 0) ? xk[1] : (j == 1 ? xk[0] : (j == 2 ? -bt : R(0)));
      if (lane == 0) fi = sg * (xk[1] - xk[0]);
      if (lane == 1) fi = xk[0] * (rho - xk[2]) - xk[1];
      if (lane == 2) fi = xk[0] * xk[1] - bt * xk[2];
    } else {  // Lorenz-96
      auto X = [&](int q) __attribute__((always_inline)) { return W[AdjOff::x + q]; };
      const int ip1 = (i + 1 >= d) ? 0 : i + 1, im1 = (i == 0) ? d - 1 : i - 1, im2 = (im1 == 0) ? d - 1 : im1 - 1;
      if (inP) {
        if (j == ip1) Fij = X(im1);
        if (j == im2) Fij = -X(im1);
        if (j == im1) Fij = X(ip1) - X(im2);
        if (j == i) Fij = R(-1);
      }
      if (lane < d) {
        const int l = lane;
        const int lp1 = (l + 1 >= d) ? 0 : l + 1, lm1 = (l == 0) ? d - 1 : l - 1, lm2 = (lm1 == 0) ? d - 1 : lm1 - 1;
        fi = rfma(X(lp1) - X(lm2), X(lm1), th[0] - X(l));
      }
    }
    if (!inP) Fij = 0;
    if (lane < 8) W[AdjOff::f + lane] = (lane < d) ? fi : R(0);
  };

  W8_TICK_DECL
  // ---- right-hand side of the moment ODEs (state_order 'first') --------------------------------------------------
  auto rhs_fwd = [&](R xs, R Ps, R& kM, R& kP) __attribute__((always_inline)) {
    fresh();
    W[AdjOff::P + lane] = Ps;
    if (lane < 8) W[AdjOff::x + lane] = xs;
    wave_sync();
    R Fij;
    if constexpr (MLP) {
      R a1, d1, a2, d2, T[8];
      mlp_fwd(a1, d1, a2, d2, T, Fij);
      if (second) {
        R E1[8], sdiv, s2, td, tq;
        mlp_second_fwd(a1, d1, a2, d2, T, E1, sdiv, s2, td, tq);
      }
    } else {
      drift_fwd(Fij);
    }
    W[AdjOff::F + lane] = Fij;
    wave_sync();
    const R acc = mm(AdjOff::F, AdjOff::P);
    W[AdjOff::A + lane] = acc;
    wave_sync();
    kP = (acc + W[AdjOff::A + j * 8 + i]) + lql;
    if (lane < 8) {
      kM = W[AdjOff::f + lane];
      if constexpr (!MLP) {
        if (a.ukf && lane < d) {  // the unscented filter's curvature term of a quadratic drift (cdkf_wave8_kernels.h, oracle: ukf_curvature)
          auto Pe = [&](int r, int c) __attribute__((always_inline)) { return W[AdjOff::P + r * 8 + c]; };
          if (a.kind == kDriftLorenz63) {
            if (lane == 1) kM -= Pe(0, 2);
            if (lane == 2) kM += Pe(0, 1);
          } else if (a.kind == kDriftLorenz96) {
            const int l = lane;
            const int lp1 = (l + 1 >= d) ? 0 : l + 1, lm1 = (l == 0) ? d - 1 : l - 1, lm2 = (lm1 == 0) ? d - 1 : lm1 - 1;
            kM += Pe(lp1, lm1) - Pe(lm2, lm1);
          }
        }
      }
      if (second) {  // + 0.5 (P g)_lane
        R hg = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) hg = rfma(W[AdjOff::P + lane * 8 + k], W[AdjOff::v + k], hg);
        kM = rfma(R(0.5), hg, kM);
      }
    }
    wave_sync();
  };

  // ---- its adjoint: given the cotangent (lam, Lam) of the slope at the stage value (xs, Ps) -----------------------
  //   Ybar_P = F^T Lam + Lam F;   Ybar_m, dtheta += gradient of  lam . f(x) + <G, F(x)>,  G = 2 Lam P
  int pf_acc = 0, pf_pending = 0;  // fp32: L2 warm-up load of the next stage's checkpoint (see rhs_adj)
  // (mck: the forward sweep's MLP checkpoint of this stage, or nullptr -- then the forward pass through the network is repeated here)
  auto rhs_adj = [&](R xs, R Ps, R lam, R Lam, R& YM, R& YP, const R* mck) __attribute__((always_inline)) {
    fresh();
    W8_TICK(10)  // between right-hand-side adjoints: stage values, cotangent combinations
    // The forward sweep's checkpoint of this stage (MLP): every load is issued HERE, before anything else -- the tile traffic and the
    // products that do not need them run while they are in flight.  fp32 adds two loads that only pull the stage BELOW this one in
    // memory (stages and intervals are walked downwards: the next one wanted) into L2; their values are consumed at the next call, when
    // they have long arrived (vector-memory loads return in order: issued ahead of this stage's they would hold them back -- and in
    // the fp64 instantiation, which still parks registers in scratch, they hold back every scratch reload behind them: measured
    // 25 k against 21 k cycles per observation step there, 15 k against 23 k in fp32; holding the next stage in 22 registers instead
    // was slower than the L2 warm-up as well: 19 k).
    R ck_a1 = 0, ck_a2 = 0, ck_T[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ck_F = 0, ck_s = 0, ck_td = 0, ck_tq = 0, ck_E1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ck_g = 0;
    if constexpr (MLP) {
      if constexpr (sizeof(R) == 4) {
        pf_acc ^= pf_pending;
        pf_pending = 0;
      }
      if (mck) {
        ck_a1 = mck[kMlpCkA1 * 64 + lane];
        ck_a2 = mck[kMlpCkA2 * 64 + lane];
#pragma unroll
        for (int k = 0; k < 8; ++k) ck_T[k] = mck[(kMlpCkT + k) * 64 + lane];
        ck_F = mck[kMlpCkF * 64 + lane];
        if (second) {
          ck_s = mck[kMlpCkS * 64 + lane];
          ck_td = mck[kMlpCkTd * 64 + lane];
          ck_tq = mck[kMlpCkTq * 64 + lane];
#pragma unroll
          for (int k = 0; k < 8; ++k) ck_E1[k] = mck[(kMlpCkE1 + k) * 64 + lane];
          if (lane < 8) ck_g = mck[kMlpCkG * 64 + lane];
        }
        if constexpr (sizeof(R) == 4) {
          if (mck > a.ckm) {  // (the first stage of the buffer has nothing below it)
            const int* below = reinterpret_cast<const int*>(mck - (long)a.ckm_nf * 64);
            pf_pending = below[lane * 32];  // 64 lines of 128 B: a whole fp32 stage (5.9 KB)
          }
        }
      }
    }
    W[AdjOff::P + lane] = Ps;
    W[AdjOff::Lam + lane] = Lam;
    if (lane < 8) {
      W[AdjOff::x + lane] = xs;
      W[AdjOff::lam + lane] = lam;
    }
    wave_sync();
    gQ += Lam;
    if constexpr (!MLP) {
      R Fij;
      drift_fwd(Fij);
      W[AdjOff::F + lane] = Fij;
      const R Gij = R(2) * mm(AdjOff::Lam, AdjOff::P);
      W[AdjOff::G + lane] = Gij;
      wave_sync();
      YP = mm_tn(AdjOff::F, AdjOff::Lam) + mm(AdjOff::Lam, AdjOff::F);
      if (a.ukf && inP) {  // cotangent of Ps through lam . curvature(Ps), symmetric (oracle: ukf_curvature_vjp)
        auto lm_ = [&](int r) __attribute__((always_inline)) { return W[AdjOff::lam + r]; };
        if (a.kind == kDriftLorenz63) {
          if ((i == 0 && j == 2) || (i == 2 && j == 0)) YP -= R(0.5) * lm_(1);
          if ((i == 0 && j == 1) || (i == 1 && j == 0)) YP += R(0.5) * lm_(2);
        } else if (a.kind == kDriftLorenz96) {
          auto wr = [&](int q) __attribute__((always_inline)) { return q < 0 ? q + d : (q >= d ? q - d : q); };
          // Pb[l+1][l-1] += lam_l, Pb[l-2][l-1] -= lam_l, then 0.5 (Pb + Pb^T): the entries of lane (i, j)
          if (j == wr(i - 2)) YP += R(0.5) * lm_(wr(i - 1));   // (i, j) = (l+1, l-1)
          if (i == wr(j - 2)) YP += R(0.5) * lm_(wr(j - 1));   // its transpose
          if (j == wr(i + 1)) YP -= R(0.5) * lm_(wr(i + 2));   // (i, j) = (l-2, l-1)
          if (i == wr(j + 1)) YP -= R(0.5) * lm_(wr(j + 2));   // its transpose
        }
      }
      auto G = [&](int r, int c) __attribute__((always_inline)) { return W[AdjOff::G + r * 8 + c]; };
      if (a.kind == kDriftLinear) gTile = rfma(W[AdjOff::lam + i], W[AdjOff::x + j], gTile + Gij);
      if (lane < 8) {
        R s = 0;  // (F^T lam)_lane
#pragma unroll
        for (int r = 0; r < 8; ++r) s = rfma(W[AdjOff::F + r * 8 + lane], W[AdjOff::lam + r], s);
        if (a.kind == kDriftLinear) {
          gVec += lam;
        } else if (a.kind == kDriftLorenz63) {
          const R x0 = W[AdjOff::x + 0], x1 = W[AdjOff::x + 1], x2 = W[AdjOff::x + 2];
          if (lane == 0) {
            gVec += lam * (x1 - x0) - G(0, 0) + G(0, 1);
            s += -G(1, 2) + G(2, 1);
          } else if (lane == 1) {
            gVec += lam * x0 + G(1, 0);
            s += G(2, 0);
          } else if (lane == 2) {
            gVec += -lam * x2 - G(2, 2);
            s += -G(1, 0);
          }
        } else {  // Lorenz-96: dF[i][i+1]/dx_{i-1} = 1, dF[i][i-2]/dx_{i-1} = -1, dF[i][i-1]/dx_{i+1} = 1, dF[i][i-1]/dx_{i-2} = -1
          if (lane < d) {
            auto wrap = [&](int q) __attribute__((always_inline)) { return q < 0 ? q + d : (q >= d ? q - d : q); };
            const int c = lane;
            const int c1 = wrap(c + 1), c2 = wrap(c + 2), cm1 = wrap(c - 1), cm2 = wrap(c - 2);
            s += G(c1, c2) - G(c1, cm1) + G(cm1, cm2) - G(c2, c1);
          }
          if (lane == 0) {
            R sl = 0;
#pragma unroll
            for (int r = 0; r < 8; ++r) sl += W[AdjOff::lam + r];
            gVec += sl;
          }
        }
        YM = (lane < d) ? s : R(0);
      }
      wave_sync();
      return;
    } else {
    R a1, d1, a2, d2, T[8], Fij;
    if (mck) {  // (uniform) what the forward sweep computed at this very stage value, from HBM: lane p = hidden unit p
      a1 = ck_a1;
      a2 = ck_a2;
#pragma unroll
      for (int k = 0; k < 8; ++k) T[k] = ck_T[k];
      Fij = ck_F;
      d1 = R(1) - a1 * a1;
      d2 = R(1) - a2 * a2;
      // the image the weight update reads: UC = [D1 W1 | a1]  (mlp_fwd leaves it as the tangent product's operand)
#pragma unroll
      for (int k = 0; k < 8; ++k) W[AdjOff::UC + lane * 9 + k] = d1 * w1row(k);
      W[AdjOff::UC + lane * 9 + 8] = a1;
    } else {
      mlp_fwd(a1, d1, a2, d2, T, Fij);
    }
    W8_TICK(11)  // forward pass through the network (or its checkpoint)
    // reverse of the 'second'-order mean term 0.5 P g (oracle/cdkf_oracle.py divgrad_vjp, same names): with u = 0.5 P lam,
    //   r = W1 u,  td_b = -2 a1 d1 r,  tc_b = d1 r,  [F1 | s2_b] = W2 [diag(td_b) W1 | tc_b]  (cotangents of A1 and s2),
    //   d2_b = sum_i W3[i][p] F1[p][i],  s_b = -2 a2 d2 s2_b;  A2 = diag(s_b) W3^T is a cotangent of the tangent T and simply joins zt2,
    //   so the first-order pipeline below carries it to dW2, dW1, db1, the state; what is new on the matrix cores is E1, F1 and
    //   the rank-9 update dW2 += [A1 | s2] [diag(td_b) W1 | tc_b]^T.
    R sb_ = 0, z2x = 0, z1x = 0;
    if (second) {
      R E1[8], sdiv, s2, td, tq;
      if (mck) {
        sdiv = ck_s;
        td = ck_td;
        tq = ck_tq;
#pragma unroll
        for (int k = 0; k < 8; ++k) E1[k] = ck_E1[k];
        s2 = R(-2) * a2 * d2 * sdiv;
        // ZC = [A1 | s2] (operand of the rank-9 weight update below) and g, as mlp_second_fwd leaves them
#pragma unroll
        for (int k = 0; k < 8; ++k) W[AdjOff::ZC + lane * 9 + k] = d2 * w3col(k);
        W[AdjOff::ZC + lane * 9 + 8] = s2;
        if (lane < 8) W[AdjOff::v + lane] = ck_g;
        wave_sync();
      } else {
        mlp_second_fwd(a1, d1, a2, d2, T, E1, sdiv, s2, td, tq);
      }
      W8_TICK(12)  // 'second': E1, tc, g (or their checkpoint)
      if (lane < 8) {
        R hu = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) hu = rfma(W[AdjOff::P + lane * 8 + k], W[AdjOff::lam + k], hu);
        W[AdjOff::w + lane] = R(0.5) * hu;
      }
      wave_sync();
      R rr = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) rr = rfma(w1row(k), W[AdjOff::w + k], rr);
      const R td_b = R(-2) * a1 * d1 * rr, tc_b = d1 * rr;
      z1x = rr * (R(-2) * a1 * tq - R(2) * d1 * d1 * td);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        gW1[k] = rfma(tq, W[AdjOff::w + k], rfma(td_b, E1[k], gW1[k]));
        W[AdjOff::TC + lane * 9 + k] = td_b * w1row(k);
      }
      W[AdjOff::TC + lane * 9 + 8] = tc_b;
      wave_sync();
      W8_TICK(13)  // 'second': u = 0.5 P lam, r, B1 image
      // F1 = W2 [B1 | tc_b] in place; the rank-9 update dW2 += [A1 | s2] [B1 | tc_b]^T reads the same image before it is overwritten
      w2_times(AdjOff::TC, AdjOff::TC, [&]() __attribute__((always_inline)) { dw2_update(AdjOff::ZC, AdjOff::TC); });
      R d2_b = 0;
      const R s2_b = W[AdjOff::TC + lane * 9 + 8];
      sb_ = R(-2) * a2 * d2 * s2_b;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const R f1 = W[AdjOff::TC + lane * 9 + k];
        d2_b = rfma(w3col(k), f1, d2_b);
        gW3[k] = rfma(d2, f1, rfma(sb_, T[k], gW3[k]));
      }
      z2x = s2_b * (R(-2) * sdiv) * d2 * (R(1) - R(3) * a2 * a2) + d2_b * (R(-2) * a2 * d2);
      wave_sync();  // (ZC is rewritten below)
      W8_TICK(14)  // 'second': F1 product (64 MFMA) + rank-9 weight update (48 MFMA) + d2_b
    }
    W[AdjOff::F + lane] = Fij;
    W[AdjOff::G + lane] = R(2) * mm(AdjOff::Lam, AdjOff::P);
    wave_sync();
    YP = mm_tn(AdjOff::F, AdjOff::Lam) + mm(AdjOff::Lam, AdjOff::F);
    if (second) YP += R(0.25) * (W[AdjOff::lam + i] * W[AdjOff::v + j] + W[AdjOff::v + i] * W[AdjOff::lam + j]);
    // layer 3, lane p: c2_j = sum_i W3[i][p] G[i][j] is the cotangent of V[p][j]
    R c2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) c2[k] = 0;
    R a2b = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      R gv = 0;  // sum_j G[r][j] V[p][j]
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const R g = W[AdjOff::G + r * 8 + k];
        c2[k] = rfma(w3col(r), g, c2[k]);
        gv = rfma(g, T[k], gv);
      }
      const R lamr = W[AdjOff::lam + r];
      gW3[r] = rfma(lamr, a2, rfma(d2, gv, gW3[r]));
      a2b = rfma(w3col(r), lamr, a2b);
    }
    R zt2[8];
    {
      R s = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) s = rfma(T[k], c2[k], s);
      a2b = rfma(R(-2) * a2, s, a2b);
#pragma unroll
      for (int k = 0; k < 8; ++k) zt2[k] = rfma(d2, c2[k], sb_ * w3col(k));
    }
    const R z2b = rfma(d2, a2b, z2x);
    gb2 += z2b;
#pragma unroll
    for (int k = 0; k < 8; ++k) W[AdjOff::ZC + lane * 9 + k] = zt2[k];
    W[AdjOff::ZC + lane * 9 + 8] = z2b;
    wave_sync();
    W8_TICK(15)  // G = 2 Lam P, Ybar_P, layer 3 reversed (c2, dW3), zt2 image
    // layer 2 weights: dW2 += [zt2 | z2b] [D1 W1 | a1]^T -- a 64 x 64 x 9 product accumulated in the tiles (three k-steps)
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      const int jj = 4 * ks + lg;
      const bool jin = jj < 9;
      R av[4], bv[4];
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
        av[t4] = jin ? W[AdjOff::ZC + (16 * t4 + lm) * 9 + jj] : R(0);
        bv[t4] = jin ? W[AdjOff::UC + (16 * t4 + lm) * 9 + jj] : R(0);
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) gW2t[mt][nt] = wg_mfma(av[mt], bv[nt], gW2t[mt][nt]);
    }
    W8_TICK(16)  // weight update dW2 (48 MFMA)
    // layer 1: [c1 | s1]^T = [zt2 | z2b]^T W2 (cotangents of U and a1): rows j = 0 .. 8, columns q
    {
      typename MTile::V4 cacc[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) cacc[nt] = typename MTile::V4{0, 0, 0, 0};
      w2t_product(AdjOff::ZC, cacc);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = MTile::row(lg, r);
          if (row < 9) W[AdjOff::CC + (16 * nt + lm) * 9 + row] = cacc[nt][r];
        }
    }
    wave_sync();
    W8_TICK(17)  // transposed product (64 MFMA) + its image
    R c1[8], s1 = W[AdjOff::CC + lane * 9 + 8];
#pragma unroll
    for (int k = 0; k < 8; ++k) c1[k] = W[AdjOff::CC + lane * 9 + k];
    wave_sync();  // (RED below shares the CC image)
    {
      R s = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) s = rfma(w1row(k), c1[k], s);
      s1 = rfma(R(-2) * a1, s, s1);
    }
    const R z1b = rfma(d1, s1, z1x);
    gb1 += z1b;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      gW1[k] = rfma(z1b, W[AdjOff::x + k], rfma(d1, c1[k], gW1[k]));
      W[AdjOff::RED + lane * 8 + k] = w1row(k) * z1b;
    }
    if (lane < 8) gb3 += lam;
    wave_sync();
    // Ybar_m[j] = sum_q W1[q][j] z1b[q]: lane (i, j) sums the hidden units q = 8 c + i, the partial sums meet in a tile
    R part = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) part += W[AdjOff::RED + (8 * c + i) * 8 + j];
    W[AdjOff::A + lane] = part;
    wave_sync();
    if (lane < 8) {
      R s = 0;
#pragma unroll
      for (int r = 0; r < 8; ++r) s += W[AdjOff::A + r * 8 + lane];
      YM = s;
    }
    wave_sync();
    W8_TICK(18)  // layer 1 reversed: z1b, dW1, Ybar_m reduction
    }
  };

  using TB = Dp5T<R>;
  // stage value i from the step start and the slopes k_0 .. k_{i-1}
  auto stage_in = [&](int si, R y0, const R (&ks)[6], R dt) __attribute__((always_inline)) {
    R s = 0;
#pragma unroll
    for (int jj = 0; jj < 5; ++jj)
      if (jj < si) s = rfma(TB::a[si][jj], ks[jj], s);
    return rfma(dt, s, y0);
  };
  // The slopes of the step in hand and the cotangents of its stage values live in LDS, in the twelve tiles that only the measurement
  // update's adjoint uses (B .. XP) plus two small vectors: the stage loops stay ROLLED -- one inlined copy of each right-hand side
  // instead of six (the unrolled sweep was ~30 k instructions, several times the instruction cache, with ~800 registers parked in
  // scratch), and the stage index may be a run-time value.  A lane reads back only what it wrote itself: no synchronisation.
  constexpr int KPo = AdjOff::B, YPo = AdjOff::B + 6 * 64, KMo = AdjOff::km, YMo = AdjOff::ym;
  static_assert(AdjOff::B + 12 * 64 == AdjOff::x, "twelve contiguous tiles B .. XP");
  const int l8 = lane & 7;
  // (the tableau of the step is the one the forward sweep used: a.rk, opts.solver -- fixed steps; nst stages)
  const int nst = SMOOTH ? 6 : a.rk.stages;
  auto stage_val = [&](int sg, R mj, R Pij, R dt, R& xm, R& Pst) __attribute__((always_inline)) {
    R sm_ = 0, sp_ = 0;
#pragma unroll
    for (int jj = 0; jj < 5; ++jj) {  // (entries jj >= sg carry a zero coefficient: same sums as the guarded form)
      const R c = W[AdjOff::rka + 6 * sg + jj];
      sp_ = rfma(c, W[KPo + 64 * jj + lane], sp_);
      sm_ = rfma(c, W[KMo + 8 * jj + l8], sm_);
    }
    xm = rfma(dt, sm_, mj);
    Pst = rfma(dt, sp_, Pij);
  };
  auto stages_fwd = [&](R mj, R Pij, R dt) __attribute__((always_inline)) {
#pragma unroll 1
    for (int sg = 0; sg < nst; ++sg) {
      R xm, Pst, kM = 0, kP = 0;
      stage_val(sg, mj, Pij, dt, xm, Pst);
      rhs_fwd(xm, Pst, kM, kP);
      W[KPo + 64 * sg + lane] = kP;
      if (lane < 8) W[KMo + 8 * sg + lane] = kM;
    }
  };
  // y + dt sum_s b_s k_s from the slopes in the tiles
  auto step_end = [&](R& mj, R& Pij, R dt) __attribute__((always_inline)) {
    R sm_ = 0, sp_ = 0;
#pragma unroll
    for (int sg = 0; sg < 6; ++sg) {  // (a stage the method does not have: zero weight)
      const R c = W[AdjOff::rkb + sg];
      sm_ = rfma(c, W[KMo + 8 * sg + l8], sm_);
      sp_ = rfma(c, W[KPo + 64 * sg + lane], sp_);
    }
    mj = rfma(dt, sm_, mj);
    Pij = rfma(dt, sp_, Pij);
  };
  // one Runge-Kutta step forward (as the filter takes it)
  auto step_fwd = [&](R& mj, R& Pij, R dt) __attribute__((always_inline)) {
    stages_fwd(mj, Pij, dt);
    step_end(mj, Pij, dt);
  };
  // ... and its adjoint: (mb, Pb) cotangent of the step's result -> cotangent of its start; dtheta accumulated
  // (slopes: the step's slopes are already in the tiles -- read back from the forward sweep's checkpoints)
  auto step_adj = [&](R mj, R Pij, R dt, R& mb, R& Pb, bool slopes, const R* mckp) __attribute__((always_inline)) {
    if (!slopes) stages_fwd(mj, Pij, dt);
#pragma unroll 1
    for (int sg = nst - 1; sg >= 0; --sg) {
      const R bs = W[AdjOff::rkb + sg];
      R lm_ = bs * mb, lp = bs * Pb;
#pragma unroll
      for (int r = 5; r > 0; --r) {  // (r <= sg or a stage the method does not have: zero coefficient)
        const R c = W[AdjOff::rka + 6 * r + sg];
        lm_ = rfma(c, W[YMo + 8 * r + l8], lm_);
        lp = rfma(c, W[YPo + 64 * r + lane], lp);
      }
      R xm, Pst, yM = 0, yP = 0;
      stage_val(sg, mj, Pij, dt, xm, Pst);
      rhs_adj(xm, Pst, dt * lm_, dt * lp, yM, yP, mckp ? mckp + (long)sg * a.ckm_nf * 64 : nullptr);
      W[YPo + 64 * sg + lane] = yP;
      if (lane < 8) W[YMo + 8 * sg + lane] = yM;
    }
#pragma unroll
    for (int sg = 0; sg < 6; ++sg)
      if (sg < nst) {
        if (lane < 8) mb += W[YMo + 8 * sg + lane];
        Pb += W[YPo + 64 * sg + lane];
      }
  };

  // ---- SMOOTH: EKF (RTS) smoother backward sweep for state_dim <= 8 (inference_ekf.py:363-448, 503-531) -------------------
  // Per interval the filtered (m_f, P_f) are constants: G = F(m_f) + psd_solve(P_f, L Qc L^T)^T and f(m_f) are formed once,
  // then  dm = -[f(m_f) + G (m_s - m_f)],  dP = -[G P_s + (G P_s)^T - L Qc L^T]  is integrated over [0, t_{k+1} - t_k].
  if constexpr (SMOOTH) {
    const R* tp = a.t + n * a.t_sn;
    auto mo = [&](long k) __attribute__((always_inline)) { return n * a.m_sn + k * a.m_sk + lane * a.m_si; };
    auto po = [&](long k) __attribute__((always_inline)) { return n * a.P_sn + k * a.P_sk + (long)(i * d + j) * a.P_si; };
    R lqlcol[8];  // column j of L Qc L^T, the right-hand side of this lane's column solve
#pragma unroll
    for (int r = 0; r < 8; ++r) lqlcol[r] = (r < d && j < d) ? (a.par + a.o_LQL)[r * d + j] : R(0);
    R ms = (lane < d) ? a.fm[mo(a.T - 1)] : R(0);
    R Ps = inP ? a.fP[po(a.T - 1)] : R(0);
    if (lane < d) a.sm[mo(a.T - 1)] = ms;
    if (inP) a.sP[po(a.T - 1)] = Ps;
    int st = 0;
    R t1 = tp[(a.T - 1) * a.t_sk];
    for (long k = a.T - 2; k >= 0; --k) {
      const R t0 = tp[k * a.t_sk];
      const R mf = (lane < d) ? a.fm[mo(k)] : R(0);
      const R Pf = inP ? a.fP[po(k)] : R(0);
      // F(m_f), f(m_f)
      if (lane < 8) W[AdjOff::x + lane] = mf;
      W[AdjOff::P + lane] = Pf;
      wave_sync();
      R Fij;
      if constexpr (MLP) {
        R a1, d1, a2, d2, Tt[8];
        mlp_fwd(a1, d1, a2, d2, Tt, Fij);
      } else {
        drift_fwd(Fij);
      }
      wave_sync();
      const R fmf = (lane < 8) ? W[AdjOff::f + lane] : R(0);
      // psd_solve(P_f, LQL): Sb = sym(P_f) + 1e-9 I, padded with the identity
      R s2 = inP ? R(0.5) * (Pf + W[AdjOff::P + j * 8 + i]) + (i == j ? R(1e-9) : R(0)) : (i == j ? R(1) : R(0));
      R inv2[8];
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        W[AdjOff::S2 + lane] = s2;
        wave_sync();
        const R p2 = W[AdjOff::S2 + p * 8 + p];
        if (p < d && !(p2 > R(0))) st |= kStatusNotPd;
        const R r2 = rrsqrt(p2);
        inv2[p] = r2;
        const R l2i = W[AdjOff::S2 + i * 8 + p] * r2, l2j = W[AdjOff::S2 + j * 8 + p] * r2;
        wave_sync();
        if (j == p && i >= p)
          s2 = (i == p) ? p2 * r2 : l2i;
        else if (i > p && j > p && j <= i)
          s2 = rfma(-l2i, l2j, s2);
      }
      W[AdjOff::S2 + lane] = s2;
      wave_sync();
      R col[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) col[r] = lqlcol[r];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        R w = col[r];
#pragma unroll
        for (int c = 0; c < r; ++c) w = rfma(-W[AdjOff::S2 + r * 8 + c], col[c], w);
        col[r] = w * inv2[r];
      }
#pragma unroll
      for (int r = 7; r >= 0; --r) {
        R w = col[r];
#pragma unroll
        for (int c = r + 1; c < 8; ++c) w = rfma(-W[AdjOff::S2 + c * 8 + r], col[c], w);
        col[r] = w * inv2[r];
      }
      R xij = 0;  // X[i][j] = (Sb^-1 LQL)[i][j]
#pragma unroll
      for (int r = 0; r < 8; ++r)
        if (r == i) xij = col[r];
      W[AdjOff::X + lane] = inP ? xij : R(0);
      wave_sync();
      W[AdjOff::G + lane] = Fij + W[AdjOff::X + j * 8 + i];  // G = F + X^T
      if (lane < 8) W[AdjOff::mb + lane] = mf;
      wave_sync();
      // reverse-time right-hand side at the stage value
      auto rhs_s = [&](R xs, R Pst, R& kM, R& kP) __attribute__((always_inline)) {
        W[AdjOff::P + lane] = Pst;
        if (lane < 8) W[AdjOff::x + lane] = xs;
        wave_sync();
        const R acc = mm(AdjOff::G, AdjOff::P);
        W[AdjOff::A + lane] = acc;
        if (lane < 8) {
          R sdot = 0;
#pragma unroll
          for (int q = 0; q < 8; ++q) sdot = rfma(W[AdjOff::G + lane * 8 + q], W[AdjOff::x + q] - W[AdjOff::mb + q], sdot);
          kM = (lane < d) ? -(fmf + sdot) : R(0);
        }
        wave_sync();
        kP = inP ? -((acc + W[AdjOff::A + j * 8 + i]) - lql) : R(0);
        wave_sync();
      };
      const R tend = t1 - t0;
      R tprev = 0, tnext = rmin(a.dt0, tend);
      long steps = 0;
      while (tprev < tend) {  // uniform over the wavefront
        if (steps >= a.max_steps) {
          st |= kStatusMaxSteps;
          break;
        }
        const R dt = tnext - tprev;
        R kM[6] = {0, 0, 0, 0, 0, 0}, kP[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int sg = 0; sg < 6; ++sg) rhs_s(stage_in(sg, ms, kM, dt), stage_in(sg, Ps, kP, dt), kM[sg], kP[sg]);
        R sm_ = 0, sp_ = 0;
#pragma unroll
        for (int sg = 0; sg < 6; ++sg) {
          sm_ = rfma(TB::b[sg], kM[sg], sm_);
          sp_ = rfma(TB::b[sg], kP[sg], sp_);
        }
        ms = rfma(dt, sm_, ms);
        Ps = rfma(dt, sp_, Ps);
        tprev = rmin(tnext, tend);
        const R tn = tnext + a.dt0;
        tnext = (tn > tend - Tol<R>::v) ? tend : tn;
        ++steps;
      }
      if (lane < d) a.sm[mo(k)] = ms;
      if (inP) a.sP[po(k)] = Ps;
      t1 = t0;
    }
    if (st && lane == 0 && a.status) atomicOr(&a.status[n], st);
    return;
  }

  // ---- backward sweep ------------------------------------------------------------------------------------------------
  const R* tp = a.t + n * a.t_sn;
  const R* yp = a.y + n * a.y_sn;
  R mb = 0, Pb = 0;  // cotangent of the filtered moments at k (mean on lanes < 8)
  int adj_st = 0;
  for (long k = a.T - 1; k >= 0; --k) {
    W8_TICK(19)  // interval bookkeeping: slope loads, step starts, chunk loops
    // (1) measurement update + log-likelihood term at k, from the predicted moments
    R mp, Pp;
    if (k == 0) {
      mp = (lane < d) ? (a.par + a.o_m0)[lane] : R(0);
      Pp = inP ? R(0.5) * ((a.par + a.o_P0)[i * d + j] + (a.par + a.o_P0)[j * d + i]) : R(0);
    } else {
      mp = (lane < d) ? a.pm[n * a.m_sn + (k - 1) * a.m_sk + lane * a.m_si] : R(0);
      Pp = inP ? a.pP[n * a.P_sn + (k - 1) * a.P_sk + (i * d + j) * a.P_si] : R(0);
    }
    const R yl = (lane < m) ? yp[k * a.y_sk + lane * a.y_si] : R(0);
    // num_iter > 1 (inference_ekf.py:153-199: every iteration starts from the previous one's posterior, symmetrize once at the end, the
    // log-likelihood term on the first one's inputs): the inputs of iterations 1 .. n-1 are recomputed forward from the predicted
    // moments -- the forward sweep's own update on a scratch block (the MLP passes' images, free here) -- and parked in LDS (72 reals
    // each); the iterations are then reversed one by one, the last first.
    const int nit = a.num_iter;
    R* const itW = W + AdjOff::UC;            // a W8Off-shaped block of 608 reals for w8_measurement_update
    R* const itS = itW + W8Off::base_end;     // stash: iteration it's inputs at itS + 72 it (P on the lane grid, the mean behind)
    if (nit > 1) {
      R Pi = Pp, mi = mp;
      double ll_unused = 0.0;
      bool bad_unused = false;
      for (int it = 1; it < nit; ++it) {
        w8_measurement_update<R, false>(itW, lane, i, j, d, m, inP, false, Hij, Rij, hbj, yl, 1, 0, Pi, mi, ll_unused, bad_unused);
        itS[72 * it + lane] = Pi;
        if (lane < 8) itS[72 * it + 64 + lane] = mi;
      }
      wave_sync();
    }
    R vb_sum = 0;
#pragma unroll 1
    for (int it = nit - 1; it >= 0; --it) {
    const bool first = it == 0, last = it == nit - 1;
    if (it > 0) {
      Pp = itS[72 * it + lane];
      mp = (lane < 8) ? itS[72 * it + 64 + lane] : R(0);
    } else if (nit > 1) {  // (back to the predicted moments)
      if (k == 0) {
        mp = (lane < d) ? (a.par + a.o_m0)[lane] : R(0);
        Pp = inP ? R(0.5) * ((a.par + a.o_P0)[i * d + j] + (a.par + a.o_P0)[j * d + i]) : R(0);
      } else {
        mp = (lane < d) ? a.pm[n * a.m_sn + (k - 1) * a.m_sk + lane * a.m_si] : R(0);
        Pp = inP ? a.pP[n * a.P_sn + (k - 1) * a.P_sk + (i * d + j) * a.P_si] : R(0);
      }
    }
    W[AdjOff::P + lane] = Pp;
    W[AdjOff::H + lane] = Hij;
    W[AdjOff::Pb + lane] = Pb;
    if (lane < 8) {
      W[AdjOff::x + lane] = mp;
      W[AdjOff::mb + lane] = mb;
    }
    wave_sync();
    const R hp = mm(AdjOff::H, AdjOff::P);
    const R Pbs = last ? R(0.5) * (Pb + W[AdjOff::Pb + j * 8 + i]) : Pb;  // (the forward symmetrises once, after the last iteration)
    W[AdjOff::HP + lane] = hp;
    R vv = 0;
    if (lane < 8) {
      R hm = 0;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) hm = rfma(W[AdjOff::H + lane * 8 + kk], W[AdjOff::x + kk], hm);
      vv = (lane < m) ? yl - (hm + hbj) : R(0);
      W[AdjOff::v + lane] = vv;
    }
    wave_sync();
    const R s = (i < m && j < m) ? mm_nt(AdjOff::HP, AdjOff::H) + Rij : R(0);
    W[AdjOff::S + lane] = s;
    W[AdjOff::Pb + lane] = Pbs;
    wave_sync();
    // factorise S (log-likelihood) and Sb = sym(S) + 1e-9 I (psd_solve), both padded with the identity
    R s1 = (i < m && j < m) ? s : (i == j ? R(1) : R(0));
    R s2 = (i < m && j < m) ? R(0.5) * (s + W[AdjOff::S + j * 8 + i]) + (i == j ? R(1e-9) : R(0)) : (i == j ? R(1) : R(0));
    R inv1[8], inv2[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      if (p >= m) {  // (uniform) identity padding: pivot 1, nothing below it
        inv1[p] = inv2[p] = R(1);
        continue;
      }
      W[AdjOff::S1 + lane] = s1;
      W[AdjOff::S2 + lane] = s2;
      wave_sync();
      const R p1 = W[AdjOff::S1 + p * 8 + p], p2 = W[AdjOff::S2 + p * 8 + p];
      const R r1 = rrsqrt(p1), r2 = rrsqrt(p2);
      inv1[p] = r1;
      inv2[p] = r2;
      const R l1i = W[AdjOff::S1 + i * 8 + p] * r1, l1j = W[AdjOff::S1 + j * 8 + p] * r1;
      const R l2i = W[AdjOff::S2 + i * 8 + p] * r2, l2j = W[AdjOff::S2 + j * 8 + p] * r2;
      wave_sync();
      if (j == p && i >= p) {
        s1 = (i == p) ? p1 * r1 : l1i;
        s2 = (i == p) ? p2 * r2 : l2i;
      } else if (i > p && j > p && j <= i) {
        s1 = rfma(-l1i, l1j, s1);
        s2 = rfma(-l2i, l2j, s2);
      }
    }
    W[AdjOff::S1 + lane] = s1;
    W[AdjOff::S2 + lane] = s2;
    wave_sync();
    // column solves with a factor tile: col <- (L L^T)^-1 col
    auto chol_solve_col = [&](int TL, const R (&inv)[8], R (&col)[8]) __attribute__((always_inline)) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        R w = col[r];
#pragma unroll
        for (int c = 0; c < r; ++c) w = rfma(-W[TL + r * 8 + c], col[c], w);
        col[r] = w * inv[r];
      }
#pragma unroll
      for (int r = 7; r >= 0; --r) {
        R w = col[r];
#pragma unroll
        for (int c = r + 1; c < 8; ++c) w = rfma(-W[TL + c * 8 + r], col[c], w);
        col[r] = w * inv[r];
      }
    };
    auto pick = [&](const R (&col)[8]) __attribute__((always_inline)) {  // col[i] without a run-time register index
      R out = 0;
#pragma unroll
      for (int r = 0; r < 8; ++r)
        if (r == i) out = col[r];
      return out;
    };
    R col[8];
    // w = S^-1 v (every lane, redundantly)
    R wv[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) wv[r] = W[AdjOff::v + r];
    chol_solve_col(AdjOff::S1, inv1, wv);
    // S^-1 (column j), masked to the m x m block
#pragma unroll
    for (int r = 0; r < 8; ++r) col[r] = (r == j) ? R(1) : R(0);
    chol_solve_col(AdjOff::S1, inv1, col);
    const R sinv = (i < m && j < m) ? pick(col) : R(0);
    // X = Sb^-1 (H P) (column j)
#pragma unroll
    for (int r = 0; r < 8; ++r) col[r] = W[AdjOff::HP + r * 8 + j];
    chol_solve_col(AdjOff::S2, inv2, col);
    const R xij = (i < m) ? pick(col) : R(0);
    W[AdjOff::X + lane] = xij;
    wave_sync();
    // XP = X Pbar;  vbar = X mbar - w
    const R xp = mm(AdjOff::X, AdjOff::Pb);
    W[AdjOff::XP + lane] = xp;
    R vb = 0;
    if (lane < 8) {
      R sacc = 0;
#pragma unroll
      for (int c = 0; c < 8; ++c) sacc = rfma(W[AdjOff::X + lane * 8 + c], W[AdjOff::mb + c], sacc);
      R wl = 0;
#pragma unroll
      for (int r = 0; r < 8; ++r)
        if (r == lane) wl = wv[r];
      vb = (lane < m) ? (first ? sacc - wl : sacc) : R(0);  // (- w: the log-likelihood term, on the first iteration's inputs only)
      W[AdjOff::vb + lane] = vb;
      vb_sum += vb;
    }
    wave_sync();
    // Kb = v mbar^T - 2 S (X Pbar)      (cotangent of K^T)
    const R kb = rfma(R(-2), mm(AdjOff::S, AdjOff::XP), W[AdjOff::v + i] * W[AdjOff::mb + j]);
    W[AdjOff::Kb + lane] = (i < m) ? kb : R(0);
    wave_sync();
    // Ub = Sb^-1 Kb (column j)
#pragma unroll
    for (int r = 0; r < 8; ++r) col[r] = W[AdjOff::Kb + r * 8 + j];
    chol_solve_col(AdjOff::S2, inv2, col);
    W[AdjOff::Ub + lane] = (i < m) ? pick(col) : R(0);
    wave_sync();
    // Sbar = -(X Pbar) X^T + w w^T / 2 - S^-1 / 2 - sym(X Ub^T)
    R wi = 0, wj = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (r == i) wi = wv[r];
      if (r == j) wj = wv[r];
    }
    const R xu = mm_nt(AdjOff::X, AdjOff::Ub);
    W[AdjOff::A + lane] = xu;
    wave_sync();
    R sbar = -mm_nt(AdjOff::XP, AdjOff::X) - R(0.5) * (xu + W[AdjOff::A + j * 8 + i]);
    if (first) sbar += R(0.5) * wi * wj - R(0.5) * sinv;
    if (!(i < m && j < m)) sbar = 0;
    W[AdjOff::B + lane] = sbar;
    wave_sync();
    // model block: dR += Sbar; dH += 2 Sbar (H P) - vbar m^T + Ub P; dbias -= vbar
    gR += sbar;
    gH += R(2) * mm(AdjOff::B, AdjOff::HP) - W[AdjOff::vb + i] * W[AdjOff::x + j] + mm(AdjOff::Ub, AdjOff::P);
    if (lane < 8) gBias -= vb;
    // Pbar <- Pbar + sym(Ub^T H) + H^T Sbar H;   mbar <- mbar - H^T vbar
    const R uh = mm_tn(AdjOff::Ub, AdjOff::H);
    const R sh = mm(AdjOff::B, AdjOff::H);
    wave_sync();
    W[AdjOff::A + lane] = uh;
    W[AdjOff::B + lane] = sh;
    wave_sync();
    Pb = Pbs + R(0.5) * (uh + W[AdjOff::A + j * 8 + i]) + mm_tn(AdjOff::H, AdjOff::B);
    if (!inP) Pb = 0;
    if (lane < 8) {
      R sacc = 0;
#pragma unroll
      for (int r = 0; r < 8; ++r) sacc = rfma(W[AdjOff::H + r * 8 + lane], W[AdjOff::vb + r], sacc);
      mb = (lane < d) ? mb - sacc : R(0);
    }
    wave_sync();
    }  // (update iterations)
    // (per-step cotangents for the linear front-end's offsets, WgArgs::gcj / gy: mb is now the cotangent of the mean PREDICTED for t_k,
    //  i.e. of the jump added at the end of interval k-1; the vb of the iterations add up to that of y_k)
    if (a.gy && lane < m) a.gy[(n * a.T + k) * m + lane] = vb_sum;
    if (a.gcj && lane < d) {
      if (k > 0) a.gcj[(n * a.T + (k - 1)) * d + lane] = mb;
      if (k == a.T - 1) a.gcj[(n * a.T + k) * d + lane] = R(0);  // (the jump behind the last observation reaches no likelihood term)
    }
    W8_TICK(20)  // measurement update reversed
    if (k == 0) break;

    // (2) predict k-1 -> k: reverse the Dormand-Prince steps, replaying the interval in chunks of kAdjCk steps
    const R t0 = tp[(k - 1) * a.t_sk], t1 = tp[k * a.t_sk];
    // an adaptive solve: the forward sweep logged the step sizes it accepted in this interval (the reverse of the solve treats them
    // as constants -- the controller's factor carries no derivative, as in the reference's reverse mode through diffrax)
    const R* dtl = a.dtlog ? a.dtlog + (n * (a.T - 1) + (k - 1)) * (1 + a.dtlog_cap) : nullptr;
    long S = 0;
    if (dtl) {
      S = (long)dtl[0];
      if (S > a.dtlog_cap) {  // more accepted steps than the log holds: the gradient of this trajectory is not valid
        S = a.dtlog_cap;
        adj_st |= kStatusMaxSteps;
      }
    } else {
      R tprev = t0, tnext = rmin(t0 + a.dt0, t1);
      while (tprev < t1 && S < a.max_steps) {
        tprev = rmin(tnext, t1);
        const R tn = tnext + a.dt0;
        tnext = (tn > t1 - Tol<R>::v) ? t1 : tn;
        ++S;
      }
    }
    const R mf = (lane < d) ? a.fm[n * a.m_sn + (k - 1) * a.m_sk + lane * a.m_si] : R(0);
    const R Pf = inP ? a.fP[n * a.P_sn + (k - 1) * a.P_sk + (i * d + j) * a.P_si] : R(0);
    // The forward sweep kept the slopes of this interval's steps when there are at most ck_smax of them (288 GB of HBM: 3.5 KB per
    // step and trajectory is cheaper than six right-hand sides): the step starts then follow from the slopes and nothing is
    // re-integrated.  Same loops either way (ONE inlined copy of step_adj): with the slopes S <= kAdjCk is a single chunk.
    const bool have = a.ck && S <= a.ck_smax && S <= kAdjCk;
    const R* ckb = have ? a.ck + ((n * (a.T - 1) + (k - 1)) * a.ck_smax) * kCkStep : nullptr;
    auto load_slopes = [&](long s) __attribute__((always_inline)) {
      const R* c = ckb + s * kCkStep;
#pragma unroll
      for (int sg = 0; sg < 6; ++sg) {
        W[KPo + 64 * sg + lane] = c[sg * 72 + lane];
        if (lane < 8) W[KMo + 8 * sg + lane] = c[sg * 72 + 64 + lane];
      }
    };
    for (long cs = ((S - 1) / kAdjCk) * kAdjCk; cs >= 0; cs -= kAdjCk) {
      const long ce = (cs + kAdjCk < S) ? cs + kAdjCk : S;
      R mj = mf, Pij = Pf;
      R tprev = t0, tnext = rmin(t0 + a.dt0, t1);
      for (long s = 0; s < ce; ++s) {
        const R dt = dtl ? dtl[1 + s] : tnext - tprev;
        if (s >= cs) {
          const int slot = (int)(s - cs);
          W[AdjOff::ck + slot * 72 + lane] = Pij;
          if (lane < 8) W[AdjOff::ck + slot * 72 + 64 + lane] = mj;
          if (lane == 0) W[AdjOff::ck + kAdjCk * 72 + slot] = dt;
        }
        if (s + 1 < ce) {
          if (have) {
            load_slopes(s);
            step_end(mj, Pij, dt);
          } else {
            step_fwd(mj, Pij, dt);
          }
        }
        tprev = rmin(tnext, t1);
        const R tn = tnext + a.dt0;
        tnext = (tn > t1 - Tol<R>::v) ? t1 : tn;
      }
      wave_sync();
      for (long s = ce - 1; s >= cs; --s) {
        const int slot = (int)(s - cs);
        const R Ps = W[AdjOff::ck + slot * 72 + lane];
        const R ms = (lane < 8) ? W[AdjOff::ck + slot * 72 + 64 + lane] : R(0);
        const R dt = W[AdjOff::ck + kAdjCk * 72 + slot];
        if (have) load_slopes(s);
        // (the forward sweep kept the network's intermediates of the interval's FIRST step)
        const R* mckp = (MLP && have && s == 0 && a.ckm) ? a.ckm + (n * (a.T - 1) + (k - 1)) * (6L * a.ckm_nf * 64) : nullptr;
        step_adj(ms, Ps, dt, mb, Pb, have, mckp);
      }
      wave_sync();
    }
  }

  if (adj_st && lane == 0 && a.status) atomicOr(&a.status[n], adj_st);
  if ((pf_acc ^ pf_pending) == 0x5a17c0de && a.N < 0 && a.status) a.status[n] = pf_acc;  // (never: keeps the warm-up loads alive)
#ifdef CDKF_W8_PROFILE
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    printf("adjoint cycles/obs-step (sizeof real %d, order %d, ckm %d):", (int)sizeof(R), a.order, a.ckm ? 1 : 0);
    for (int q = 10; q < 21; ++q) printf(" [%d] %lld", q, w8_prof[q] / a.T);
    printf("\n");
    for (int q = 10; q < 21; ++q) w8_prof[q] = 0;
  }
#endif
  // ---- model block: m0 | P0 | LQL | H | bias | R ------------------------------------------------------------------------
  if (grad_model) {
    R* gm = grad_model + n * adj_model_grad_size(d, m);
    if (lane < d) gm[lane] = mb;
    if (inP) {
      gm[d + i * d + j] = Pb;
      gm[d + d * d + i * d + j] = gQ;
    }
    R* gh = gm + d + 2 * d * d;
    if (i < m && j < d) gh[i * d + j] = gH;
    if (lane < m) gh[m * d + lane] = gBias;
    if (i < m && j < m) gh[m * d + m + i * m + j] = gR;
  }
  // ---- drift block --------------------------------------------------------------------------------------------------------
  if constexpr (!MLP) {
    if (a.kind == kDriftLinear) {
      R* g = grad + n * (long)(d * d + d);
      if (inP) g[i * d + j] = gTile;
      if (lane < d) g[d * d + lane] = gVec;
    } else if (a.kind == kDriftLorenz63) {
      if (lane < 3) grad[n * 3 + lane] = gVec;
    } else {
      if (lane == 0) grad[n] = gVec;
    }
  } else {
  // MLP (theta ordering: W1, b1, W2, b2, W3, b3)
  R* g = grad + n * (ob3 + d);
  if (lane < h1) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < d) g[oW1 + (long)lane * d + k] = gW1[k];
    g[ob1 + lane] = gb1;
  }
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int pr = 16 * mt + MTile::row(lg, r), qc = 16 * nt + lm;
        if (pr < h2 && qc < h1) g[oW2 + (long)pr * h1 + qc] = gW2t[mt][nt][r];
      }
  if (lane < h2) {
    g[ob2 + lane] = gb2;
#pragma unroll
    for (int r = 0; r < 8; ++r)
      if (r < d) g[oW3 + (long)r * h2 + lane] = gW3[r];
  }
  if (lane < d) g[ob3 + lane] = gb3;
  }
}

}  // namespace cdkf
