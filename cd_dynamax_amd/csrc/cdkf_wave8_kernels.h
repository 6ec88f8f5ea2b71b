// cdkf_wave8_kernels.h -- wavefront-per-trajectory EKF sweep for state_dim <= 8 (the MLP-drift configuration).
//
// Mapping: one WAVEFRONT owns one trajectory for the whole time scan; four wavefronts (four
// trajectories) share a workgroup only to share the drift's weights in LDS.  Lane l = (i, j) = (l >> 3, l & 7) owns
// covariance entry P_ij and its six Dormand-Prince slopes in registers; lanes 0..7 also own the mean.  The 8 x 8
// products (F P, H P H^T, the Cholesky updates, X^T S X) exchange operands through a per-wavefront LDS tile with
// wavefront-scope fences only -- there is no workgroup barrier inside the time loop, so the four trajectories of
// a workgroup never wait for each other's RK step counts.
// MLP drift (d -> h1 -> h2 -> d, tanh, h1, h2 <= 64): lane p is hidden unit p; its rows of W2 and of
// G = ((W1 W3)^T o W2) stay in registers, the row-major copies in LDS serve the column-wise uses; the Jacobian is the
// forward-mode tangent  W3 D2 (W2 (D1 W1)).
//
// Reference functions restated: extended_kalman_filter and helpers, inference_ekf.py:46-148, 153-199, 202-326.
#pragma once
#include "cdkf_wg2_kernels.h"

namespace cdkf {

constexpr int kW8 = 8;        // lane grid is 8 x 8
constexpr int kHid = 64;      // hidden units padded to the wavefront width
constexpr int kW8Waves = 4;   // trajectories per workgroup

// accumulator row of register r in lane group g = lane >> 4 (C/D layout of v_mfma_f64_16x16x4 / v_mfma_f32_16x16x4)
template <typename R>
struct W8Tile;
template <>
struct W8Tile<double> {
  using V4 = wg_f64x4;
  static CDKF_DEV int row(int g, int r) { return g + 4 * r; }
};
template <>
struct W8Tile<float> {
  using V4 = wg_f32x4;
  static CDKF_DEV int row(int g, int r) { return 4 * g + r; }
};

#ifdef CDKF_W8_PROFILE  // local diagnostic build (scripts/w8_prof_build.sh): cycles per phase (s_memtime), printed by trajectory 0
static __device__ long long w8_prof[24];
#define W8_TICK(i)                                                              \
  {                                                                             \
    const long long w8_now = clock64();                                         \
    if (threadIdx.x == 0 && blockIdx.x == 0) w8_prof[i] += w8_now - w8_last;   \
    w8_last = clock64();                                                        \
  }
#define W8_TICK_DECL long long w8_last = clock64();
#else
#define W8_TICK(i)
#define W8_TICK_DECL
#endif

CDKF_DEV void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ---- sums over the 8 x 8 lane grid (lane = 8 i + j) without LDS: the result lands in every lane that took part -------------------------
// over j (lane bits 0 .. 2): quad permutations, then the half-row mirror; over i (bits 3 .. 5): the half-row rotation, then gfx950's
// v_permlane16_swap / v_permlane32_swap (a register swapped with its own copy: both halves of the pair come back, their sum is the
// all-reduce over that lane bit).  64-bit values travel as two 32-bit DPP moves; fp32 folds the DPP control into the add.
template <int CTRL>
CDKF_DEV float w8_dpp(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
CDKF_DEV double w8_dpp(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// The swaps are written as inline assembly with two read-write registers: through __builtin_amdgcn_permlane{16,32}_swap the fp32 sums
// came out as r0 + r0 (ROCm 7.2 lowers the builtin's second result to the first register when both feed one 32-bit add:
// "v_permlane16_swap_b32 v4, v0; v_add_f32 v0, v4, v4"), i.e. twice one half of the sum -- 'second'-order fp32 sweeps were off by
// 1e-2 from this round's restructuring on; found by scripts/gpu_fuzz_filters.py.  (s_nop 1: the two wait states a VALU-written
// register needs before a lane-permuting instruction reads it, which the compiler cannot insert around inline assembly.)
CDKF_DEV void w8_swap16(unsigned& a, unsigned& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
CDKF_DEV void w8_swap32(unsigned& a, unsigned& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
CDKF_DEV float w8_swap_sum16(float x) {
  unsigned a = __builtin_bit_cast(unsigned, x), b = a;
  w8_swap16(a, b);
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
CDKF_DEV float w8_swap_sum32(float x) {
  unsigned a = __builtin_bit_cast(unsigned, x), b = a;
  w8_swap32(a, b);
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
CDKF_DEV double w8_swap_sum16(double x) {
  unsigned la = __double2loint(x), lb = la, ha = __double2hiint(x), hb = ha;
  w8_swap16(la, lb);
  w8_swap16(ha, hb);
  return __hiloint2double(ha, la) + __hiloint2double(hb, lb);
}
CDKF_DEV double w8_swap_sum32(double x) {
  unsigned la = __double2loint(x), lb = la, ha = __double2hiint(x), hb = ha;
  w8_swap32(la, lb);
  w8_swap32(ha, hb);
  return __hiloint2double(ha, la) + __hiloint2double(hb, lb);
}
template <typename R>
CDKF_DEV R w8_sum_j(R x) {  // sum over the eight lanes of a grid row
  x += w8_dpp<0xB1>(x);   // quad_perm [1, 0, 3, 2]
  x += w8_dpp<0x4E>(x);   // quad_perm [2, 3, 0, 1]
  x += w8_dpp<0x141>(x);  // row_half_mirror: the other quad of the eight
  return x;
}
template <typename R>
CDKF_DEV R w8_sum_i(R x) {  // sum over the eight grid rows (same j)
  x += w8_dpp<0x128>(x);  // row_ror:8: the other half of the sixteen-lane row
  x = w8_swap_sum16(x);
  x = w8_swap_sum32(x);
  return x;
}

// per-wavefront LDS tile (in reals)
struct W8Off {
  static constexpr int P = 0, A = 64, F = 128, X = 192, SX = 256, S1 = 320, S2 = 384, HP = 448;
  static constexpr int x = 512, f = 520, g = 528, v = 536, z = 544, y = 552, tmp = 560;
  static constexpr int km = 568;  // mean parts of the Dormand-Prince slopes k1 .. k5 of the step in hand [5][8]; their covariance
                                  // parts take the five tiles X .. HP, which only the measurement update uses
  static constexpr int base_end = 608;
  // MLP only
  static constexpr int a1 = 608, d1 = 672, a2 = 736, d2 = 800, s2 = 864, tq = 928;
  static constexpr int U = 992, Z = 992 + 576;  // two [64][9] images (stride 9: a lane's row is conflict-free), state_order 'second'
  static constexpr int mlp_end = 992 + 2 * 576;
  static constexpr int rk = 36;  // behind either layout: the Dormand-Prince a[sg][jj] as a 6 x 6 table with zeros for jj >= sg (uniform LDS
                                 // reads in one basic block instead of a constant-memory load and a branch per entry of the rolled stage loop)
};
struct W8Sh {  // per workgroup: the MLP's weights, hidden sizes padded to 64, state to 8.  W2 is read from LDS as the B operand of the
                // transposed product only (lane (lg, lm) takes W2[p0 + lg][16 nt + lm]): with rows of LD2 = 80 (= 16 mod 32 doubles, = 16
                // mod 64 words) the four rows of a k-step fall into disjoint bank ranges -- no conflicts in either precision (rows of 65
                // put the lanes with equal lg + lm on one bank: four-way)
  static constexpr int LD2 = 80;
  static constexpr int W1 = 0, b1 = 512, W2 = 576, b2 = 576 + 64 * LD2, W3 = b2 + 64, b3 = W3 + 8 * 65;
  static constexpr int end = b3 + 8;
};
__host__ __device__ inline long wave8_lds_reals(int kind) {
  return (kind == kDriftMlp) ? (long)W8Sh::end + kW8Waves * (W8Off::mlp_end + W8Off::rk) : (long)kW8Waves * (W8Off::base_end + W8Off::rk);
}

// ---- the measurement update of one observation on the 8 x 8 lane grid (inference_ekf.py:153-199, 285-286), shared by the sweeps of this
// file and of cdkf_wave8s_kernels.h: W = the wavefront's tile block (W8Off offsets P .. tmp), lane (i, j) holds P_ij, lanes < d the mean.
// Two factorisations side by side (TFP's of S for the log-likelihood, psd_solve's of sym(S) + 1e-9 I for the gain), num_iter
// relinearisations, then symmetrize.  Wavefront-scope synchronisation only.
// SYM = false: without the final symmetrisation (one iteration of several, for the reverse sweep's recomputation of their inputs).
template <typename R, bool SYM = true>
CDKF_DEV void w8_measurement_update(R* W, int lane, int i, int j, int d, int m, bool inP, bool hsel, R Hij, R Rij, R hbj, R yl, int num_iter,
                                    int forecast, R& Pij, R& mj, double& ll, bool& bad) {
  for (int it = 0; it < (forecast ? 0 : num_iter); ++it) {
    W[W8Off::P + lane] = Pij;
    if (lane < kW8) W[W8Off::x + lane] = mj;
    wave_sync();
    // HP[r][c] (lane (r=i, c=j)) = sum_k H[r][k] P[k][c];  H row r lives on lanes (r, *)
    R hp;
    if (hsel) {
      hp = (i < m) ? Pij : R(0);
    } else {
      W[W8Off::F + lane] = Hij;  // borrow the F tile for H
      wave_sync();
      hp = 0;
#pragma unroll
      for (int kk = 0; kk < kW8; ++kk) hp = rfma(W[W8Off::F + i * kW8 + kk], W[W8Off::P + kk * kW8 + j], hp);
    }
    W[W8Off::HP + lane] = hp;
    wave_sync();
    // S[r][c] = sum_k HP[r][k] H[c][k] + R[r][c]
    R s;
    if (hsel) {
      s = (i < m && j < m) ? Pij + Rij : R(0);
    } else {
      s = 0;
#pragma unroll
      for (int kk = 0; kk < kW8; ++kk) s = rfma(W[W8Off::HP + i * kW8 + kk], W[W8Off::F + j * kW8 + kk], s);
      s = (i < m && j < m) ? s + Rij : R(0);
    }
    // innovation on lanes < m
    R vv = 0;
    if (lane < kW8) {
      if (hsel) {
        vv = (lane < m) ? yl - mj : R(0);
      } else {
        R hm = 0;
#pragma unroll
        for (int kk = 0; kk < kW8; ++kk) hm = rfma(W[W8Off::F + lane * kW8 + kk], W[W8Off::x + kk], hm);
        vv = (lane < m) ? yl - (hm + hbj) : R(0);
      }
      W[W8Off::v + lane] = vv;
    }
    // two factorisations side by side: S1 = S (TFP), S2 = symmetrize(S) + 1e-9 I (psd_solve); pad with identity
    W[W8Off::S1 + lane] = s;
    wave_sync();
    R s1 = (i < m && j < m) ? s : (i == j ? R(1) : R(0));
    R s2v = (i < m && j < m) ? R(0.5) * (s + W[W8Off::S1 + j * kW8 + i]) + (i == j ? R(1e-9) : R(0)) : (i == j ? R(1) : R(0));
    wave_sync();
    R inv1[kW8], inv2[kW8];
#pragma unroll
    for (int p = 0; p < kW8; ++p) {
      if (p >= m) {  // (uniform) identity padding: pivot 1, nothing below it -- two synchronisations saved per padded column
        inv1[p] = inv2[p] = R(1);
        continue;
      }
      W[W8Off::S1 + lane] = s1;
      W[W8Off::S2 + lane] = s2v;
      wave_sync();
      const R p1 = W[W8Off::S1 + p * kW8 + p], p2 = W[W8Off::S2 + p * kW8 + p];
      if (p < m && (!(p1 > R(0)) || !(p2 > R(0)))) bad = true;
      const R r1 = rrsqrt(p1), r2 = rrsqrt(p2);
      inv1[p] = r1;
      inv2[p] = r2;
      // column p below the pivot, scaled; then the trailing update of the lower triangle
      const R l1i = W[W8Off::S1 + i * kW8 + p] * r1, l1j = W[W8Off::S1 + j * kW8 + p] * r1;
      const R l2i = W[W8Off::S2 + i * kW8 + p] * r2, l2j = W[W8Off::S2 + j * kW8 + p] * r2;
      wave_sync();
      if (j == p && i >= p) {
        s1 = (i == p) ? p1 * r1 : l1i;
        s2v = (i == p) ? p2 * r2 : l2i;
      } else if (i > p && j > p && j <= i) {
        s1 = rfma(-l1i, l1j, s1);
        s2v = rfma(-l2i, l2j, s2v);
      }
    }
    // now s1 / s2v hold L1 / L2 (lower triangles)
    W[W8Off::S1 + lane] = s1;
    W[W8Off::S2 + lane] = s2v;
    wave_sync();
    if (it == 0) {
      // z = L1^-1 v (forward substitution on every lane redundantly), log-likelihood term
      R z[kW8];
      double qd = 0.0, pinv = 1.0;
#pragma unroll
      for (int r = 0; r < kW8; ++r) {
        R w = W[W8Off::v + r];
#pragma unroll
        for (int c = 0; c < r; ++c) w = rfma(-W[W8Off::S1 + r * kW8 + c], z[c], w);
        z[r] = w * inv1[r];
        if (r < m) {
          qd += (double)z[r] * (double)z[r];
          pinv *= (double)inv1[r];
        }
      }
      ll += -0.5 * qd + log(pinv) - 0.5 * m * 1.8378770664093454835606594728112;
    }
    // X = Sb^-1 HP : column c = j solved by the 8 lanes (*, j); every lane walks its own column redundantly
    R xcol[kW8];
#pragma unroll
    for (int r = 0; r < kW8; ++r) {
      R w = W[W8Off::HP + r * kW8 + j];
#pragma unroll
      for (int c = 0; c < r; ++c) w = rfma(-W[W8Off::S2 + r * kW8 + c], xcol[c], w);
      xcol[r] = w * inv2[r];
    }
#pragma unroll
    for (int r = kW8 - 1; r >= 0; --r) {
      R w = xcol[r];
#pragma unroll
      for (int c = r + 1; c < kW8; ++c) w = rfma(-W[W8Off::S2 + c * kW8 + r], xcol[c], w);
      xcol[r] = w * inv2[r];
    }
    R xij = 0;  // X[r=i][c=j] (select instead of a run-time register index)
#pragma unroll
    for (int r = 0; r < kW8; ++r)
      if (r == i && r < m) xij = xcol[r];
    W[W8Off::X + lane] = xij;
    wave_sync();
    // SX[r][c] = sum_q S[r][q] X[q][c]   (S re-read from the original tile: recompute from s saved in HP slot order)
    // the original S was overwritten by its factor; keep a copy in SX's slot first
    W[W8Off::SX + lane] = s;
    wave_sync();
    R sx = 0;
#pragma unroll
    for (int q = 0; q < kW8; ++q) sx = rfma(W[W8Off::SX + i * kW8 + q], xcol[q] * ((q < m) ? R(1) : R(0)), sx);
    wave_sync();
    W[W8Off::SX + lane] = (i < m) ? sx : R(0);
    wave_sync();
    // T[a][b] = sum_r X[r][a] SX[r][b];  P <- P - T;  m <- m + X^T v
    R tt = 0;
#pragma unroll
    for (int r = 0; r < kW8; ++r) tt = rfma(W[W8Off::X + r * kW8 + i], W[W8Off::SX + r * kW8 + j], tt);
    if (inP) Pij -= tt;
    if (lane < kW8) {
      R dm = 0;
#pragma unroll
      for (int r = 0; r < kW8; ++r) dm = rfma(W[W8Off::X + r * kW8 + lane], W[W8Off::v + r], dm);
      if (lane < d) mj += dm;
    }
    wave_sync();
  }
  if constexpr (!SYM) return;
  // symmetrize
  W[W8Off::P + lane] = Pij;
  wave_sync();
  Pij = R(0.5) * (Pij + W[W8Off::P + j * kW8 + i]);
  wave_sync();
}

template <typename R>
__global__ __launch_bounds__(256, 1) void ekf_filter_wave8_kernel(const WgArgs<R> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  R* smem = reinterpret_cast<R*>(smem_raw);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = lane >> 3, j = lane & 7;
  const int d = a.d, m = a.m;
  const bool mlp = a.kind == kDriftMlp;
  const int h1 = a.h1, h2 = a.h2;
  R* Sh = smem;
  R* W = smem + (mlp ? W8Sh::end : 0) + wave * ((mlp ? W8Off::mlp_end : W8Off::base_end) + W8Off::rk);
  const int o_rk = mlp ? W8Off::mlp_end : W8Off::base_end;
  const R* th = a.par + a.o_theta;

  // ---- shared MLP weights (zero-padded), built by the whole workgroup --------------------------------------------
  if (mlp) {
    for (int e = threadIdx.x; e < W8Sh::end; e += blockDim.x) Sh[e] = 0;
    __syncthreads();
    const R* gW1 = th;
    const R* gb1 = gW1 + h1 * d;
    const R* gW2 = gb1 + h1;
    const R* gb2 = gW2 + h2 * h1;
    const R* gW3 = gb2 + h2;
    const R* gb3 = gW3 + d * h2;
    for (int e = threadIdx.x; e < h1 * d; e += blockDim.x) Sh[W8Sh::W1 + fdiv(e, d) * kW8 + (e - fdiv(e, d) * d)] = gW1[e];
    for (int e = threadIdx.x; e < h1; e += blockDim.x) Sh[W8Sh::b1 + e] = gb1[e];
    for (int e = threadIdx.x; e < h2 * h1; e += blockDim.x) Sh[W8Sh::W2 + fdiv(e, h1) * W8Sh::LD2 + (e - fdiv(e, h1) * h1)] = gW2[e];
    for (int e = threadIdx.x; e < h2; e += blockDim.x) Sh[W8Sh::b2 + e] = gb2[e];
    for (int e = threadIdx.x; e < d * h2; e += blockDim.x) Sh[W8Sh::W3 + fdiv(e, h2) * 65 + (e - fdiv(e, h2) * h2)] = gW3[e];
    for (int e = threadIdx.x; e < d; e += blockDim.x) Sh[W8Sh::b3 + e] = gb3[e];
    __syncthreads();
  }
  const long n = (long)blockIdx.x * kW8Waves + wave;
  if (n >= a.N) return;  // whole wavefront; no workgroup barrier follows
  {  // tableau table; the slope tiles start from zeros (a zero coefficient must not meet the NaN patterns uninitialised LDS may hold;
     // afterwards the tiles only ever hold this trajectory's own numbers: slopes, or the measurement update's intermediates)
    using TBi = Dp5T<R>;
    if (lane < 36) {
      const int r = lane / 6, c = lane - 6 * r;
      R v = 0;
#pragma unroll
      for (int rr = 1; rr < 6; ++rr)
#pragma unroll
        for (int cc = 0; cc < 5; ++cc)
          if (rr == r && cc == c && cc < rr) v = TBi::a[rr][cc];
      W[o_rk + lane] = v;
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) W[W8Off::X + 64 * q + lane] = R(0);
    if (lane < 40) W[W8Off::km + lane] = R(0);
    wave_sync();
  }

  // ---- per-lane constants ---------------------------------------------------------------------------------------
  const bool inP = (i < d) && (j < d);
  const R lql = inP ? (a.par + a.o_LQL)[i * d + j] : R(0);
  const R lqlz = inP ? (a.par + a.o_LQLz)[i * d + j] : R(0);
  const bool hsel = a.hsel != 0;
  const R Hij = (i < m && j < d) ? (a.par + a.o_H)[i * d + j] : R(0);       // lane (r=i, k=j) holds H[r][k]
  const R Rij = (i < m && j < m) ? (a.par + a.o_R)[i * m + j] : R(0);       // lane (r, c) holds R[r][c]
  const R hbj = (lane < m) ? (a.par + a.o_hb)[lane] : R(0);
  const R Wlin = (a.kind == kDriftLinear && inP) ? th[i * d + j] : R(0);    // linear drift: lane (i,k) holds W[i][k]
  const R blin = (a.kind == kDriftLinear && lane < d) ? th[d * d + lane] : R(0);
  // MLP: the layer products run on the matrix cores (v_mfma_*_16x16x4: A[m = lane & 15][k = lane >> 4], B[k][n = lane & 15]).
  // Operands that are weights stay in registers for the whole sweep, in the layout the instruction wants:
  //   w2A[mt][ks] = W2[16 mt + lm][4 ks + lg]                     A operand of  [T | z2] = W2 [D1 W1 | a1]      (64 x 64 x 9)
  //   w1B[ks]     = W1[4 ks + lg][lm] (lm < 8)                    weight part of that product's B operand
  //   w3A[mt][r]  = W3[lm][16 mt + row(lg, r)] (lm < 8)           A operand of  [F | f] = W3 [D2 T | a2]        (8 x 64 x 9):
  //                 its k-steps are taken in the order in which the first product's accumulator holds the rows of T, so the
  //                 accumulator registers ARE the B operand (scaled by d2): no data movement between the two products.
  // (the rows of W2 / G that the scalar code kept pinned -- 128 registers each -- are gone; grad(div f) uses
  //  s_p = sum_i W3[i][p] T[p][i] instead of sum_q G[p][q] d1_q: the same number, from the tangent that is already there)
  using MTile = W8Tile<R>;
  const int lm = lane & 15, lg = lane >> 4;
  R w1row[kW8], w3col[kW8], w2A[4][16], w1B[16], w3A[4][4];
  R b1l = 0, b2l = 0;
  const R e8 = (lm == 8) ? R(1) : R(0), ne8 = (lm == 8) ? R(0) : R(1);
  const int sc_off = (lm == 8) ? W8Off::a2 : W8Off::d2;  // column 8 of the second product carries a2 (the drift itself)
  const bool second = (a.order == 2) && mlp;
  if (mlp) {
#pragma unroll
    for (int jj = 0; jj < kW8; ++jj) {
      w1row[jj] = pin(Sh[W8Sh::W1 + lane * kW8 + jj]);
      w3col[jj] = pin(Sh[W8Sh::W3 + jj * 65 + lane]);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) w2A[mt][ks] = pin(Sh[W8Sh::W2 + (16 * mt + lm) * W8Sh::LD2 + 4 * ks + lg]);
#pragma unroll
      for (int r = 0; r < 4; ++r) w3A[mt][r] = pin(lm < kW8 ? Sh[W8Sh::W3 + lm * 65 + 16 * mt + MTile::row(lg, r)] : R(0));
    }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) w1B[ks] = pin(lm < kW8 ? Sh[W8Sh::W1 + (4 * ks + lg) * kW8 + lm] : R(0));
    b1l = Sh[W8Sh::b1 + lane];
    b2l = Sh[W8Sh::b2 + lane];
  }

  // ---- state ---------------------------------------------------------------------------------------------------------
  R Pij = inP ? R(0.5) * ((a.par + a.o_P0)[i * d + j] + (a.par + a.o_P0)[j * d + i]) : R(0);
  R mj = (lane < d) ? (a.par + a.o_m0)[lane] : R(0);  // lanes 0..d-1 own the mean
  double ll = 0.0;
  int st = 0;
  bool bad = false;
  const bool zeroth = a.order == 0;

  // right-hand side of the moment ODEs for the stage value (xs: mean on lanes < d, Ps: this lane's covariance entry)
  W8_TICK_DECL
  R* mck = nullptr;  // MLP stage checkpoint of the right-hand side in hand (reverse sweep's forward pass only; uniform)
  auto rhs = [&](R xs, R Ps, R& kM, R& kP) __attribute__((always_inline)) {
    W8_TICK(0)  // stage combination (outside the right-hand side)
    W[W8Off::P + lane] = Ps;
    if (lane < kW8) W[W8Off::x + lane] = xs;
    wave_sync();
    R xk[kW8];
#pragma unroll
    for (int k = 0; k < kW8; ++k) xk[k] = W[W8Off::x + k];
    R Fij = 0;  // lane (i, k=j) computes F[i][k]
    R fi = 0;   // lanes < d: f_lane
    if (a.kind == kDriftLinear) {
      Fij = Wlin;
      // f_i = sum_k W[i][k] x_k + b_i : every lane forms its product, row sums through the tile
      W[W8Off::A + lane] = Wlin * xk[j];
      wave_sync();
      if (lane < kW8) {
        R s = blin;
#pragma unroll
        for (int k = 0; k < kW8; ++k) s += W[W8Off::A + lane * kW8 + k];
        fi = s;
      }
      wave_sync();
    } else if (a.kind == kDriftLorenz63) {
      const R sg = th[0], rho = th[1], bt = th[2];
      if (i == 0) Fij = (j == 0) ? -sg : (j == 1 ? sg : R(0));
      if (i == 1) Fij = (j == 0) ? rho - xk[2] : (j == 1 ? R(-1) : (j == 2 ? -xk[0] : R(0)));
      if (i == 2) Fij = (j == 0) ? xk[1] : (j == 1 ? xk[0] : (j == 2 ? -bt : R(0)));
      if (!inP) Fij = 0;
      if (lane == 0) fi = sg * (xk[1] - xk[0]);
      if (lane == 1) fi = xk[0] * (rho - xk[2]) - xk[1];
      if (lane == 2) fi = xk[0] * xk[1] - bt * xk[2];
    } else if (a.kind == kDriftLorenz96) {
      auto X = [&](int q) { return W[W8Off::x + q]; };
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
    } else {  // MLP: six synchronisations per right-hand side (state_order 'second'; five otherwise)
      // ---- layer 1: lane = hidden unit q ----------------------------------------------------------------------------
      R z1 = b1l;
#pragma unroll
      for (int k = 0; k < kW8; ++k) z1 = rfma(w1row[k], xk[k], z1);
      const R a1 = rtanh_fast(z1);
      const R d1 = R(1) - a1 * a1;
      W[W8Off::a1 + lane] = a1;
      W[W8Off::d1 + lane] = d1;
      if (mck) mck[kMlpCkA1 * 64 + lane] = a1;
      wave_sync();
      W8_TICK(1)  // state broadcast, layer 1, tanh
      // ---- layer 2 on the matrix cores: acc[mt][r] = [T | z2 - b2][16 mt + row(lg, r)][lm],  T = W2 D1 W1 (tangent), column 8: W2 a1
      typename MTile::V4 acc[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt] = typename MTile::V4{0, 0, 0, 0};
      if constexpr (sizeof(R) == 4) {
        // fp32: a k-step's four products take 128 cycles, an LDS round trip about as long -- one k-step of look-ahead leaves the
        // matrix pipe waiting (scripts/mb/mb_mfma16.hip: 2 950 cycles per product against 2 048 for the pipe alone), and there are
        // registers to spare: every activation of the product is requested up front
        R dq[16], aq[16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          dq[ks] = W[W8Off::d1 + 4 * ks + lg];
          aq[ks] = W[W8Off::a1 + 4 * ks + lg];
        }
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          const R bv = rfma(dq[ks], w1B[ks], aq[ks] * e8);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) acc[mt] = wg_mfma(w2A[mt][ks], bv, acc[mt]);
        }
      } else {  // (software pipeline as below: the activations of k-step ks + 1 are requested before the products of k-step ks are issued)
        R dq_n = W[W8Off::d1 + lg], aq_n = W[W8Off::a1 + lg];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          R bv = rfma(dq_n, w1B[ks], aq_n * e8);
          if (ks < 15) {
            dq_n = W[W8Off::d1 + 4 * (ks + 1) + lg];
            aq_n = W[W8Off::a1 + 4 * (ks + 1) + lg];
          }
          asm volatile("" : "+v"(bv) : : "memory");
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) acc[mt] = wg_mfma(w2A[mt][ks], bv, acc[mt]);
        }
      }
      if (lm == 8) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) W[W8Off::s2 + 16 * mt + MTile::row(lg, r)] = acc[mt][r];
      }
      const bool rows = second || mck;  // the rows of T in lane = hidden-unit order: grad(div f) and the checkpoint
      if (rows && lm < kW8) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) W[W8Off::U + (16 * mt + MTile::row(lg, r)) * 9 + lm] = acc[mt][r];
      }
      wave_sync();
      W8_TICK(2)  // tangent product (64 MFMA) + its images
      // ---- layer 2 activations, s_p = sum_i W3[i][p] T[p][i] and s2 = -2 a2 d2 s in one phase: ONE synchronisation serves both products below
      const R z2 = W[W8Off::s2 + lane] + b2l;  // lane = hidden unit p
      const R a2 = rtanh_fast(z2);
      const R d2 = R(1) - a2 * a2;
      W[W8Off::a2 + lane] = a2;
      W[W8Off::d2 + lane] = d2;
      if (rows) {
        R sdiv = 0;
#pragma unroll
        for (int k = 0; k < kW8; ++k) {
          const R tk = W[W8Off::U + lane * 9 + k];
          sdiv = rfma(w3col[k], tk, sdiv);
          if (mck) mck[(kMlpCkT + k) * 64 + lane] = tk;
        }
        if (second) W[W8Off::s2 + lane] = R(-2) * a2 * d2 * sdiv;  // (this lane's own slot: z2 - b2 has been taken above)
        if (mck) {
          mck[kMlpCkA2 * 64 + lane] = a2;
          if (second) mck[kMlpCkS * 64 + lane] = sdiv;
        }
      }
      wave_sync();
      W8_TICK(3)  // tanh of layer 2, s, s2
      // ---- layer 3 and, for 'second', the transposed product, k-step by k-step on FIVE independent accumulator chains ----------------
      //   [F | f - b3] = W3 [D2 T | a2]: its k index runs over the rows the first product's accumulators hold (scaled by d2; column 8: a2);
      //   [E1 | tc]^T  = [diag(d2) W3^T | s2]^T W2  (g = grad(div f) = W1^T tq, tq = d1 (-2 a1 td + tc), td_q = sum_i W1[q][i] E1[q][i];
      //                  oracle/cdkf_oracle.py MLPDrift.divgrad; G = (W1 W3)^T * W2 is never formed): the same k order, A operand from the
      //                  same pinned W3 slice, B operand = W2 rows from LDS.  The sixteen chained products of layer 3 hide behind the 64 here.
      typename MTile::V4 acc3{0, 0, 0, 0};
      typename MTile::V4 cacc[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) cacc[nt] = typename MTile::V4{0, 0, 0, 0};
      const int off2 = (lm == 8) ? W8Off::s2 : W8Off::d2;
      if (second) {
        // software pipeline: the LDS operands of k-step it + 1 are requested BEFORE the five products of k-step it are issued (issued
        // after them they would arrive while the matrix pipe has already drained: a quarter of every k-step idle)
        auto prow = [&](int it) __attribute__((always_inline)) { return 16 * (it >> 2) + MTile::row(lg, it & 3); };
        // look-ahead in k-steps: one in fp64 (five products = 320 cycles cover an LDS round trip), four in fp32 (160 cycles do not)
        constexpr int LA = sizeof(R) == 4 ? 4 : 1;
        R scq[16], x2q[16], wq[16][4];
        auto request = [&](int it) __attribute__((always_inline)) {
          const int pn = prow(it);
          scq[it] = W[sc_off + pn];
          x2q[it] = W[off2 + pn];
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) wq[it][nt] = Sh[W8Sh::W2 + pn * W8Sh::LD2 + 16 * nt + lm];
        };
#pragma unroll
        for (int it = 0; it < LA; ++it) request(it);
#pragma unroll
        for (int it = 0; it < 16; ++it) {
          const R sc = scq[it], x2 = x2q[it], w0 = wq[it][0], w1 = wq[it][1], w2 = wq[it][2], w3 = wq[it][3];
          const R w3a = w3A[it >> 2][it & 3];
          R av = x2 * (w3a + e8);  // lm < 8: d2_p W3[lm][p];  lm = 8: s2_p;  else 0
          R b3v = sc * rfma(acc[it >> 2][it & 3], ne8, e8);
          if (it + LA < 16) request(it + LA);
          // (the products' operands pass through this fence: the loads above stay above it, the products below stay below)
          asm volatile("" : "+v"(av), "+v"(b3v) : : "memory");
          acc3 = wg_mfma(w3a, b3v, acc3);
          cacc[0] = wg_mfma(av, w0, cacc[0]);
          cacc[1] = wg_mfma(av, w1, cacc[1]);
          cacc[2] = wg_mfma(av, w2, cacc[2]);
          cacc[3] = wg_mfma(av, w3, cacc[3]);
        }
      } else {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const R sc = W[sc_off + 16 * mt + MTile::row(lg, r)];
            acc3 = wg_mfma(w3A[mt][r], sc * rfma(acc[mt][r], ne8, e8), acc3);
          }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = MTile::row(lg, r);
        if (row < kW8 && lm < kW8) W[W8Off::F + row * kW8 + lm] = acc3[r];       // F[row][lm]
        if (row < kW8 && lm == 8) W[W8Off::f + row] = acc3[r] + Sh[W8Sh::b3 + row];
      }
      if (second) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = MTile::row(lg, r);
            if (row < 9) W[W8Off::U + (16 * nt + lm) * 9 + row] = cacc[nt][r];  // (the rows of T were consumed above)
          }
      }
      wave_sync();
      W8_TICK(4)  // layer 3 (16 MFMA) + transposed product (64 MFMA), F, f, E1 | tc to LDS
      // ---- k_P = F P + P F^T + L Qc L^T straight from the two tiles (no transposed partner to wait for) ---------------------------------
      if (lane < kW8) kM = (lane < d) ? W[W8Off::f + lane] : R(0);
      if (mck) mck[kMlpCkF * 64 + lane] = inP ? W[W8Off::F + lane] : R(0);
      if (!zeroth) {
        // (F P)_ij and (F P)_ji -- the second sum is, term by term and in the same order, what lane (j, i) forms as its first:
        // k_P stays exactly symmetric without waiting for the partner's value
        R sa = 0, sb = 0;
#pragma unroll
        for (int k = 0; k < kW8; ++k) {
          sa = rfma(W[W8Off::F + i * kW8 + k], W[W8Off::P + k * kW8 + j], sa);
          sb = rfma(W[W8Off::F + j * kW8 + k], W[W8Off::P + k * kW8 + i], sb);
        }
        kP = (sa + sb) + lql;
      }
      if (second) {
        // tq, then g_i = sum_q W1[q][i] tq_q: lane (i, j) adds the hidden units q = 8 c + j, the eight partial sums of a grid row meet by
        // DPP; 0.5 (P g)_j = 0.5 sum_i P_ij g_i over the grid rows by the half-row rotation and the two swaps -- no LDS round trip
        R td = 0;
#pragma unroll
        for (int k = 0; k < kW8; ++k) {
          const R ek = W[W8Off::U + lane * 9 + k];
          td = rfma(w1row[k], ek, td);
          if (mck) mck[(kMlpCkE1 + k) * 64 + lane] = ek;
        }
        const R tqv = d1 * rfma(R(-2) * a1, td, W[W8Off::U + lane * 9 + 8]);
        if (mck) {
          mck[kMlpCkTd * 64 + lane] = td;
          mck[kMlpCkTq * 64 + lane] = tqv;
        }
#pragma unroll
        for (int k = 0; k < kW8; ++k) W[W8Off::Z + lane * 8 + k] = w1row[k] * tqv;
        wave_sync();
        R part = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) part += W[W8Off::Z + (8 * c + j) * 8 + i];
        const R gi = w8_sum_j(part);  // g_i in the lanes (i, *)
        if (mck && j == 0) mck[kMlpCkG * 64 + i] = (i < d) ? gi : R(0);
        if (!zeroth) {
          const R pg = w8_sum_i(Ps * gi);  // (P g)_j in the lanes (*, j)
          if (lane < kW8) kM = rfma(R(0.5), pg, kM);
        }
      }
      W8_TICK(6)  // k_P, td, tq, g, 0.5 P g
      return;
    }
    if (a.ukf && lane < d) {
      // The UNSCENTED filter's mean equation for a quadratic drift, in closed form (exact for every alpha, beta, kappa; oracle:
      // ukf_curvature): f(m) + 0.5 sum_jk (d^2 f / dx_j dx_k) Ps_jk -- the covariance equation is the extended filter's.  Set by the
      // unscented filter's reverse-sweep gradient only (cdkf_ukf_loglik_grad_all_*), whose forward pass this then is.
      auto Pe = [&](int r, int c) __attribute__((always_inline)) { return W[W8Off::P + r * kW8 + c]; };
      if (a.kind == kDriftLorenz63) {
        if (lane == 1) fi -= Pe(0, 2);
        if (lane == 2) fi += Pe(0, 1);
      } else if (a.kind == kDriftLorenz96) {
        const int l = lane;
        const int lp1 = (l + 1 >= d) ? 0 : l + 1, lm1 = (l == 0) ? d - 1 : l - 1, lm2 = (lm1 == 0) ? d - 1 : lm1 - 1;
        fi += Pe(lp1, lm1) - Pe(lm2, lm1);
      }
    }
    if (lane < kW8) kM = (lane < d) ? fi : R(0);
    if (zeroth) return;
    // A = F Ps ; kP = A_ij + A_ji + LQL_ij
    W[W8Off::F + lane] = Fij;
    wave_sync();
    R acc = 0;
#pragma unroll
    for (int k = 0; k < kW8; ++k) acc = rfma(W[W8Off::F + i * kW8 + k], W[W8Off::P + k * kW8 + j], acc);
    W[W8Off::A + lane] = acc;
    wave_sync();
    kP = (acc + W[W8Off::A + j * kW8 + i]) + lql;
    if (second && lane < kW8) {
      R s = 0;
#pragma unroll
      for (int k = 0; k < kW8; ++k) s = rfma(W[W8Off::g + k], W[W8Off::P + k * kW8 + lane], s);
      kM = rfma(R(0.5), s, kM);
    }
    wave_sync();
    W8_TICK(7)  // F P, k_P, 0.5 P g
  };

  const R* tp = a.t + n * a.t_sn;
  const R* yp = a.y + n * a.y_sn;
  using C = Dp5<R>;
  for (long k = 0; k < a.T; ++k) {
    W8_TICK(8)  // end of step: combination, stores
    // ---------------- update (inference_ekf.py:153-199, 285-286) ----------------
    const R yl = (lane < m) ? yp[k * a.y_sk + lane * a.y_si] : R(0);
    w8_measurement_update<R>(W, lane, i, j, d, m, inP, hsel, Hij, Rij, hbj, yl, a.num_iter, a.forecast, Pij, mj, ll, bad);
    if (mj != mj) st |= kStatusNan;
    // ---------------- store filtered ----------------
    if (a.fm && lane < d) a.fm[n * a.m_sn + k * a.m_sk + lane * a.m_si] = mj;
    if (a.fP && inP) a.fP[n * a.P_sn + k * a.P_sk + (i * d + j) * a.P_si] = Pij;
    W8_TICK(9)  // measurement update, symmetrise, filtered stores
    // ---------------- predict ----------------
    const R t0 = tp[k * a.t_sk];
    const R t1 = (k + 1 < a.T) ? tp[(k + 1) * a.t_sk] : t0 + a.dt_final;
    {
      R tprev = t0;
      R tnext = rmin(t0 + a.dt0, t1);
      long steps = 0;
      while (tprev < t1) {  // uniform over the wavefront
        if (steps >= a.max_steps) {
          st |= kStatusMaxSteps;
          break;
        }
        const R dt = tnext - tprev;
        R* ckp = (a.ck && k + 1 < a.T && steps < a.ck_smax) ? a.ck + ((n * (a.T - 1) + k) * a.ck_smax + steps) * kCkStep : nullptr;
        R* mckp = (a.ckm && k + 1 < a.T && steps == 0) ? a.ckm + (n * (a.T - 1) + k) * (6L * a.ckm_nf * 64) : nullptr;
        // The stage loop stays ROLLED (one inlined copy of the right-hand side instead of six: the unrolled sweep was several
        // times the instruction cache): the slopes k1 .. k5 wait in LDS -- a lane reads back only what it wrote itself, no
        // synchronisation -- and the stage index may be a run-time value.  Same sums in the same order as the unrolled form.
        using TB = Dp5T<R>;
        R kM6 = 0, kP6 = 0;
#pragma unroll 1
        for (int sg = 0; sg < 6; ++sg) {
          R sm = 0, sp = 0;
#pragma unroll
          for (int jj = 0; jj < 5; ++jj) {  // (jj >= sg: zero coefficient -- the same sums as the guarded form, without its branches)
            const R c = W[o_rk + 6 * sg + jj];
            sm = rfma(c, W[W8Off::km + 8 * jj + (lane & 7)], sm);
            sp = rfma(c, W[W8Off::X + 64 * jj + lane], sp);
          }
          R kM = 0, kP = 0;
          mck = mckp ? mckp + (long)sg * a.ckm_nf * 64 : nullptr;
          rhs(rfma(dt, sm, mj), rfma(dt, sp, Pij), kM, kP);
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
        {
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
    if (zeroth) Pij = rfma(rsqrt_(t1 - t0), lqlz, Pij);
    if (a.cj && lane < d) mj += a.cj[(n * a.T + k) * d + lane];  // (a jump of the predicted mean: WgArgs::cj)
    if (a.pm && lane < d) a.pm[n * a.m_sn + k * a.m_sk + lane * a.m_si] = mj;
    if (a.pP && inP) a.pP[n * a.P_sn + k * a.P_sk + (i * d + j) * a.P_si] = Pij;
  }
  if (lane == 0) {
    if (bad) st |= kStatusNotPd;
    if (ll != ll) st |= kStatusNan;
    a.ll[n] = (R)ll;
    if (a.status) a.status[n] = st;
  }
#ifdef CDKF_W8_PROFILE
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    printf("w8 cycles/obs-step (sizeof real %d, order %d):", (int)sizeof(R), a.order);
    for (int q = 0; q < 10; ++q) printf(" [%d] %lld", q, w8_prof[q] / a.T);
    printf("\n");
    for (int q = 0; q < 10; ++q) w8_prof[q] = 0;
  }
#endif
}

}  // namespace cdkf
