// cdkf_lpe_kernels.h -- Lorenz-63 EKF filter sweep with SIXTEEN LANES PER TRAJECTORY, for batches too small to fill the chip
// with the lane-per-trajectory kernel (4096 trajectories: 64 wavefronts' worth of lanes on 1024 SIMDs).
//
// A lone wavefront pays one 4-cycle issue slot per instruction whatever the lane count, so with SIMDs to spare the sweep
// gets faster only by issuing fewer instructions per observation step.  Here a trajectory occupies one 16-lane DPP row as a
// 4 x 4 grid, lane (i, j) = 4 i + j:
//
//      (i, j), i, j < 3 : P_ij          (i, 3), i < 3 : m_i          (3, j), j < 3 : y_j (observation stream)     (3, 3) : t
//
// * predict: every lane integrates ITS entry -- 20 tableau FMAs per Runge-Kutta step instead of 180, and the right-hand
//   side F P + P F^T + L Qc L^T as two 4-term dot products whose operands arrive by full-rate DPP moves (row_ror by 4 s: the
//   entry s rows below in the same column; quad_perm rotation: s columns to the right in the same row); the Jacobian
//   entries a lane needs, F_{i,(i+s)%4} and F_{j,(j+s)%4}, are affine in the mean with per-lane constants (at most two of
//   the three mean components per slot), the mean itself arrives through row_newbcast operands of the fp64 ALU;
// * update: the twelve moments are broadcast to every lane of the row (row_newbcast), each lane runs the SAME update code as
//   the lane-per-trajectory kernel (ekf_update) redundantly and keeps its own entry -- one source for the arithmetic;
// * stores: lane (i, j) writes its own entry -- ONE store instruction per moment set (mean and covariance together, per-lane
//   pointers) instead of twelve; the row-3 lanes stream the observations and times in, one load per step.
//
// Per observation step ~600 issue slots instead of ~830 (+ 24 stores at ~19 cycles): 4096 x 1000, fp64: 1.77 -> see DESIGN.md.
// Scope: drift Lorenz-63, emission H = I (update inside the lane grid; num_iter 1, symmetric R) or any linear emission with
// m < 3 / iterated updates (per-lane update code), state_order first / second (identical for this drift), fixed-step
// Dormand-Prince, outputs: all four, none, or the filtered pair.  Everything else runs on filter_reg_kernel.
#pragma once
#ifndef __HIPCC_RTC__
#include <cstdlib>
#include <type_traits>

#include "cdkf_launch.h"
#endif
#include "cdkf_reg_kernels.h"

namespace cdkf {

template <int CTRL>
CDKF_DEV double lpe_dpp(double v) {
  const long long old = 0, src = __builtin_bit_cast(long long, v);
  const long long r = __builtin_amdgcn_update_dpp(old, src, CTRL, 0xF, 0xF, true);  // every source lane is valid: no merge with `old`
  return __builtin_bit_cast(double, r);
}
template <int CTRL>
CDKF_DEV float lpe_dpp(float v) {
  const int old = 0, src = __builtin_bit_cast(int, v);
  const int r = __builtin_amdgcn_update_dpp(old, src, CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(float, r);
}
// value of lane L of this 16-lane row (folds into the consuming fp64 instruction as a row_newbcast operand)
template <int L, typename R>
CDKF_DEV R lpe_bcast(R v) { return lpe_dpp<0x150 + L>(v); }

// Right-hand side of the moment ODEs for the entry this lane owns (see the header comment for the grid).
template <typename R>
struct LpeRhs {
  // Slots relative to the lane's row i and column j: F_{q,(q+1)%4} = c1 + gx1 x, F_{q,(q+2)%4} = gy2 y,
  // F_{q,(q+3)%4} = c3 + gz3 z + gx3 x (row / column 3 of the grid is not part of P: all zero).  The mean lanes (column 3)
  // ride on the SAME instructions: the drift is f(m) = M(m) m with M = [[-s, s, 0], [rho, -1, -x], [0, x, -b]], which differs
  // from the Jacobian F only in the (1,0) and (2,0) entries, so a mean lane carries M's constants in its row slots and zeros in
  // its column slots and its slope is the row dot product alone.
  R c1i, gx1i, gy2i, c3i, gz3i, gx3i, c1j, gx1j, gy2j, c3j, gz3j, gx3j;
  R g0, q;  // diagonal slot(s): F_ii + F_jj (covariance lanes) / M_ii (mean lanes); (L Qc L^T)_ij for the covariance lanes
  CDKF_DEV void init(int i, int j, R sigma, R rho, R beta, const R* LQL) {
    auto slot = [&](int r, bool jac, R& c1, R& gx1, R& gy2, R& c3, R& gz3, R& gx3) {
      c1 = (r == 0) ? sigma : R(0);
      gx1 = (r == 1) ? R(-1) : R(0);
      gy2 = (r == 2 && jac) ? R(1) : R(0);  // F_20 = y, M_20 = 0
      c3 = (r == 1) ? rho : R(0);
      gz3 = (r == 1 && jac) ? R(-1) : R(0);  // F_10 = rho - z, M_10 = rho
      gx3 = (r == 2) ? R(1) : R(0);
    };
    const bool cov = i < 3 && j < 3, mean = i < 3 && j == 3;
    slot((cov || mean) ? i : 3, cov, c1i, gx1i, gy2i, c3i, gz3i, gx3i);
    slot(cov ? j : 3, true, c1j, gx1j, gy2j, c3j, gz3j, gx3j);
    const R diag[4] = {-sigma, R(-1), -beta, R(0)};
    g0 = cov ? diag[i] + diag[j] : (mean ? diag[i] : R(0));
    q = cov ? LQL[sidx<3>(i, j)] : R(0);
  }
  CDKF_DEV void operator()(const R (&s)[1], R (&k)[1]) const {
    const R v = s[0];
    const R x = lpe_bcast<3>(v), y = lpe_bcast<7>(v), z = lpe_bcast<11>(v);
    const R d1 = lpe_dpp<0x120 + 12>(v), d2 = lpe_dpp<0x120 + 8>(v), d3 = lpe_dpp<0x120 + 4>(v);  // rows i+1, i+2, i+3
    const R r1 = lpe_dpp<0x39>(v), r2 = lpe_dpp<0x4E>(v), r3 = lpe_dpp<0x93>(v);                  // columns j+1, j+2, j+3
    const R fi1 = rfma(gx1i, x, c1i), fi2 = gy2i * y, fi3 = rfma(gz3i, z, rfma(gx3i, x, c3i));
    const R fj1 = rfma(gx1j, x, c1j), fj2 = gy2j * y, fj3 = rfma(gz3j, z, rfma(gx3j, x, c3j));
    R acc = rfma(g0, v, q);
    acc = rfma(fi1, d1, acc);
    acc = rfma(fi2, d2, acc);
    acc = rfma(fi3, d3, acc);
    acc = rfma(fj1, r1, acc);
    acc = rfma(fj2, r2, acc);
    acc = rfma(fj3, r3, acc);
    k[0] = acc;
  }
};

// Measurement update in the same grid (num_iter = 1, symmetric R): the four rows of a trajectory's grid factorise and solve
// DIFFERENT systems with ONE instruction stream -- rows i < 3: psd_solve's S + 1e-9 I against column i of H P = P (which,
// P being symmetric, is the row's own three covariance lanes: quad broadcasts), i.e. row i of the gain K; row 3: the
// log-likelihood's S against the innovation.  A lane then holds K_i. ; the rows K_j. it needs for P+_ij = P_ij - (K S K^T)_ij
// come from a broadcast of the three gain rows and a two-level select on j.  ~140 instructions instead of the ~270 of
// the redundant per-lane update, the same operations on each number (chol_lower / substitution order of ekf_update).
template <typename R, typename Args>
CDKF_DEV void lpe_update(const Args& a, R& v, R cur, int i, int j, LlAcc& ll, bool& bad) {
  constexpr int D = 3;
  const bool row3 = i == 3, cov = i < 3 && j < 3, mean = i < 3 && j == 3;
  const R Pg[6] = {lpe_bcast<0>(v), lpe_bcast<1>(v), lpe_bcast<2>(v), lpe_bcast<5>(v), lpe_bcast<6>(v), lpe_bcast<10>(v)};
  const R m0 = lpe_bcast<3>(v), m1 = lpe_bcast<7>(v), m2 = lpe_bcast<11>(v);
  const R inn[D] = {lpe_bcast<12>(cur) - m0, lpe_bcast<13>(cur) - m1, lpe_bcast<14>(cur) - m2};
  R S[D][D];
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c < D; ++c) S[r][c] = Pg[sidx<D>(r, c)] + a.Rm[r][c];
  const R eps = row3 ? R(0) : R(1e-9);
  R A[D][D], L[D][D], inv[D];
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) A[r][c] = (r == c) ? S[r][c] + eps : S[r][c];
  chol_lower<R, D>(A, L, inv, bad);
  // right-hand side: row i of P (= column i) for the gain rows, the innovation for row 3
  const R q0 = lpe_dpp<0x00>(v), q1 = lpe_dpp<0x55>(v), q2 = lpe_dpp<0xAA>(v);
  const R b0 = row3 ? inn[0] : q0, b1 = row3 ? inn[1] : q1, b2 = row3 ? inn[2] : q2;
  const R w0 = b0 * inv[0];
  const R w1 = rfma(-L[1][0], w0, b1) * inv[1];
  const R w2 = rfma(-L[2][1], w1, rfma(-L[2][0], w0, b2)) * inv[2];
  const R quad = rfma(w2, w2, rfma(w1, w1, w0 * w0));
  const R pinv = (inv[0] * inv[1]) * inv[2];
  ll.add((double)lpe_bcast<12>(quad), (double)lpe_bcast<12>(pinv), D);
  R x[D];  // K_i.
  x[2] = w2 * inv[2];
  x[1] = rfma(-L[2][1], x[2], w1) * inv[1];
  x[0] = rfma(-L[2][0], x[2], rfma(-L[1][0], x[1], w0)) * inv[0];
  R Kj[D];
#pragma unroll
  for (int c = 0; c < D; ++c) {
    const R k0 = lpe_bcast<0>(x[c]), k1 = lpe_bcast<4>(x[c]), k2 = lpe_bcast<8>(x[c]);
    Kj[c] = (j == 0) ? k0 : ((j == 1) ? k1 : k2);
  }
  R KSi[D], KSj[D];
#pragma unroll
  for (int c = 0; c < D; ++c) {
    KSi[c] = rfma(x[2], S[2][c], rfma(x[1], S[1][c], x[0] * S[0][c]));
    KSj[c] = rfma(Kj[2], S[2][c], rfma(Kj[1], S[1][c], Kj[0] * S[0][c]));
  }
  const R tij = rfma(KSi[2], Kj[2], rfma(KSi[1], Kj[1], KSi[0] * Kj[0]));
  const R tji = rfma(KSj[2], x[2], rfma(KSj[1], x[1], KSj[0] * x[0]));
  const R pn = R(0.5) * ((v - tij) + (v - tji));
  const R mn = rfma(x[2], inn[2], rfma(x[1], inn[1], rfma(x[0], inn[0], v)));
  v = cov ? pn : (mean ? mn : R(0));
  // The lanes below the diagonal adopt the value of their transpose partner (lanes 4 <- 1, 9 <- 6: row_shr:3; 8 <- 2:
  // row_shr:6).  Each lane integrates its own entry, so P_ij and P_ji differ by rounding after a predict, and the
  // antisymmetric part is amplified by the (chaotic) flow -- left alone it reached 1e-8 relative within 300 steps and
  // destroyed the filter within 1000; the packed-symmetric kernels cannot develop it.
  const R s3 = lpe_dpp<0x110 + 3>(v), s6 = lpe_dpp<0x110 + 6>(v);
  v = ((i == 1 && j == 0) || (i == 2 && j == 1)) ? s3 : v;
  v = (i == 2 && j == 0) ? s6 : v;
}

// groups of four trajectories per wavefront; the four wavefronts that share a 128-byte line of the [T,comp,N] arrays sit on
// one XCD (same renumbering as reg_unit_index with xcd_shift = 2 for fp64, 3 for fp32)
template <typename R>
constexpr int lpe_xcd_shift() { return sizeof(R) == 8 ? 2 : 3; }
template <typename R>
inline unsigned lpe_blocks(int64_t N) {
  const int64_t groups = (N + 3) / 4, round = (int64_t)8 << lpe_xcd_shift<R>();
  return (unsigned)((groups + round - 1) / round * round);
}

// OUT: 0 log-likelihood only, 1 all four moment arrays, 2 filtered moments only (the smoother's forward sweep)
// M: emission dimension (1..3).  M == 3 is launched for H = I (HSEL update code, in-grid update when a.lpe_fast); M < 3 takes any
// linear emission through the per-lane update.
template <typename R, int M, int OUT>
__global__ __launch_bounds__(64) void filter_lpe_l63_kernel(const RegArgs<R, 3, M, DriftLorenz63<R, 3>> a) {
  constexpr int D = 3, NS = Dims<D>::NS;
  const int lane = threadIdx.x, l = lane & 15, i = l >> 2, j = l & 3;
  constexpr int sh = lpe_xcd_shift<R>();
  const long b = blockIdx.x;
  const long grp = ((b >> (3 + sh)) << (3 + sh)) + ((b & 7) << sh) + ((b >> 3) & ((1 << sh) - 1));
  if (grp * 4 >= a.N) return;  // surplus wavefront of the rounded-up grid
  const long n_raw = grp * 4 + (lane >> 4);
  const bool live = n_raw < a.N;
  const long n = live ? n_raw : a.N - 1;  // an idle row shadows the last trajectory (same values to the same addresses)
  const bool cov = i < 3 && j < 3, mean = i < 3 && j == 3;

  LpeRhs<R> rhs;
  rhs.init(i, j, a.drift.sigma, a.drift.rho, a.drift.beta, a.LQL);
  const auto C = TabSel<R, false>::get(a);

  // own entry; index of that entry in the gathered state [m_0..m_2, P_00, P_01, P_02, P_11, P_12, P_22] (-1: none)
  const int own = cov ? D + sidx<D>(i, j) : (mean ? i : -1);
  R v = cov ? a.P0[sidx<D>(i, j)] : (mean ? a.m0[i] : R(0));

  // Input stream of the row-3 lanes: (3, j < 3) reads y_j, (3, 3) reads t one row further on (t_{k+1} for step k); the other
  // lanes re-read t_0.  TWO buffers in static registers (the time loop is unrolled by two): a buffer is consumed by the
  // update at the start of its step and reloaded right after it for the step after next, ~1.6 steps (~1.6 us) before it is
  // read again -- a single buffer loaded one step ahead left the load latency partly exposed once a step took ~1 us.
  const R* __restrict__ tp0 = a.t + n * a.t_sn;
  const R* ldbase = tp0;
  long ld_stride = 0, ld_off = 0;
  if (i == 3 && j < M) {
    ldbase = a.y + n * a.y_sn + j * a.y_si;
    ld_stride = a.y_sk;
  } else if (l == 15) {
    ld_stride = a.t_sk;
    ld_off = 1;
  }
  const long last = a.T - 1;
  const R* pA = ldbase + (ld_off < last ? ld_off : last) * ld_stride;
  const R* pB = ldbase + (ld_off + 1 < last ? ld_off + 1 : last) * ld_stride;
  const long ld_stride2 = 2 * ld_stride;
  R tcur = tp0[0];
  R bufA = pA[0], bufB = pB[0];
  // output pointers of this lane (filtered and predicted arrays share the geometry).  The row-3 lanes own no moment; so that
  // the stores stay unconditional (exact vmcnt accounting: behind a branch the compiler drains every store before the next
  // step's load, ~600 cycles per step) they write their zero to ll[n], which lane 0 overwrites after the sweep.
  R* fout = a.ll + n;
  R* pout = a.ll + n;
  long out_stride = 0;
  if constexpr (OUT) {
    if (cov) {
      fout = a.fP + n * a.P_sn + (i * D + j) * a.P_si;
      if constexpr (OUT == 1) pout = a.pP + n * a.P_sn + (i * D + j) * a.P_si;
      out_stride = a.P_sk;
    } else if (mean) {
      fout = a.fm + n * a.m_sn + i * a.m_si;
      if constexpr (OUT == 1) pout = a.pm + n * a.m_sn + i * a.m_si;
      out_stride = a.m_sk;
    }
  }

  LlAcc ll;
  int st = 0;
  bool bad = false;  // a non-positive pivot in this row's factorisations (lpe_update)
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see cdkf_filter_reg_body.inc
  auto step = [&](const long k, R& cur, const R*& ldp) {
    const R tnext_obs = lpe_bcast<15>(cur);
    bool in_grid = false;
    if constexpr (M == 3) {
      if (a.lpe_fast) {
        lpe_update(a, v, cur, i, j, ll, bad);
        in_grid = true;
      }
    }
    if (!in_grid) {
      // measurement update, redundantly in every lane of the row, on the gathered moments (iterated updates, an emission
      // covariance that is not exactly symmetric)
      R ys[NS], yobs[M];
      ys[0] = lpe_bcast<3>(v);
      ys[1] = lpe_bcast<7>(v);
      ys[2] = lpe_bcast<11>(v);
      ys[3] = lpe_bcast<0>(v);
      ys[4] = lpe_bcast<1>(v);
      ys[5] = lpe_bcast<2>(v);
      ys[6] = lpe_bcast<5>(v);
      ys[7] = lpe_bcast<6>(v);
      ys[8] = lpe_bcast<10>(v);
      yobs[0] = lpe_bcast<12>(cur);
      if constexpr (M > 1) yobs[1] = lpe_bcast<13>(cur);
      if constexpr (M > 2) yobs[2] = lpe_bcast<14>(cur);
      ekf_update<R, D, M, (M == 3)>(a, ys, yobs, ll, st);
      if (ys[0] != ys[0]) st |= kStatusNan;
      R upd = R(0);
#pragma unroll
      for (int e = 0; e < NS; ++e) upd = (own == e) ? ys[e] : upd;
      v = upd;
    }
    if (k + 2 + ld_off < a.T) ldp += ld_stride2;
    cur = ldp[0];  // this buffer's next row: y_{k+2} / t_{k+3}
    if constexpr (OUT) *fout = v;

    // predict to t_{k+1} (to t_k + dt_final after the last observation)
    const R t1 = (k + 1 < a.T) ? tnext_obs : tcur + a.dt_final;
    R y1[1] = {v};
    if (integrate<R, 1>(y1, tcur, t1, a.dt0, a.max_steps, rhs, C)) st |= kStatusMaxSteps;
    v = y1[0];
    if constexpr (OUT == 1) {
      *pout = v;
      pout += out_stride;
    }
    if constexpr (OUT) fout += out_stride;
    tcur = tnext_obs;
  };
  long k = 0;
  for (; k + 1 < a.T; k += 2) {  // straight-line pair of steps: the vmcnt accounting stays exact (no branch between the loads)
    step(k, bufA, pA);
    step(k + 1, bufB, pB);
  }
  if (k < a.T) step(k, bufA, pA);
  ll.flush();
  if (a.lpe_fast) {  // flags of the in-grid update: either factorisation (gain rows, log-likelihood row) failed; NaN is sticky
    const R bf = bad ? R(1) : R(0);
    if (lpe_bcast<0>(bf) + lpe_bcast<12>(bf) > R(0)) st |= kStatusNotPd;
    const R m_last = lpe_bcast<3>(v);
    if (m_last != m_last) st |= kStatusNan;
  }
  if (live && l == 0) {
    a.ll[n] = (R)ll.ll;
    if (a.status) a.status[n] = st;
  }
}

// ---- EKF smoother backward sweep in the same grid (extended_kalman_smoother's reverse scan, inference_ekf.py:363-448) -----------
// Per interval the right-hand side is LINEAR with coefficients that are constant over the interval (G = F(m_f) + psd_solve(P_f,
// L Qc L^T)^T and f(m_f) are evaluated at the filtered moments): row i of the grid solves (P_f + 1e-9 I) x = (L Qc L^T)[:, i], which is
// row i of psd_solve(...)^T, so G_i. is local to the row; the rows G_j. a covariance lane also needs come from a broadcast and a
// masked sum.  The slots are then put into the rotated order of the DPP fetches once per interval, and a Runge-Kutta stage is 12
// moves + 7 FMAs (the lane-per-trajectory sweep: ~88 instructions per stage).  Lanes below the diagonal re-adopt their transpose
// partner every step (see lpe_update).  Inputs: the filtered moments the forward sweep just wrote; each lane loads its own entry.
template <typename R>
struct LpeLinRhs {
  R g0, q, ci1, ci2, ci3, cj1, cj2, cj3;
  CDKF_DEV void operator()(const R (&s)[1], R (&k)[1]) const {
    const R v = s[0];
    const R d1 = lpe_dpp<0x120 + 12>(v), d2 = lpe_dpp<0x120 + 8>(v), d3 = lpe_dpp<0x120 + 4>(v);
    const R r1 = lpe_dpp<0x39>(v), r2 = lpe_dpp<0x4E>(v), r3 = lpe_dpp<0x93>(v);
    R acc = rfma(g0, v, q);
    acc = rfma(ci1, d1, acc);
    acc = rfma(ci2, d2, acc);
    acc = rfma(ci3, d3, acc);
    acc = rfma(cj1, r1, acc);
    acc = rfma(cj2, r2, acc);
    acc = rfma(cj3, r3, acc);
    k[0] = acc;
  }
};

template <typename R, int M>
__global__ __launch_bounds__(64) void smoother_lpe_l63_kernel(const RegArgs<R, 3, M, DriftLorenz63<R, 3>> a, R* __restrict__ sm,
                                                              R* __restrict__ sP) {
  constexpr int D = 3;
  const int lane = threadIdx.x, l = lane & 15, i = l >> 2, j = l & 3;
  constexpr int sh = lpe_xcd_shift<R>();
  const long b = blockIdx.x;
  const long grp = ((b >> (3 + sh)) << (3 + sh)) + ((b & 7) << sh) + ((b >> 3) & ((1 << sh) - 1));
  if (grp * 4 >= a.N) return;
  const long n_raw = grp * 4 + (lane >> 4);
  const bool live = n_raw < a.N;
  const long n = live ? n_raw : a.N - 1;
  const bool cov = i < 3 && j < 3, mean = i < 3 && j == 3;
  const auto C = TabSel<R, false>::get(a);
  const R sigma = a.drift.sigma, rho = a.drift.rho, beta = a.drift.beta;

  // per-lane constants: row / column indicator masks (negated: the right-hand side is -(...)), Jacobian row i in absolute columns
  // F_i0 = a0 + bz0 z + by0 y, F_i1 = a1 + bx1 x, F_i2 = a2 + bx2 x, this row's right-hand side of psd_solve, L Qc L^T entry
  const R e0 = (i == 0) ? R(-1) : R(0), e1 = (i == 1) ? R(-1) : R(0), e2 = (i == 2) ? R(-1) : R(0);
  const R f0 = (cov && j == 0) ? R(1) : R(0), f1 = (cov && j == 1) ? R(1) : R(0), f2 = (cov && j == 2) ? R(1) : R(0);
  const R a0 = (i == 0) ? -sigma : ((i == 1) ? rho : R(0)), bz0 = (i == 1) ? R(-1) : R(0), by0 = (i == 2) ? R(1) : R(0);
  const R a1 = (i == 0) ? sigma : ((i == 1) ? R(-1) : R(0)), bx1 = (i == 2) ? R(1) : R(0);
  const R a2 = (i == 2) ? -beta : R(0), bx2 = (i == 1) ? R(-1) : R(0);
  R bq[D];
#pragma unroll
  for (int c = 0; c < D; ++c) bq[c] = (i < 3) ? a.LQL[sidx<D>(c, i < 3 ? i : 0)] : R(0);
  const R qc = cov ? a.LQL[sidx<D>(i, j < 3 ? j : 0)] : R(0);
  const R mmask = mean ? R(1) : R(0);

  // this lane's entry of the filtered moments (input) and of the smoothed moments (output).  The row-3 lanes own no moment:
  // lane 15 streams t in; on the output side all four mirror the row above them (same address, same value), which keeps the
  // stores unconditional without a scratch target
  const int io = (i == 3) ? 2 : i;
  const R* in = (j == 3) ? a.fm + n * a.m_sn + io * a.m_si : a.fP + n * a.P_sn + (io * D + j) * a.P_si;
  R* out = (j == 3) ? sm + n * a.m_sn + io * a.m_si : sP + n * a.P_sn + (io * D + j) * a.P_si;
  const long out_stride = (j == 3) ? a.m_sk : a.P_sk;
  long in_stride = out_stride;
  if (l == 15) {
    in = a.t + n * a.t_sn;
    in_stride = a.t_sk;
  }
  const long last = a.T - 1;
  R v = in[last * in_stride];  // smoothed = filtered at the last time (lane 15: t_{T-1})
  R t1 = lpe_bcast<15>(v);
  if (i == 3) v = R(0);
  {
    const R up = lpe_dpp<0x120 + 4>(v);  // row (i + 3) % 4: for the row-3 lanes the row above
    out[last * out_stride] = (i == 3) ? up : v;
  }
  const long rowA = last - 1 > 0 ? last - 1 : 0, rowB = last - 2 > 0 ? last - 2 : 0;
  const R* pA = in + rowA * in_stride;
  const R* pB = in + rowB * in_stride;
  R bufA = pA[0], bufB = pB[0];
  const long stride2 = 2 * in_stride;
  R* op = out + (last - 1) * out_stride;
  int st = 0;
  bool bad = false;
  __builtin_amdgcn_s_waitcnt(0x0F70);
  auto step = [&](const long k, R& fb, const R*& ldp) {
    const R t0 = lpe_bcast<15>(fb);
    const R mx = lpe_bcast<3>(fb), my = lpe_bcast<7>(fb), mz = lpe_bcast<11>(fb);
    const R Pg[6] = {lpe_bcast<0>(fb), lpe_bcast<1>(fb), lpe_bcast<2>(fb), lpe_bcast<5>(fb), lpe_bcast<6>(fb), lpe_bcast<10>(fb)};
    if (k - 2 >= 0) ldp -= stride2;
    fb = ldp[0];  // this buffer's next row (k - 2), ~1.6 steps before it is read
    R A[D][D], L[D][D], inv[D];
#pragma unroll
    for (int r = 0; r < D; ++r)
#pragma unroll
      for (int c = 0; c <= r; ++c) A[r][c] = (r == c) ? Pg[sidx<D>(r, c)] + R(1e-9) : Pg[sidx<D>(r, c)];
    chol_lower<R, D>(A, L, inv, bad);
    const R w0 = bq[0] * inv[0];
    const R w1 = rfma(-L[1][0], w0, bq[1]) * inv[1];
    const R w2 = rfma(-L[2][1], w1, rfma(-L[2][0], w0, bq[2])) * inv[2];
    R x[D];
    x[2] = w2 * inv[2];
    x[1] = rfma(-L[2][1], x[2], w1) * inv[1];
    x[0] = rfma(-L[2][0], x[2], rfma(-L[1][0], x[1], w0)) * inv[0];
    // G_i. = F_i.(m_f) + x
    const R Gi0 = rfma(by0, my, rfma(bz0, mz, a0)) + x[0], Gi1 = rfma(bx1, mx, a1) + x[1], Gi2 = rfma(bx2, mx, a2) + x[2];
    const R Gi[D] = {Gi0, Gi1, Gi2};
    R Gj[D];
#pragma unroll
    for (int c = 0; c < D; ++c)
      Gj[c] = rfma(f2, lpe_bcast<8>(Gi[c]), rfma(f1, lpe_bcast<4>(Gi[c]), f0 * lpe_bcast<0>(Gi[c])));
    LpeLinRhs<R> rhs;
    // slots in the rotated order of the fetches, negated (e* are -1 on their row): -G_{i,(i+s)%4}, -G_{j,(j+s)%4}
    rhs.ci1 = rfma(e1, Gi[2], e0 * Gi[1]);
    rhs.ci2 = rfma(e2, Gi[0], e0 * Gi[2]);
    rhs.ci3 = rfma(e2, Gi[1], e1 * Gi[0]);
    rhs.cj1 = -rfma(f1, Gj[2], f0 * Gj[1]);
    rhs.cj2 = -rfma(f2, Gj[0], f0 * Gj[2]);
    rhs.cj3 = -rfma(f2, Gj[1], f1 * Gj[0]);
    const R gii = rfma(e2, Gi[2], rfma(e1, Gi[1], e0 * Gi[0]));          // -G_ii
    const R gjj = rfma(f2, Gj[2], rfma(f1, Gj[1], f0 * Gj[0]));          // +G_jj (covariance lanes), 0 elsewhere
    rhs.g0 = gii - gjj;
    // mean lanes: -(f_i(m_f) - sum_c G_ic m_f,c) = x . m_f - corr_i, corr = z x (row 1), -y x (row 2): f = M(m) m, M - F = corr
    const R xm = rfma(x[2], mz, rfma(x[1], my, x[0] * mx));
    const R corr = rfma(e1, mz, -(e2 * my)) * mx;  // e1 = -1 on row 1: -z x; e2 = -1 on row 2: +y x  => this is -corr_i
    rhs.q = rfma(mmask, xm + corr, qc);
    R y1[1] = {v};
    if (integrate<R, 1>(y1, R(0), t1 - t0, a.dt0, a.max_steps, rhs, C)) st |= kStatusMaxSteps;
    v = y1[0];
    const R s3 = lpe_dpp<0x110 + 3>(v), s6 = lpe_dpp<0x110 + 6>(v);
    v = ((i == 1 && j == 0) || (i == 2 && j == 1)) ? s3 : v;
    v = (i == 2 && j == 0) ? s6 : v;
    const R up = lpe_dpp<0x120 + 4>(v);
    *op = (i == 3) ? up : v;
    op -= out_stride;
    t1 = t0;
  };
  long k = a.T - 2;
  for (; k - 1 >= 0; k -= 2) {
    step(k, bufA, pA);
    step(k - 1, bufB, pB);
  }
  if (k >= 0) step(k, bufA, pA);
  if (bad) st |= kStatusNotPd;
  if (live && l == 0 && a.status && st) atomicOr(&a.status[n], st);
}

inline bool lpe_batch_is_small(int64_t N) {
  // 4 trajectories per wavefront: faster than the lane-per-trajectory sweep while every wavefront has a SIMD to itself
  // (MI355X: 1024 SIMDs -> 4096 trajectories: 1.19 against 1.78 ms; 5120: 1.90 against 1.79 ms)
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    cus = 256;
  return (N + 3) / 4 <= 4 * (int64_t)cus;
}

#ifndef __HIPCC_RTC__
// Small Lorenz-63 batches (H = I): sixteen lanes per trajectory (cdkf_lpe_kernels.h).  CDKF_NO_LPE=1 keeps the
// lane-per-trajectory kernel (A/B timing, tests of the latter at small N).
template <typename R, int D, int M, typename Drift>
inline bool try_lpe(const RegArgs<R, D, M, Drift>& a, const cdkf_model* mdl, const cdkf_opts* o, hipStream_t stream) {
  if constexpr (std::is_same<Drift, DriftLorenz63<R, 3>>::value && D == 3 && M <= 3) {
    static const bool off = [] { const char* e = std::getenv("CDKF_NO_LPE"); return e && e[0] == '1'; }();
    const bool all = a.fm && a.fP && a.pm && a.pP, none = !a.fm && !a.fP && !a.pm && !a.pP;
    const bool filt = a.fm && a.fP && !a.pm && !a.pP;
    if (off || !lpe_batch_is_small(a.N) || !(all || none || filt) || (M == 3 && !emission_is_selection(mdl)) || o->forecast ||
        o->state_order == CDKF_ORDER_ZEROTH || o->solver != CDKF_SOLVER_DOPRI5 || o->adaptive)
      return false;
    const dim3 grid(lpe_blocks<R>(a.N)), block(64);
    RegArgs<R, D, M, Drift> b = a;
    bool sym = true;
    for (int r = 0; r < M; ++r)
      for (int c = 0; c < r; ++c) sym = sym && b.Rm[r][c] == b.Rm[c][r];
    b.lpe_fast = (M == 3 && o->num_iter == 1 && sym) ? 1 : 0;
    if (all)
      hipLaunchKernelGGL((filter_lpe_l63_kernel<R, M, 1>), grid, block, 0, stream, b);
    else if (filt)
      hipLaunchKernelGGL((filter_lpe_l63_kernel<R, M, 2>), grid, block, 0, stream, b);
    else
      hipLaunchKernelGGL((filter_lpe_l63_kernel<R, M, 0>), grid, block, 0, stream, b);
    return true;
  } else {
    return false;
  }
}

// backward sweep of the smoother on the same grid (launch_eks.hip, after the forward sweep)
template <typename R, int D, int M, typename Drift>
inline bool try_lpe_smoother(const RegArgs<R, D, M, Drift>& a, const cdkf_opts* o, R* sm, R* sP, hipStream_t stream) {
  if constexpr (std::is_same<Drift, DriftLorenz63<R, 3>>::value && D == 3 && M <= 3) {
    static const bool off = [] { const char* e = std::getenv("CDKF_NO_LPE"); return e && e[0] == '1'; }();
    if (off || !lpe_batch_is_small(a.N) || o->solver != CDKF_SOLVER_DOPRI5 || o->adaptive) return false;
    hipLaunchKernelGGL((smoother_lpe_l63_kernel<R, M>), dim3(lpe_blocks<R>(a.N)), dim3(64), 0, stream, a, sm, sP);
    return true;
  } else {
    return false;
  }
}
#endif

}  // namespace cdkf
