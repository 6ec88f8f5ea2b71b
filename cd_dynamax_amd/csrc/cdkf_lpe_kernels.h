// cdkf_lpe_kernels.h -- EKF / UKF filter sweep for three-dimensional states (Lorenz-63, linear drift) with SIXTEEN LANES PER TRAJECTORY, for batches too small to fill the chip
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

// the same move that leaves the lanes of the banks (= grid rows) outside BANKS with `keep` instead of the fetched value
template <int CTRL, int BANKS>
CDKF_DEV double lpe_dpp_keep(double keep, double v) {
  const long long r = __builtin_amdgcn_update_dpp(__builtin_bit_cast(long long, keep), __builtin_bit_cast(long long, v), CTRL, 0xF,
                                                  BANKS, false);
  return __builtin_bit_cast(double, r);
}
template <int CTRL, int BANKS>
CDKF_DEV float lpe_dpp_keep(float keep, float v) {
  const int r = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep), __builtin_bit_cast(int, v), CTRL, 0xF, BANKS, false);
  return __builtin_bit_cast(float, r);
}

// acc + (lane L of this row's value of src) * mult as ONE instruction: the row_newbcast operand of v_fmac_f64 (the compiler keeps
// a separate v_mov_b64_dpp in front of the 64-bit multiply-add; 32-bit it folds the move by itself).  The statement is opaque to
// the hazard recogniser: `src` must have been written at least two instructions earlier (VALU write -> DPP read, 2 wait states) --
// every caller passes a `mult` that is itself computed from values fetched from `src`, which puts those instructions in between.
template <int L>
CDKF_DEV double lpe_fmac_bcast(double acc, double src, double mult) {
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mult), "n"(L));
  return acc;
}
template <int L>
CDKF_DEV float lpe_fmac_bcast(float acc, float src, float mult) {
  asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mult), "n"(L));
  return acc;
}
// 32-bit: EVERY DPP control folds into the multiply-add (acc + dpp(src) * mult), so the sixteen-lane right-hand side needs no
// separate fetches at all.  Same hazard rule as above; lpe_dpp_fence(v) puts the two wait states behind v's producer.
#define CDKF_LPE_FMAC32(NAME, CTRL)                                                                                       \
  CDKF_DEV float NAME(float acc, float src, float mult) {                                                                 \
    asm volatile("v_fmac_f32_dpp %0, %1, %2 " CTRL " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mult));      \
    return acc;                                                                                                           \
  }
#define CDKF_LPE_MUL32(NAME, CTRL)                                                                       \
  CDKF_DEV float NAME(float src, float mult) {                                                           \
    float out;                                                                                           \
    asm volatile("v_mul_f32_dpp %0, %1, %2 " CTRL " row_mask:0xf bank_mask:0xf" : "=v"(out) : "v"(src), "v"(mult)); \
    return out;                                                                                          \
  }
CDKF_LPE_MUL32(lpe_mul_ror12, "row_ror:12")
CDKF_LPE_MUL32(lpe_mul_ror8, "row_ror:8")
CDKF_LPE_MUL32(lpe_mul_ror4, "row_ror:4")
CDKF_LPE_MUL32(lpe_mul_bcast7, "row_newbcast:7")
#undef CDKF_LPE_MUL32
CDKF_LPE_FMAC32(lpe_fmac_ror12, "row_ror:12")
CDKF_LPE_FMAC32(lpe_fmac_ror8, "row_ror:8")
CDKF_LPE_FMAC32(lpe_fmac_ror4, "row_ror:4")
CDKF_LPE_FMAC32(lpe_fmac_qA, "quad_perm:[1,2,0,3]")
CDKF_LPE_FMAC32(lpe_fmac_qB, "quad_perm:[2,0,1,3]")
CDKF_LPE_FMAC32(lpe_fmac_b1, "row_newbcast:1")
CDKF_LPE_FMAC32(lpe_fmac_b2, "row_newbcast:2")
CDKF_LPE_FMAC32(lpe_fmac_b3, "row_newbcast:3")
CDKF_LPE_FMAC32(lpe_fmac_b11, "row_newbcast:11")
#undef CDKF_LPE_FMAC32
CDKF_DEV float lpe_dpp_fence(float v) {
  asm volatile("s_nop 1" : "+v"(v));
  return v;
}

// Right-hand side of the moment ODEs for the entry this lane owns (see the header comment for the grid).
//
// d P_ij = sum_k F_ik P_kj + sum_k P_ik F_jk + (L Qc L^T)_ij.  The row side fetches the entries of the lane's COLUMN from the rows
// i+1, i+2, i+3 (mod 4; row_ror), the column side the entries of the lane's ROW from the columns (j+1) % 3, (j+2) % 3 -- a quad
// permutation may be any map, so the rotation runs over the three covariance columns only: two fetches instead of three.  The
// Jacobian entries are affine in the mean with per-lane constants; the sum is grouped by mean component,
//      k = [g0 v + q + c1i d1 + c3i d3 + cAj r1 + cBj r2] + x [gx1i d1 + gx3i d3 + gxAj r1 + gxBj r2] + y [gy2i d2 + gyAj r1]
//          + z [gz3i d3 + gzBj r2],
// so that x, y, z enter as the row_newbcast operand of one accumulating instruction each (no separate broadcast moves):
// 16 fp64 instructions + 10 32-bit moves per stage (was 15 + 3 + 12).
// The mean lanes (column 3) ride on the SAME instructions: the drift is f(m) = M(m) m with M = [[-s, s, 0], [rho, -1, -x],
// [0, x, -b]], which differs from the Jacobian F only in the (1,0) and (2,0) entries, so a mean lane carries M's constants in its
// row slots and zeros in its column slots and its slope is the row dot product alone.
//
// UKF = true: the unscented filter's moment equations for THIS drift.  Lorenz-63 is quadratic, f(m + o) = f(m) + F(m) o + b(o, o) with
// b(o, o) = (0, -o_x o_z, o_x o_y), and the sigma points are symmetric about the mean (m, m +- c chol(P)_{:,i}, inference_ukf.py:45-60), so
// the weighted sums of _predict (inference_ukf.py:124-143) collapse exactly, for every alpha, beta, kappa:
//     dm/dt = f_X^T w_mean              = f(m) + 2 w_i c^2 b(P) = f(m) + (0, -P_02, P_01)      (2 w_i c^2 = 1, sum_i o_i o_i^T = c^2 P)
//     f_X^T W X = w_i sum_i (f(X_i+) - f(X_i-)) o_i^T = 2 w_i c^2 F(m) P = F(m) P             (the centre point's term carries X_0 - m = 0)
// i.e. the EKF's covariance equation and a mean equation with the TRUE second-order term (which the reference's EKF 'second' order
// lacks for this drift).  No Cholesky factor and no sigma point is formed: two more multiply-adds on the mean lanes.  What is NOT
// reproduced: the reference turns a trajectory into NaN when a Runge-Kutta STAGE covariance loses positive definiteness
// (jnp.linalg.cholesky); here positive definiteness is tested where the update forms its sigma points (lpe_pd_check), once per step.
template <typename R, bool UKF = false>
struct LpeRhs {
  R g0, q;                   // F_ii + F_jj (covariance lanes) / M_ii (mean lanes); (L Qc L^T)_ij
  R u1, u2;                  // UKF: coefficients of P_01 (lane (2,3): +1) and P_02 (lane (1,3): -1) in the mean equation
  R c1i, c3i, cAj, cBj;      // constant parts: F_{i,i+1}, F_{i,i+3} (mod 4), F_{j,(j+1)%3}, F_{j,(j+2)%3}
  R gx1i, gx3i, gxAj, gxBj;  // their coefficients of x
  R gy2i, gyAj;              // of y (row slot i+2, column slot A)
  R gz3i, gzBj;              // of z (row slot i+3, column slot B)
  CDKF_DEV void init(int i, int j, R sigma, R rho, R beta, const R* LQL) {
    const bool cov = i < 3 && j < 3, mean = i < 3 && j == 3;
    const int r = (cov || mean) ? i : 3, c = cov ? j : 3;
    // row slots of row r: (r, r+1): F01 = s, F12 = -x; (r, r+2): F20 = y (M20 = 0); (r, r+3): F10 = rho - z (M10 = rho), F21 = x
    c1i = (r == 0) ? sigma : R(0);
    gx1i = (r == 1) ? R(-1) : R(0);
    gy2i = (r == 2 && cov) ? R(1) : R(0);
    c3i = (r == 1) ? rho : R(0);
    gz3i = (r == 1 && cov) ? R(-1) : R(0);
    gx3i = (r == 2) ? R(1) : R(0);
    // column slots of column c: A = F_{c,(c+1)%3}: F01 = s, F12 = -x, F20 = y;  B = F_{c,(c+2)%3}: F02 = 0, F10 = rho - z, F21 = x
    cAj = (c == 0) ? sigma : R(0);
    gxAj = (c == 1) ? R(-1) : R(0);
    gyAj = (c == 2) ? R(1) : R(0);
    cBj = (c == 1) ? rho : R(0);
    gzBj = (c == 1) ? R(-1) : R(0);
    gxBj = (c == 2) ? R(1) : R(0);
    const R diag[4] = {-sigma, R(-1), -beta, R(0)};
    g0 = cov ? diag[i] + diag[j] : (mean ? diag[i] : R(0));
    q = cov ? LQL[sidx<3>(i, j)] : R(0);
    u1 = (mean && i == 2) ? R(1) : R(0);
    u2 = (mean && i == 1) ? R(-1) : R(0);
  }
  CDKF_DEV R eval(const R vin) const {
    if constexpr (sizeof(R) == 4) {  // every fetch is the DPP operand of the multiply-add that consumes it: 16 (+2) instructions
      const float v = lpe_dpp_fence(vin);
      // independent chains, interleaved and pinned in this order (volatile): a lone wavefront issues in order, and the assembler's
      // hazard pass puts a wait state between two of these statements that touch the same register within two instructions
      float acc = rfma(g0, v, q);
      float X = lpe_mul_ror12(v, gx1i);
      float Y = lpe_mul_ror8(v, gy2i);
      float Z = lpe_mul_ror4(v, gz3i);
      acc = lpe_fmac_ror12(acc, v, c1i);
      X = lpe_fmac_ror4(X, v, gx3i);
      Y = lpe_fmac_qA(Y, v, gyAj);
      Z = lpe_fmac_qB(Z, v, gzBj);
      acc = lpe_fmac_ror4(acc, v, c3i);
      X = lpe_fmac_qA(X, v, gxAj);
      float B = lpe_mul_bcast7(v, Y);  // second accumulator for the mean-weighted sums: y Y + z Z (+ the unscented terms)
      acc = lpe_fmac_qA(acc, v, cAj);
      X = lpe_fmac_qB(X, v, gxBj);
      B = lpe_fmac_b11(B, v, Z);
      acc = lpe_fmac_qB(acc, v, cBj);
      if constexpr (UKF) B = lpe_fmac_b1(B, v, u1);
      acc = lpe_fmac_b3(acc, v, X);
      if constexpr (UKF) B = lpe_fmac_b2(B, v, u2);
      return acc + B;
    }
    const R v = vin;
    const R d1 = lpe_dpp<0x120 + 12>(v), d2 = lpe_dpp<0x120 + 8>(v), d3 = lpe_dpp<0x120 + 4>(v);  // rows i+1, i+2, i+3
    const R r1 = lpe_dpp<0xC9>(v), r2 = lpe_dpp<0xD2>(v);  // quad_perm [1,2,0,3], [2,0,1,3]: columns (j+1) % 3, (j+2) % 3
    R acc = rfma(g0, v, q);
    acc = rfma(c1i, d1, acc);
    acc = rfma(c3i, d3, acc);
    acc = rfma(cAj, r1, acc);
    acc = rfma(cBj, r2, acc);
    const R X = rfma(gxBj, r2, rfma(gxAj, r1, rfma(gx3i, d3, gx1i * d1)));
    const R Y = rfma(gyAj, r1, gy2i * d2);
    const R Z = rfma(gzBj, r2, gz3i * d3);
    acc = lpe_fmac_bcast<3>(acc, v, X);
    acc = lpe_fmac_bcast<7>(acc, v, Y);
    acc = lpe_fmac_bcast<11>(acc, v, Z);
    if constexpr (UKF) {  // chained on acc: behind the statements above, hence behind v's hazard window
      acc = lpe_fmac_bcast<1>(acc, v, u1);
      acc = lpe_fmac_bcast<2>(acc, v, u2);
    }
    return acc;
  }
  CDKF_DEV void operator()(const R (&s)[1], R (&k)[1]) const { k[0] = eval(s[0]); }
};

// A right-hand side that is LINEAR in the lane's grid with per-lane constant coefficients: the smoother's backward equations
// (coefficients rebuilt per interval) and a linear drift f = W x + b (coefficients fixed for the whole sweep).
template <typename R>
struct LpeLinRhs {
  R g0, q, ci1, ci2, ci3, cjA, cjB;
  CDKF_DEV R eval(const R v) const {
    const R d1 = lpe_dpp<0x120 + 12>(v), d2 = lpe_dpp<0x120 + 8>(v), d3 = lpe_dpp<0x120 + 4>(v);  // rows i+1, i+2, i+3
    const R r1 = lpe_dpp<0xC9>(v), r2 = lpe_dpp<0xD2>(v);                                          // columns (j+1) % 3, (j+2) % 3
    R acc = rfma(g0, v, q);
    acc = rfma(ci1, d1, acc);
    acc = rfma(ci2, d2, acc);
    acc = rfma(ci3, d3, acc);
    acc = rfma(cjA, r1, acc);
    acc = rfma(cjB, r2, acc);
    return acc;
  }
  CDKF_DEV void operator()(const R (&s)[1], R (&k)[1]) const { k[0] = eval(s[0]); }
};

// Right-hand side of the sweep for a drift: Lorenz-63 (affine Jacobian) or linear (constant coefficients: covariance lanes
// d P_ij = sum_k W_ik P_kj + sum_k P_ik W_jk + (L Qc L^T)_ij, mean lanes d m_i = sum_k W_ik m_k + b_i on the same row fetches).  A
// linear drift has no curvature: its unscented moment equations ARE these (the sigma-point sums collapse with b(o, o) = 0).
template <typename R, typename Drift, bool UKF>
struct LpeRhsOf;
template <typename R, bool UKF>
struct LpeRhsOf<R, DriftLorenz63<R, 3>, UKF> {
  using type = LpeRhs<R, UKF>;
  template <typename Args>
  static CDKF_DEV void init(type& rhs, int i, int j, const Args& a) {
    rhs.init(i, j, a.drift.sigma, a.drift.rho, a.drift.beta, a.LQL);
  }
};
template <typename R, bool UKF>
struct LpeRhsOf<R, DriftLinear<R, 3>, UKF> {
  using type = LpeLinRhs<R>;
  template <typename Args>
  static CDKF_DEV void init(type& rhs, int i, int j, const Args& a) {
    const bool cov = i < 3 && j < 3, mean = i < 3 && j == 3;
    auto Wm = [&](int r, int c) { return (r < 3 && c < 3) ? a.drift.W[r][c] : R(0); };
    const int r = (cov || mean) ? i : 3, c = cov ? j : 3;
    rhs.ci1 = Wm(r, (r + 1) & 3);
    rhs.ci2 = Wm(r, (r + 2) & 3);
    rhs.ci3 = Wm(r, (r + 3) & 3);
    rhs.cjA = (c < 3) ? Wm(c, (c + 1) % 3) : R(0);
    rhs.cjB = (c < 3) ? Wm(c, (c + 2) % 3) : R(0);
    rhs.g0 = cov ? Wm(i, i) + Wm(j, j) : (mean ? Wm(i, i) : R(0));
    rhs.q = cov ? a.LQL[sidx<3>(i, j)] : (mean ? a.drift.b[i] : R(0));
  }
};

// The unscented update draws its sigma points from chol(P) (inference_ukf.py:184, 57): a covariance that is not positive definite
// makes the reference's factor -- and from there the whole trajectory -- NaN.  Same test without the factor: the three leading
// principal minors (Sylvester), from broadcasts of the grid; every lane of the row gets the same answer.
template <typename R>
CDKF_DEV bool lpe_pd_check(const R v) {
  const R p00 = lpe_bcast<0>(v), p01 = lpe_bcast<1>(v), p02 = lpe_bcast<2>(v), p11 = lpe_bcast<5>(v), p12 = lpe_bcast<6>(v),
          p22 = lpe_bcast<10>(v);
  const R m2 = rfma(p00, p11, -(p01 * p01));
  const R c0 = rfma(p11, p22, -(p12 * p12)), c1 = rfma(p01, p22, -(p12 * p02)), c2 = rfma(p01, p12, -(p11 * p02));
  const R m3 = rfma(p02, c2, rfma(p00, c0, -(p01 * c1)));
  return (p00 > R(0)) && (m2 > R(0)) && (m3 > R(0));
}

// One Dormand-Prince step of the lane's entry.  fp64: slopes scaled by the step once (k_s = dt f_s: six products) and combined
// with the tableau constants from registers -- 26 instructions beside the right-hand sides; dopri5_step's fp64 form (c_sj = dt a_sj
// formed per step) is laid out for nine entries per lane and costs 40 for one.  fp32 keeps dopri5_step's association, which is the
// reference's (`y0 + dt (a_lower[i] @ ks)`).
template <typename R, typename Rhs>
CDKF_DEV void lpe_step(R& y, R dt, const Rhs& rhs, const Dp5V<R>& C) {
  if constexpr (sizeof(R) == 8) {
    const R k1 = dt * rhs.eval(y);
    const R k2 = dt * rhs.eval(rfma(C.a21, k1, y));
    const R k3 = dt * rhs.eval(rfma(C.a32, k2, rfma(C.a31, k1, y)));
    const R k4 = dt * rhs.eval(rfma(C.a43, k3, rfma(C.a42, k2, rfma(C.a41, k1, y))));
    const R k5 = dt * rhs.eval(rfma(C.a54, k4, rfma(C.a53, k3, rfma(C.a52, k2, rfma(C.a51, k1, y)))));
    const R k6 = dt * rhs.eval(rfma(C.a65, k5, rfma(C.a64, k4, rfma(C.a63, k3, rfma(C.a62, k2, rfma(C.a61, k1, y))))));
    y = rfma(C.b6, k6, rfma(C.b5, k5, rfma(C.b4, k4, rfma(C.b3, k3, rfma(C.b1, k1, y)))));
  } else {
    R ys[1] = {y};
    dopri5_step<R, 1>(ys, dt, rhs, C);
    y = ys[0];
  }
}

// diffrax's loop (see integrate in cdkf_math.h) around lpe_step; true when max_steps was hit
template <typename R, typename Rhs>
CDKF_DEV bool lpe_integrate(R& y, R t0, R t1, R dt0, long max_steps, const Rhs& rhs, const Dp5V<R>& C) {
  R tprev = t0;
  R tnext = rmin(t0 + dt0, t1);
  long steps = 0;
  bool capped = false;
  while (tprev < t1) {
    if (steps >= max_steps) {
      capped = true;
      break;
    }
    lpe_step<R>(y, tnext - tprev, rhs, C);
    tprev = rmin(tnext, t1);
    const R tn = tnext + dt0;
    tnext = (tn > t1 - Tol<R>::v) ? t1 : tn;
    ++steps;
  }
  return capped;
}

// 1 / sqrt(x) for the factorisations of the in-grid update: v_rsq_f64 and one third-order correction (the library routine's own
// arithmetic without its special-casing of 0 and +inf; a non-positive pivot still gives NaN, which is what is wanted of it).
CDKF_DEV double lpe_rsqrt(double x) {
  const double y0 = __builtin_amdgcn_rsq(x);
  const double e = rfma(-(x * y0), y0, 1.0);
  return rfma(y0 * e, rfma(e, 0.375, 0.5), y0);
}
CDKF_DEV float lpe_rsqrt(float x) { return rrsqrt(x); }

// Measurement update in the same grid (num_iter = 1, symmetric R): the four rows of a trajectory's grid factorise and solve
// DIFFERENT systems with ONE instruction stream -- rows i < 3: psd_solve's S + 1e-9 I against column i of H P = P (which,
// P being symmetric, is the row's own three covariance lanes: quad broadcasts), i.e. row i of the gain K; row 3: the
// log-likelihood's S against the innovation (the quad broadcast writes banks 0..2 only and leaves bank 3 the innovation it held).
// A lane then holds K_i. and (K S)_i. ; what it subtracts from its entry is (K S)_i. K_j.^T for a covariance lane and -K_i. (y - m)
// for a mean lane.  K_j. belongs to another row: the three candidates j = 0, 1, 2 are formed with the gain rows as row_newbcast
// operands and the lane picks its own.  Lanes on and above the diagonal keep P_ij - (K S K^T)_ij as computed (the reference
// averages it with the (j, i) value, which differs by rounding only), lanes below adopt their transpose partner's value.
// Every lane accumulates the log-likelihood terms of ITS row's system; the row-3 lanes hold the trajectory's.
template <typename R, typename Args>
CDKF_DEV void lpe_update(const Args& a, R& v, R cur, int i, int j, LlAcc& ll, bool& bad) {
  constexpr int D = 3;
  const bool row3 = i == 3;
  const R Pg[6] = {lpe_bcast<0>(v), lpe_bcast<1>(v), lpe_bcast<2>(v), lpe_bcast<5>(v), lpe_bcast<6>(v), lpe_bcast<10>(v)};
  R S[D][D];
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c < D; ++c) S[r][c] = Pg[sidx<D>(r, c)] + a.Rm[r][c];
  // innovation y_c - m_c: the observation's broadcast, then one multiply-add with the mean's broadcast as its operand.  neg1 is
  // tied to a value the compiler fetched from v with a move of its own, which keeps the statements behind v's hazard window.
  R neg1 = R(-1);
  asm volatile("" : "+v"(neg1) : "v"(Pg[5]));
  R b0 = lpe_fmac_bcast<3>(lpe_bcast<12>(cur), v, neg1);
  R b1 = lpe_fmac_bcast<7>(lpe_bcast<13>(cur), v, neg1);
  R b2 = lpe_fmac_bcast<11>(lpe_bcast<14>(cur), v, neg1);
  // right-hand sides: row i of P (= column i) for the gain rows; the innovation stays where it is in row 3
  b0 = lpe_dpp_keep<0x00, 0x7>(b0, v);
  b1 = lpe_dpp_keep<0x55, 0x7>(b1, v);
  b2 = lpe_dpp_keep<0xAA, 0x7>(b2, v);
  const R eps = row3 ? R(0) : R(1e-9);
  // Cholesky factor of S (+ eps I), chol_lower's operations with lpe_rsqrt
  const R d0 = S[0][0] + eps;
  bad = bad || !(d0 > R(0));
  const R i0 = lpe_rsqrt(d0);
  const R L10 = S[1][0] * i0, L20 = S[2][0] * i0;
  const R d1 = rfma(-L10, L10, S[1][1] + eps);
  bad = bad || !(d1 > R(0));
  const R i1 = lpe_rsqrt(d1);
  const R L21 = rfma(-L20, L10, S[2][1]) * i1;
  const R d2 = rfma(-L21, L21, rfma(-L20, L20, S[2][2] + eps));
  bad = bad || !(d2 > R(0));
  const R i2 = lpe_rsqrt(d2);
  const R w0 = b0 * i0;
  const R w1 = rfma(-L10, w0, b1) * i1;
  const R w2 = rfma(-L21, w1, rfma(-L20, w0, b2)) * i2;
  const R quad = rfma(w2, w2, rfma(w1, w1, w0 * w0));
  const R pinv = (i0 * i1) * i2;
  ll.add((double)quad, (double)pinv, D);
  R x[D];  // K_i.
  x[2] = w2 * i2;
  x[1] = rfma(-L21, x[2], w1) * i1;
  x[0] = rfma(-L20, x[2], rfma(-L10, x[1], w0)) * i0;
  R KS[D];  // -(K S)_i.
#pragma unroll
  for (int c = 0; c < D; ++c) KS[c] = rfma(-x[2], S[2][c], rfma(-x[1], S[1][c], -x[0] * S[0][c]));
  // candidates P_ij - (K S)_i. K_j.^T for j = 0, 1, 2 (gain row j = lanes 4 j .. of this grid) and m_i + K_i. (y - m) for the mean
  // lanes (the innovation: row 3's right-hand sides)
  R n0 = v, n1 = v, n2 = v, n3 = v;
#pragma unroll
  for (int c = 0; c < D; ++c) {
    n0 = lpe_fmac_bcast<0>(n0, x[c], KS[c]);
    n1 = lpe_fmac_bcast<4>(n1, x[c], KS[c]);
    n2 = lpe_fmac_bcast<8>(n2, x[c], KS[c]);
  }
  n3 = lpe_fmac_bcast<12>(n3, b0, x[0]);
  n3 = lpe_fmac_bcast<12>(n3, b1, x[1]);
  n3 = lpe_fmac_bcast<12>(n3, b2, x[2]);
  const R nv = (j == 0) ? n0 : ((j == 1) ? n1 : ((j == 2) ? n2 : n3));
  v = row3 ? R(0) : nv;
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
// M: emission dimension (1..3).  M == 3 is launched for H = I (HSEL update code); FAST: the in-grid update (lpe_update: one update
// iteration, symmetric R); otherwise -- M < 3 with any linear emission, iterated updates -- the per-lane update.
// UKF: the unscented filter (LpeRhs<R, true>; its update with a linear emission is the same algebra as the EKF's -- the sigma points
// pass through h(x) = x exactly: pred_mean = m, pred_cov = P + R, pred_cross = P, inference_ukf.py:186-203 -- plus the
// positive-definiteness test the reference's Cholesky of P implies).
template <typename R, typename Drift, int M, int OUT, bool FAST, bool UKF = false>
__global__ __launch_bounds__(64) void filter_lpe_kernel(const RegArgs<R, 3, M, Drift> a) {
  static_assert(!FAST || M == 3, "the in-grid update is written for H = I");
  static_assert(!UKF || FAST, "the unscented filter runs with the in-grid update");
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

  typename LpeRhsOf<R, Drift, UKF>::type rhs;
  LpeRhsOf<R, Drift, UKF>::init(rhs, i, j, a);
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
  // step's load, ~600 cycles per step) they write their zero to ll[n], which is overwritten after the sweep.
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
  // One observation step.  TAIL = false: a step of the main loop, at least four rows before the end of the arrays -- the
  // reload's row exists for every lane and the predict ends at the next observation, so neither is tested.
  auto step = [&](auto tail, const long k, R& cur, const R*& ldp) {
    constexpr bool TAIL = decltype(tail)::value;
    const R tnext_obs = lpe_bcast<15>(cur);
    if constexpr (UKF) {
      if (!lpe_pd_check(v)) {  // chol(P) of the update's sigma points fails: NaN from here on, as in the reference
        bad = true;
        v = (i < 3) ? R(__builtin_nan("")) : v;
      }
    }
    if constexpr (FAST) {
      lpe_update(a, v, cur, i, j, ll, bad);
    } else {
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
    if (!TAIL || k + 2 + ld_off < a.T) ldp += ld_stride2;
    cur = ldp[0];  // this buffer's next row: y_{k+2} / t_{k+3}
    if constexpr (OUT) *fout = v;

    // predict to t_{k+1} (to t_k + dt_final after the last observation)
    const R t1 = (!TAIL || k + 1 < a.T) ? tnext_obs : tcur + a.dt_final;
    if (lpe_integrate<R>(v, tcur, t1, a.dt0, a.max_steps, rhs, C)) st |= kStatusMaxSteps;
    if constexpr (OUT == 1) {
      *pout = v;
      pout += out_stride;
    }
    if constexpr (OUT) fout += out_stride;
    tcur = tnext_obs;
  };
  // straight-line pairs of steps: the vmcnt accounting stays exact (no branch between the loads)
  long k = 0;
  for (; k + 5 <= a.T; k += 2) {  // rows k + 3 and k + 4 exist
    step(std::false_type{}, k, bufA, pA);
    step(std::false_type{}, k + 1, bufB, pB);
  }
  for (; k + 1 < a.T; k += 2) {
    step(std::true_type{}, k, bufA, pA);
    step(std::true_type{}, k + 1, bufB, pB);
  }
  if (k < a.T) step(std::true_type{}, k, bufA, pA);
  ll.flush();
  if constexpr (FAST) {  // flags of the in-grid update: either factorisation (gain rows, log-likelihood row) failed; NaN is sticky
    const R bf = bad ? R(1) : R(0);
    if (lpe_bcast<0>(bf) + lpe_bcast<12>(bf) > R(0)) st |= kStatusNotPd;
    const R m_last = lpe_bcast<3>(v);
    if (m_last != m_last) st |= kStatusNan;
  }
  // the trajectory's log-likelihood: every lane's with the per-lane update, the row-3 lanes' with the in-grid update
  if (live && l == (FAST ? 12 : 0)) {
    a.ll[n] = (R)ll.ll;
    if (a.status) a.status[n] = st;
  }
}

// ---- EKF smoother backward sweep in the same grid (extended_kalman_smoother's reverse scan, inference_ekf.py:363-448) -----------
// Per interval the right-hand side is LINEAR with coefficients that are constant over the interval (G = F(m_f) + psd_solve(P_f,
// L Qc L^T)^T and f(m_f) are evaluated at the filtered moments): row i of the grid solves (P_f + 1e-9 I) x = (L Qc L^T)[:, i], which is
// row i of psd_solve(...)^T, so G_i. is local to the row; the rows G_j. a covariance lane also needs come from a broadcast and a
// masked sum.  The slots are then put into the rotated order of the DPP fetches once per interval, and a Runge-Kutta stage is 10
// moves + 6 FMAs (the lane-per-trajectory sweep: ~88 instructions per stage).  Lanes below the diagonal re-adopt their transpose
// partner every step (see lpe_update).  Inputs: the filtered moments the forward sweep just wrote; each lane loads its own entry.
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
    rhs.cjA = -rfma(f2, Gj[0], rfma(f1, Gj[2], f0 * Gj[1]));  // -G_{j,(j+1)%3}
    rhs.cjB = -rfma(f2, Gj[1], rfma(f1, Gj[0], f0 * Gj[2]));  // -G_{j,(j+2)%3}
    const R gii = rfma(e2, Gi[2], rfma(e1, Gi[1], e0 * Gi[0]));          // -G_ii
    const R gjj = rfma(f2, Gj[2], rfma(f1, Gj[1], f0 * Gj[0]));          // +G_jj (covariance lanes), 0 elsewhere
    rhs.g0 = gii - gjj;
    // mean lanes: -(f_i(m_f) - sum_c G_ic m_f,c) = x . m_f - corr_i, corr = z x (row 1), -y x (row 2): f = M(m) m, M - F = corr
    const R xm = rfma(x[2], mz, rfma(x[1], my, x[0] * mx));
    const R corr = rfma(e1, mz, -(e2 * my)) * mx;  // e1 = -1 on row 1: -z x; e2 = -1 on row 2: +y x  => this is -corr_i
    rhs.q = rfma(mmask, xm + corr, qc);
    if (lpe_integrate<R>(v, R(0), t1 - t0, a.dt0, a.max_steps, rhs, C)) st |= kStatusMaxSteps;
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

// Which batches take the sixteen-lane grid (4 trajectories per wavefront) instead of the lane-per-trajectory sweep.  `per_simd`:
// wavefronts per SIMD up to which the grid is the faster one for that sweep (measured: profiles/r03_*_n_sweep.json).
// CDKF_LPE_MAX_N=<n> overrides the threshold (scripts/n_sweep_table.py: 0 = never, a huge value = always).
enum LpeSweep { kLpeFilter = 0, kLpeUkf = 1, kLpeSmoother = 2, kLpeGrad = 3 };
inline bool lpe_batch_is_small(int64_t N, LpeSweep which = kLpeFilter) {
  if (const char* e = std::getenv("CDKF_LPE_MAX_N")) return N <= std::atoll(e);
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    cus = 256;
  // measured on MI355X, 4096 ... 131072 trajectories x 1000 steps (profiles/r03_a_n_sweep.json): EKF filter grid 1.70 / lane 1.89 ms at
  // 8192, 2.43 / 2.00 at 12288; unscented filter (fp32; the lane kernel forms the sigma points) 3.59 / 3.69 at 24576, 4.60 / 3.86 at
  // 32768; smoother 2.78 / 3.59 at 8192, 3.98 / 3.63 at 12288; drift-block gradient 2.34 / 4.02 at 4096, 5.07 / 4.10 at 5120
  static constexpr int per_simd[4] = {2, 6, 2, 1};
  return (N + 3) / 4 <= (int64_t)per_simd[which] * 4 * cus;
}

#ifndef __HIPCC_RTC__
// Small Lorenz-63 batches (H = I): sixteen lanes per trajectory (cdkf_lpe_kernels.h).  CDKF_NO_LPE=1 keeps the
// lane-per-trajectory kernel (A/B timing, tests of the latter at small N).
template <typename R, int D, int M, typename Drift>
inline bool try_lpe(const RegArgs<R, D, M, Drift>& a, const cdkf_model* mdl, const cdkf_opts* o, hipStream_t stream, bool ukf = false,
                    bool any_batch = false) {  // any_batch: also beyond one wavefront per SIMD (the reverse sweep's forward pass)
  if constexpr ((std::is_same<Drift, DriftLorenz63<R, 3>>::value || std::is_same<Drift, DriftLinear<R, 3>>::value) && D == 3 && M <= 3) {
    static const bool off = [] { const char* e = std::getenv("CDKF_NO_LPE"); return e && e[0] == '1'; }();
    // opts.flags & CDKF_FLAG_UKF_SIGMA_POINTS (or CDKF_UKF_SIGMA_POINTS=1 in the environment): the unscented filter on the
    // lane-per-trajectory kernel, which forms the sigma points and factorises every stage covariance as the reference does
    static const bool ukf_env = [] { const char* e = std::getenv("CDKF_UKF_SIGMA_POINTS"); return e && e[0] == '1'; }();
    const bool ukf_off = ukf_env || (o->flags & CDKF_FLAG_UKF_SIGMA_POINTS);
    const bool all = a.fm && a.fP && a.pm && a.pP, none = !a.fm && !a.fP && !a.pm && !a.pP;
    const bool filt = a.fm && a.fP && !a.pm && !a.pP;
    if (off || (!any_batch && !lpe_batch_is_small(a.N, ukf ? kLpeUkf : kLpeFilter)) || !(all || none || filt) || (M == 3 && !emission_is_selection(mdl)) || o->forecast ||
        o->solver != CDKF_SOLVER_DOPRI5 || o->adaptive)
      return false;
    if (!ukf && o->state_order == CDKF_ORDER_ZEROTH) return false;
    const dim3 grid(lpe_blocks<R>(a.N)), block(64);
    bool sym = true;
    for (int r = 0; r < M; ++r)
      for (int c = 0; c < r; ++c) sym = sym && a.Rm[r][c] == a.Rm[c][r];
    const bool fast = M == 3 && (ukf || o->num_iter == 1) && sym;  // the in-grid update (lpe_update)
    if (ukf && (ukf_off || !fast || !(a.ukf_c > R(0)))) return false;  // (n + lambda <= 0: the reference's NaNs come from the other kernel)
    auto launch = [&](auto out) {
      constexpr int OUT = decltype(out)::value;
      note_kernel("filter_lpe_kernel<%s, cdkf::%s<%s, 3>, %d, %d, %s, %s>", real_name<R>(),
                  std::is_same<Drift, DriftLinear<R, 3>>::value ? "DriftLinear" : "DriftLorenz63", real_name<R>(), M, OUT,
                  fast ? "true" : "false", ukf ? "true" : "false");
      if constexpr (M == 3) {
        if (ukf) {
          hipLaunchKernelGGL((filter_lpe_kernel<R, Drift, M, OUT, true, true>), grid, block, 0, stream, a);
          return;
        }
        if (fast) {
          hipLaunchKernelGGL((filter_lpe_kernel<R, Drift, M, OUT, true, false>), grid, block, 0, stream, a);
          return;
        }
      }
      hipLaunchKernelGGL((filter_lpe_kernel<R, Drift, M, OUT, false, false>), grid, block, 0, stream, a);
    };
    if (all)
      launch(std::integral_constant<int, 1>{});
    else if (filt)
      launch(std::integral_constant<int, 2>{});
    else
      launch(std::integral_constant<int, 0>{});
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
    if (off || !lpe_batch_is_small(a.N, kLpeSmoother) || o->solver != CDKF_SOLVER_DOPRI5 || o->adaptive) return false;
    note_kernel("smoother_lpe_l63_kernel<%s, %d>", real_name<R>(), M);
    hipLaunchKernelGGL((smoother_lpe_l63_kernel<R, M>), dim3(lpe_blocks<R>(a.N)), dim3(64), 0, stream, a, sm, sP);
    return true;
  } else {
    return false;
  }
}
#endif

}  // namespace cdkf
