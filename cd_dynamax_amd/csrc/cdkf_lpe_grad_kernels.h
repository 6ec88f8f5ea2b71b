// cdkf_lpe_grad_kernels.h -- reverse sweep of the Lorenz-63 log-likelihood gradient (d ll / d (sigma, rho, beta), the drift block
// of jax.value_and_grad(_loss_fn), ssm_temissions.py:550-568) on the sixteen-lanes-per-trajectory grid of cdkf_lpe_kernels.h.
//
// The forward-sensitivity kernel (ekf_grad_reg_kernel) carries a lane per (trajectory, parameter): 4096 x 3 lanes are 192
// wavefronts on 1024 SIMDs.  Here the forward sweep is filter_lpe_kernel itself (all four moment arrays into the reverse sweep's
// workspace) and the backward sweep keeps the same grid -- lane (i, j) of a 16-lane row owns the cotangent of P_ij, column 3 the
// cotangent of the mean:
//
// * predict, reversed: the interval is re-integrated from the filtered moments at k-1 (the start of every Runge-Kutta step is parked
//   in LDS, one value per lane); per step the six stage values are replayed and the stages reversed.  The cotangent of a slope
//   (lam, Lam) gives  Pbar = F^T Lam + Lam F,  xbar = F^T lam + dF/dm : 2 Lam P,  thetabar += df/dtheta . lam + dF/dtheta : 2 Lam P.
//   The first two terms are the forward right-hand side with F transposed: the SAME fetch pattern (column entries from the rows
//   i+1, i+2, i+3; row entries from the columns (j+1) % 3, (j+2) % 3) with the per-lane constants of F^T.  F is affine in the mean
//   with five non-constant entries, so the contractions with 2 Lam P need only a handful of products Lam_kj P_lj: a covariance lane
//   forms the one its ROW contributes from the row fetches it already holds and the three lanes of the row add up in the row's mean
//   lane (quad permutations); the parameter sums stay distributed over the lanes -- row 0 collects
//   d/d sigma, row 1 d/d rho, row 2 d/d beta -- and are added up once, after the sweep.
// * update, reversed (lpe_update_adj): the update's reverse in Joseph form, the rows of the grid carrying different rows of the
//   3 x 3 matrices with one instruction stream, as lpe_update does for the forward update;
// * ALL: the model block as well -- m0, P0 (what is left after the first update), L Qc L^T (the sum of the slope cotangents), H, bias,
//   R (from the reverse update's intermediate cotangents) -- for cdkf_ekf_loglik_grad_all_*.
//
// What is reversed: _predict's moment equations (inference_ekf.py:76-123) under diffeqsolve's fixed-step Dopri5 (diffrax_utils.py:40-165)
// and _condition_on + the log-likelihood term (inference_ekf.py:153-199, 285-286), for H = I, symmetric R, num_iter 1 -- the numbers JAX's
// reverse mode gives for the discretised recursion (the tests hold them to a NumPy restatement of that adjoint, itself pinned by finite
// differences).  grad(div f) = 0 for this drift, so state_order 'first' and 'second' coincide.
#pragma once
#include "cdkf_lpe_kernels.h"

namespace cdkf {

// Step starts of one observation interval, in LDS (one value per lane and step): a WINDOW of kLpeGradWin consecutive starts that the
// reversed steps read, and kLpeGradCoarse COARSE starts, one per segment of C steps (C = the window while the interval has at most
// kLpeGradWin * kLpeGradCoarse = 768 steps, else ceil(steps / kLpeGradCoarse)).  Walking backwards the window is refilled from the
// segment's coarse start: about 2 S forward steps for an interval of S <= 768 steps, S^2 / 1536 beyond (a gap of 100 time units at
// dt0 = 0.01: 6.5 S) -- never the (S - window)^2 / 2 of re-integrating from one kept start for every reversed step.
#ifndef CDKF_LPE_GRAD_WIN
#define CDKF_LPE_GRAD_WIN 48
#endif
constexpr int kLpeGradWin = CDKF_LPE_GRAD_WIN, kLpeGradCoarse = 16;

// per-lane constants of the reversed right-hand side
template <typename R>
struct LpeAdjRhs {
  R g0;                      // F_ii + F_jj (covariance lanes) / F_ii (mean lanes)
  R c1, c3, cA, cB;          // constant parts of F_{i+1,i}, F_{i+3,i} (rows mod 4), F_{(j+1)%3,j}, F_{(j+2)%3,j}
  R gx1, gx3, gxA, gxB;      // their coefficients of x
  R gy2, gyB;                // of y
  R gz1, gzA;                // of z
  R k12, k13, k21, k32;      // the row's share of dF/dm : 2 Lam P (see stage())
  R t0, t1, t3;              // the lane's share of the parameter contraction
  R mm;                      // 1 on the mean lanes
  CDKF_DEV void init(int i, int j, R sigma, R rho, R beta) {
    const bool cov = i < 3 && j < 3, mean = i < 3 && j == 3;
    const int r = (cov || mean) ? i : 3, c = cov ? j : 3;
    // F = [[-s, s, 0], [rho - z, -1, -x], [y, x, -b]].  Row slots of row r hold F_{(r+s)%4, r}:
    //   r = 0: F10 = rho - z, F20 = y;  r = 1: F21 = x, F01 = s (slot 3);  r = 2: F12 = -x (slot 3)
    c1 = (r == 0) ? rho : R(0);
    gz1 = (r == 0) ? R(-1) : R(0);
    gx1 = (r == 1) ? R(1) : R(0);
    gy2 = (r == 0) ? R(1) : R(0);
    c3 = (r == 1) ? sigma : R(0);
    gx3 = (r == 2) ? R(-1) : R(0);
    // column slots of column c: A = F_{(c+1)%3, c}: F10 = rho - z, F21 = x, F02 = 0;  B = F_{(c+2)%3, c}: F20 = y, F01 = s, F12 = -x
    cA = (c == 0) ? rho : R(0);
    gzA = (c == 0) ? R(-1) : R(0);
    gxA = (c == 1) ? R(1) : R(0);
    gyB = (c == 0) ? R(1) : R(0);
    cB = (c == 1) ? sigma : R(0);
    gxB = (c == 2) ? R(-1) : R(0);
    const R diag[4] = {-sigma, R(-1), -beta, R(0)};
    g0 = cov ? diag[i] + diag[j] : (mean ? diag[i] : R(0));
    // dF/dx : G = G21 - G12, dF/dy : G = G20, dF/dz : G = -G10 with G = 2 Lam P, G_ab = 2 sum_j Lam_aj P_bj:
    //   row 0 forms 2 (Lam_2j P_1j - Lam_1j P_2j), row 1  2 Lam_2j P_0j, row 2  -2 Lam_1j P_0j  (column j = the lane's)
    k12 = (cov && i == 0) ? R(-2) : R(0);
    k21 = (cov && i == 0) ? R(2) : R(0);
    k13 = (cov && i == 1) ? R(2) : R(0);
    k32 = (cov && i == 2) ? R(-2) : R(0);
    // d/d sigma: lam_0 (y - x) + G01 - G00 (row 0); d/d rho: lam_1 x + G10 (row 1); d/d beta: -lam_2 z - G22 (row 2)
    const R w = cov ? R(2) : (mean ? R(1) : R(0));
    t0 = (r == 0 || r == 2) ? -w : R(0);
    t1 = (r == 0) ? w : R(0);
    t3 = (r == 1) ? w : R(0);
    mm = mean ? R(1) : R(0);
  }
  // cotangent of the stage VALUE (this lane's entry) from the cotangent L of the stage SLOPE; Ys: the stage value, p1..p3 its entries
  // in the rows i+1, i+2, i+3 (the fetches the replayed forward right-hand side made).  th accumulates the lane's share of the
  // parameter gradient.
  template <bool ALL>
  CDKF_DEV R stage(const R L, const R Ys, const R p1, const R p2, const R p3, R& th, R& lql) const {
    if constexpr (ALL) lql += L;  // cotangent of L Qc L^T: the slope of P carries it as a constant term
    const R d1 = lpe_dpp<0x120 + 12>(L), d2 = lpe_dpp<0x120 + 8>(L), d3 = lpe_dpp<0x120 + 4>(L);  // rows i+1, i+2, i+3
    const R r1 = lpe_dpp<0xC9>(L), r2 = lpe_dpp<0xD2>(L);                                          // columns (j+1) % 3, (j+2) % 3
    R acc = g0 * L;
    acc = rfma(c1, d1, acc);
    acc = rfma(c3, d3, acc);
    acc = rfma(cA, r1, acc);
    acc = rfma(cB, r2, acc);
    const R X = rfma(gxB, r2, rfma(gxA, r1, rfma(gx3, d3, gx1 * d1)));
    const R Y = rfma(gyB, r2, gy2 * d2);
    const R Z = rfma(gzA, r1, gz1 * d1);
    // the row's three products (zero on the mean lane), added up over the quad and kept by the mean lane (mm: 1 there, 0 elsewhere)
    const R msg = rfma(d3, k32 * p2, rfma(d2, k21 * p1, d1 * rfma(k13, p3, k12 * p2)));
    const R h = msg + lpe_dpp<0xB1>(msg);  // quad_perm [1,0,3,2]
    const R q4 = h + lpe_dpp<0x4E>(h);     // quad_perm [2,3,0,1]
    acc = lpe_fmac_bcast<3>(acc, Ys, X);   // (Ys was written by the replay, long before: see lpe_fmac_bcast)
    acc = lpe_fmac_bcast<7>(acc, Ys, Y);
    acc = lpe_fmac_bcast<11>(acc, Ys, Z);
    acc = rfma(mm, q4, acc);
    th = rfma(L, rfma(t3, p3, rfma(t1, p1, t0 * Ys)), th);
    return acc;
  }
};

// The forward right-hand side (LpeRhs::eval) that also hands out its row fetches of v
template <typename R>
CDKF_DEV R lpe_eval_keep(const LpeRhs<R, false>& c, const R v, R& d1, R& d2, R& d3) {
  d1 = lpe_dpp<0x120 + 12>(v);
  d2 = lpe_dpp<0x120 + 8>(v);
  d3 = lpe_dpp<0x120 + 4>(v);
  if constexpr (sizeof(R) == 4) {
    return c.eval(v);  // (32-bit: the fetches are operands of the multiply-adds there)
  } else {
    const R r1 = lpe_dpp<0xC9>(v), r2 = lpe_dpp<0xD2>(v);
    R acc = rfma(c.g0, v, c.q);
    acc = rfma(c.c1i, d1, acc);
    acc = rfma(c.c3i, d3, acc);
    acc = rfma(c.cAj, r1, acc);
    acc = rfma(c.cBj, r2, acc);
    const R X = rfma(c.gxBj, r2, rfma(c.gxAj, r1, rfma(c.gx3i, d3, c.gx1i * d1)));
    const R Y = rfma(c.gyAj, r1, c.gy2i * d2);
    const R Z = rfma(c.gzBj, r2, c.gz3i * d3);
    acc = lpe_fmac_bcast<3>(acc, v, X);
    acc = lpe_fmac_bcast<7>(acc, v, Y);
    acc = lpe_fmac_bcast<11>(acc, v, Z);
    return acc;
  }
}

// One Dormand-Prince step from y over dt, reversed: vb (cotangent of the step's result) becomes the cotangent of y.
template <bool ALL, typename R>
CDKF_DEV void lpe_step_adj(const LpeRhs<R, false>& rhs, const LpeAdjRhs<R>& adj, const Dp5V<R>& C, const R y, const R dt, R& vb,
                           R& th, R& lql) {
  R p[6][3];  // row fetches of the six stage values
  const R k1 = dt * lpe_eval_keep(rhs, y, p[0][0], p[0][1], p[0][2]);
  const R Y2 = rfma(C.a21, k1, y);
  const R k2 = dt * lpe_eval_keep(rhs, Y2, p[1][0], p[1][1], p[1][2]);
  const R Y3 = rfma(C.a32, k2, rfma(C.a31, k1, y));
  const R k3 = dt * lpe_eval_keep(rhs, Y3, p[2][0], p[2][1], p[2][2]);
  const R Y4 = rfma(C.a43, k3, rfma(C.a42, k2, rfma(C.a41, k1, y)));
  const R k4 = dt * lpe_eval_keep(rhs, Y4, p[3][0], p[3][1], p[3][2]);
  const R Y5 = rfma(C.a54, k4, rfma(C.a53, k3, rfma(C.a52, k2, rfma(C.a51, k1, y))));
  const R k5 = dt * lpe_eval_keep(rhs, Y5, p[4][0], p[4][1], p[4][2]);
  R Y6 = rfma(C.a65, k5, rfma(C.a64, k4, rfma(C.a63, k3, rfma(C.a62, k2, rfma(C.a61, k1, y)))));
  // Y6 is a row_newbcast operand of the first stage reversed, whose other inputs do not depend on it: two wait states by hand (the
  // earlier stage values are behind whole right-hand sides)
  asm volatile("s_nop 1" : "+v"(Y6));
  p[5][0] = lpe_dpp<0x120 + 12>(Y6);
  p[5][1] = lpe_dpp<0x120 + 8>(Y6);
  p[5][2] = lpe_dpp<0x120 + 4>(Y6);
  // cotangent of slope s: dt (b_s vb + sum_{q > s} a_qs Yb_q)
  const R Yb6 = adj.template stage<ALL>(dt * (C.b6 * vb), Y6, p[5][0], p[5][1], p[5][2], th, lql);
  const R Yb5 = adj.template stage<ALL>(dt * rfma(C.a65, Yb6, C.b5 * vb), Y5, p[4][0], p[4][1], p[4][2], th, lql);
  const R Yb4 = adj.template stage<ALL>(dt * rfma(C.a64, Yb6, rfma(C.a54, Yb5, C.b4 * vb)), Y4, p[3][0], p[3][1], p[3][2], th, lql);
  const R Yb3 = adj.template stage<ALL>(dt * rfma(C.a63, Yb6, rfma(C.a53, Yb5, rfma(C.a43, Yb4, C.b3 * vb))), Y3, p[2][0], p[2][1], p[2][2], th, lql);
  const R Yb2 =
      adj.template stage<ALL>(dt * rfma(C.a62, Yb6, rfma(C.a52, Yb5, rfma(C.a42, Yb4, C.a32 * Yb3))), Y2, p[1][0], p[1][1], p[1][2], th, lql);
  const R Yb1 = adj.template stage<ALL>(dt * rfma(C.a61, Yb6, rfma(C.a51, Yb5, rfma(C.a41, Yb4, rfma(C.a31, Yb3, rfma(C.a21, Yb2, C.b1 * vb))))), y,
                          p[0][0], p[0][1], p[0][2], th, lql);
  vb += ((Yb1 + Yb2) + (Yb3 + Yb4)) + (Yb5 + Yb6);
}

// two wait states behind the producers of a, b, c: what a row_newbcast read of them by lpe_fmac_bcast needs (see there)
template <typename R>
CDKF_DEV void lpe_fence3(R& a, R& b, R& c) {
  asm volatile("s_nop 1" : "+v"(a), "+v"(b), "+v"(c));
}
// (j == 0) ? a0 : (j == 1) ? a1 : a2
template <typename R>
CDKF_DEV R lpe_sel3(int j, R a0, R a1, R a2) {
  return (j == 0) ? a0 : ((j == 1) ? a1 : a2);
}

// The measurement update at one observation, reversed (H = I, symmetric R, one iteration): vp = the grid of the PREDICTED moments
// with the observation in row 3, vb = the cotangents of the filtered moments; returns this lane's cotangent of the predicted
// moments (the log-likelihood term of the observation included).
//
// The reverse of the update (cotangents vbar, Kbar, Sbar, Ubar of the innovation, the gain, S and the solve) collapses, for H = I, to the
// Joseph form: with
// A = (S + eps I)^-1, X = A P (the transposed gain), J = I - X = A (R + eps I), u = A v, w = S^-1 v and S A = I - eps A,
//     Pbar <- J Pbar J^T + sym((J mbar) u^T) + 2 eps sym(J Pbar X^T A) + w w^T / 2 - S^-1 / 2,      mbar <- J mbar + w
// -- the same numbers to rounding (the eps term is kept: it is 1e-9 of the first, which is the tolerance of the parity tests;
// S^-1 = A + eps A^2 up to eps^2 = 1e-18).  As in lpe_update the rows of the grid work on DIFFERENT rows of these matrices with one
// instruction stream: row i solves (S + eps I) x = e_i (one factorisation, per-lane right-hand sides) and carries row i of A, J,
// J Pbar, ...; the rows of another grid row arrive as row_newbcast operands, entries of the lane's own column by selection, and the
// transposed entries of the two non-symmetric terms by the fetches of lpe_update's symmetrisation in both directions.
// accumulators of the model block (ALL): this lane's entry of the cotangents of R, H (covariance lanes) and of the emission bias (row i's,
// in every lane of the row)
template <typename R>
struct LpeModelAcc {
  R r = R(0), h = R(0), b = R(0), lql = R(0);
};
// lanes 4 <-> 1, 9 <-> 6 (three apart), 8 <-> 2 (six apart) exchange values; every other lane keeps its own
template <typename R>
CDKF_DEV R lpe_transpose(const R Z, const int l) {
  const R up3 = lpe_dpp<0x110 + 3>(Z), up6 = lpe_dpp<0x110 + 6>(Z);  // row_shr: from the lane 3 / 6 below
  const R dn3 = lpe_dpp<0x100 + 3>(Z), dn6 = lpe_dpp<0x100 + 6>(Z);  // row_shl: from the lane 3 / 6 above
  R Zt = Z;
  Zt = (l == 4 || l == 9) ? up3 : Zt;
  Zt = (l == 8) ? up6 : Zt;
  Zt = (l == 1 || l == 6) ? dn3 : Zt;
  Zt = (l == 2) ? dn6 : Zt;
  return Zt;
}

template <bool ALL, typename R, typename Args>
CDKF_DEV R lpe_update_adj(const Args& a, R vp, R vb, const int i, const int j, LpeModelAcc<R>& acc) {
  const int l = 4 * i + j;
  constexpr R eps = R(1e-9);
  R neg1 = R(-1);
  lpe_fence3(vp, vb, neg1);  // vp, vb are row_newbcast operands below
  const R Pg[6] = {lpe_bcast<0>(vp), lpe_bcast<1>(vp), lpe_bcast<2>(vp), lpe_bcast<5>(vp), lpe_bcast<6>(vp), lpe_bcast<10>(vp)};
  R S[3][3], v[3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) S[r][c] = Pg[sidx<3>(r, c)] + a.Rm[r][c];
  // innovation y - m: the observation's broadcast, then one multiply-add with the mean's broadcast as its operand
  v[0] = lpe_fmac_bcast<3>(lpe_bcast<12>(vp), vp, neg1);
  v[1] = lpe_fmac_bcast<7>(lpe_bcast<13>(vp), vp, neg1);
  v[2] = lpe_fmac_bcast<11>(lpe_bcast<14>(vp), vp, neg1);
  // Cholesky factor of S + eps I (chol_lower's operations with lpe_rsqrt), then row i of its inverse
  const R i0 = lpe_rsqrt(S[0][0] + eps);
  const R L10 = S[1][0] * i0, L20 = S[2][0] * i0;
  const R i1 = lpe_rsqrt(rfma(-L10, L10, S[1][1] + eps));
  const R L21 = rfma(-L20, L10, S[2][1]) * i1;
  const R i2 = lpe_rsqrt(rfma(-L21, L21, rfma(-L20, L20, S[2][2] + eps)));
  const R b0 = (i == 0) ? R(1) : R(0), b1 = (i == 1) ? R(1) : R(0), b2 = (i == 2) ? R(1) : R(0);
  const R f0 = b0 * i0;
  const R f1 = rfma(-L10, f0, b1) * i1;
  const R f2 = rfma(-L21, f1, rfma(-L20, f0, b2)) * i2;
  R Ai[3];
  Ai[2] = f2 * i2;
  Ai[1] = rfma(-L21, Ai[2], f1) * i1;
  Ai[0] = rfma(-L20, Ai[2], rfma(-L10, Ai[1], f0)) * i0;
  lpe_fence3(Ai[0], Ai[1], Ai[2]);  // (row_newbcast operands below)
  // row i of J = A (R + eps I), J mbar, J Pbar (the cotangents as broadcast operands: lane 4 c + k holds Pbar_ck, 4 c + 3 mbar_c)
  R Ji[3], JPi[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    Ji[k] = rfma(Ai[2], (k == 2) ? a.Rm[2][k] + eps : a.Rm[2][k],
                 rfma(Ai[1], (k == 1) ? a.Rm[1][k] + eps : a.Rm[1][k], Ai[0] * ((k == 0) ? a.Rm[0][k] + eps : a.Rm[0][k])));
  }
  R ui = rfma(Ai[2], v[2], rfma(Ai[1], v[1], Ai[0] * v[0]));
  R Jmi = lpe_bcast<3>(vb) * Ji[0];
  Jmi = lpe_fmac_bcast<7>(Jmi, vb, Ji[1]);
  Jmi = lpe_fmac_bcast<11>(Jmi, vb, Ji[2]);
  JPi[0] = lpe_bcast<0>(vb) * Ji[0];
  JPi[1] = lpe_bcast<1>(vb) * Ji[0];
  JPi[2] = lpe_bcast<2>(vb) * Ji[0];
  JPi[0] = lpe_fmac_bcast<4>(JPi[0], vb, Ji[1]);
  JPi[1] = lpe_fmac_bcast<5>(JPi[1], vb, Ji[1]);
  JPi[2] = lpe_fmac_bcast<6>(JPi[2], vb, Ji[1]);
  JPi[0] = lpe_fmac_bcast<8>(JPi[0], vb, Ji[2]);
  JPi[1] = lpe_fmac_bcast<9>(JPi[1], vb, Ji[2]);
  JPi[2] = lpe_fmac_bcast<10>(JPi[2], vb, Ji[2]);
  // row i of S^-1 = A + eps A^2 and of J Pbar J^T: the rows c = 0, 1, 2 of A / J from the lanes 4 c
  lpe_fence3(Ji[0], Ji[1], Ji[2]);
  const R eA0 = eps * Ai[0], eA1 = eps * Ai[1], eA2 = eps * Ai[2];
  R Si[3], n[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) Si[c] = Ai[c];
  n[0] = lpe_bcast<0>(Ji[0]) * JPi[0];
  n[1] = lpe_bcast<4>(Ji[0]) * JPi[0];
  n[2] = lpe_bcast<8>(Ji[0]) * JPi[0];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    Si[k] = lpe_fmac_bcast<0>(Si[k], Ai[k], eA0);
    Si[k] = lpe_fmac_bcast<4>(Si[k], Ai[k], eA1);
    Si[k] = lpe_fmac_bcast<8>(Si[k], Ai[k], eA2);
    if (k > 0) {
      n[0] = lpe_fmac_bcast<0>(n[0], Ji[k], JPi[k]);
      n[1] = lpe_fmac_bcast<4>(n[1], Ji[k], JPi[k]);
      n[2] = lpe_fmac_bcast<8>(n[2], Ji[k], JPi[k]);
    }
  }
  R wi = rfma(Si[2], v[2], rfma(Si[1], v[1], Si[0] * v[0]));
  // E = J Pbar X^T = J Pbar - J Pbar J^T (row i), eps W = eps E A
  const R E0 = eps * (JPi[0] - n[0]), E1 = eps * (JPi[1] - n[1]), E2 = eps * (JPi[2] - n[2]);
  R hJm = R(0.5) * Jmi, hw = R(0.5) * wi;
  lpe_fence3(ui, wi, hJm);
  // candidates for the lane's column k:  Z_ik = (J mbar)_i u_k / 2 + eps W_ik  and  Q_ik = n_ik + (w_i w_k - (S^-1)_ik) / 2;
  // u_k, w_k belong to grid row k
  R Zc[3], Qc[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    Zc[k] = lpe_bcast<0>(Ai[k]) * E0;
    Zc[k] = lpe_fmac_bcast<4>(Zc[k], Ai[k], E1);
    Zc[k] = lpe_fmac_bcast<8>(Zc[k], Ai[k], E2);
    Qc[k] = rfma(R(-0.5), Si[k], n[k]);
  }
  Zc[0] = lpe_fmac_bcast<0>(Zc[0], ui, hJm);
  Zc[1] = lpe_fmac_bcast<4>(Zc[1], ui, hJm);
  Zc[2] = lpe_fmac_bcast<8>(Zc[2], ui, hJm);
  Qc[0] = lpe_fmac_bcast<0>(Qc[0], wi, hw);
  Qc[1] = lpe_fmac_bcast<4>(Qc[1], wi, hw);
  Qc[2] = lpe_fmac_bcast<8>(Qc[2], wi, hw);
  const R Z = lpe_sel3(j, Zc[0], Zc[1], Zc[2]), Q = lpe_sel3(j, Qc[0], Qc[1], Qc[2]);
  const R Zt = lpe_transpose(Z, l);
  const R pn = Q + (Z + Zt);
  if constexpr (ALL) {
    // The intermediate cotangents in these terms: vbar = mbar - J mbar - w;  Ubar = u mbar^T - 2 X Pbar + 2 eps A X Pbar
    // (X Pbar = Pbar - J Pbar);  Sbar = Pbar_new - Pbar - sym(Ubar^T).  Then  Rbar += Sbar,  biasbar -= vbar,
    // Hbar += (2 Sbar + Ubar) P - vbar m^T   (H = I: H P = P).
    const R mbi = lpe_dpp<0xFF>(vb);  // quad_perm [3,3,3,3]: the row's mean lane
    const R vbar = (mbi - Jmi) - wi;
    R XP[3] = {lpe_dpp<0x00>(vb) - JPi[0], lpe_dpp<0x55>(vb) - JPi[1], lpe_dpp<0xAA>(vb) - JPi[2]};  // row i of Pbar, minus J Pbar's
    lpe_fence3(XP[0], XP[1], XP[2]);
    R Ub[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      R axp = lpe_bcast<0>(XP[k]) * Ai[0];
      axp = lpe_fmac_bcast<4>(axp, XP[k], Ai[1]);
      axp = lpe_fmac_bcast<8>(axp, XP[k], Ai[2]);
      Ub[k] = rfma(R(2) * eps, axp, R(-2) * XP[k]);
    }
    Ub[0] = lpe_fmac_bcast<3>(Ub[0], vb, ui);
    Ub[1] = lpe_fmac_bcast<7>(Ub[1], vb, ui);
    Ub[2] = lpe_fmac_bcast<11>(Ub[2], vb, ui);
    const R Uij = lpe_sel3(j, Ub[0], Ub[1], Ub[2]);
    const R Uji = lpe_transpose(Uij, l);
    const R Sbar = (pn - vb) - R(0.5) * (Uij + Uji);  // (covariance lanes; the others are not stored)
    acc.r += Sbar;
    acc.b -= vbar;
    // row i of 2 Sbar + Ubar, times P (every lane holds P), minus vbar m^T: the candidates for the columns 0, 1, 2
    const R T0 = rfma(R(2), lpe_dpp<0x00>(Sbar), Ub[0]), T1 = rfma(R(2), lpe_dpp<0x55>(Sbar), Ub[1]),
            T2 = rfma(R(2), lpe_dpp<0xAA>(Sbar), Ub[2]);
    R nv = -vbar;
    lpe_fence3(nv, vp, vb);
    R Hc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) Hc[c] = rfma(T2, Pg[sidx<3>(2, c)], rfma(T1, Pg[sidx<3>(1, c)], T0 * Pg[sidx<3>(0, c)]));
    Hc[0] = lpe_fmac_bcast<3>(Hc[0], vp, nv);
    Hc[1] = lpe_fmac_bcast<7>(Hc[1], vp, nv);
    Hc[2] = lpe_fmac_bcast<11>(Hc[2], vp, nv);
    acc.h += lpe_sel3(j, Hc[0], Hc[1], Hc[2]);
  }
  return (j == 3) ? Jmi + wi : pn;
}

// grad [N, 3].  The forward sweep (filter_lpe_kernel, OUT = 1) has written fm, fP, pm, pP with the strides of `a`.
// inverse of a general M x M matrix, M <= 3 (adjugate / determinant)
template <typename R, int M>
CDKF_DEV void lpe_invm(const R (&A)[M][M], R (&inv)[M][M]) {
  if constexpr (M == 1) {
    inv[0][0] = R(1) / A[0][0];
  } else if constexpr (M == 2) {
    const R id = R(1) / rfma(A[0][0], A[1][1], -(A[0][1] * A[1][0]));
    inv[0][0] = A[1][1] * id;
    inv[0][1] = -A[0][1] * id;
    inv[1][0] = -A[1][0] * id;
    inv[1][1] = A[0][0] * id;
  } else {
    const R c00 = rfma(A[1][1], A[2][2], -(A[1][2] * A[2][1]));
    const R c01 = rfma(A[1][2], A[2][0], -(A[1][0] * A[2][2]));
    const R c02 = rfma(A[1][0], A[2][1], -(A[1][1] * A[2][0]));
    const R id = R(1) / rfma(A[0][2], c02, rfma(A[0][1], c01, A[0][0] * c00));
    inv[0][0] = c00 * id;
    inv[1][0] = c01 * id;
    inv[2][0] = c02 * id;
    inv[0][1] = rfma(A[0][2], A[2][1], -(A[0][1] * A[2][2])) * id;
    inv[1][1] = rfma(A[0][0], A[2][2], -(A[0][2] * A[2][0])) * id;
    inv[2][1] = rfma(A[0][1], A[2][0], -(A[0][0] * A[2][1])) * id;
    inv[0][2] = rfma(A[0][1], A[1][2], -(A[0][2] * A[1][1])) * id;
    inv[1][2] = rfma(A[0][2], A[1][0], -(A[0][0] * A[1][2])) * id;
    inv[2][2] = rfma(A[0][0], A[1][1], -(A[0][1] * A[1][0])) * id;
  }
}

// The reverse update for the emissions the in-grid form does not cover -- H = I[:M] with M < 3 (the forward sweep then runs
// filter_lpe_kernel<..., FAST = false>; R symmetric as there): every lane gathers the twelve moments, the observation and the
// cotangents and evaluates the reverse of _condition_on + the log-likelihood term (inference_ekf.py:153-199, 285-286) redundantly, in the
// literal order of the update: S = H P H^T + R, v = y - H m, w = S^-1 v, X = (sym(S) + eps I)^-1 H P (the transposed gain), then
//     vbar = X mbar - w,  Kbar = v mbar^T - 2 S X Pbar,  Ubar = (sym(S) + eps I)^-1 Kbar,
//     Sbar = -X Pbar X^T + w w^T / 2 - S^-1 / 2 - sym(X Ubar^T),
//     Pbar <- Pbar + sym(Ubar^T H) + H^T Sbar H,  mbar <- mbar - H^T vbar;   Rbar += Sbar,  biasbar -= vbar,  Hbar += 2 Sbar H P - vbar m^T + Ubar P.
template <bool ALL, int M, typename R, typename Args>
CDKF_DEV R lpe_update_adj_gen(const Args& a, const R vp, const R vb, const int i, const int j, LpeModelAcc<R>& acc) {
  const R Pg[6] = {lpe_bcast<0>(vp), lpe_bcast<1>(vp), lpe_bcast<2>(vp), lpe_bcast<5>(vp), lpe_bcast<6>(vp), lpe_bcast<10>(vp)};
  const R m[3] = {lpe_bcast<3>(vp), lpe_bcast<7>(vp), lpe_bcast<11>(vp)};
  const R yo[3] = {lpe_bcast<12>(vp), lpe_bcast<13>(vp), lpe_bcast<14>(vp)};
  const R B[9] = {lpe_bcast<0>(vb), lpe_bcast<1>(vb), lpe_bcast<2>(vb), lpe_bcast<4>(vb), lpe_bcast<5>(vb),
                  lpe_bcast<6>(vb), lpe_bcast<8>(vb), lpe_bcast<9>(vb), lpe_bcast<10>(vb)};
  const R mb[3] = {lpe_bcast<3>(vb), lpe_bcast<7>(vb), lpe_bcast<11>(vb)};
  R P[3][3], Pb[3][3], S[M][M], Sb[M][M], Sinv[M][M], Sbinv[M][M], v[M], w[M], vbar[M];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      P[r][c] = Pg[sidx<3>(r, c)];
      Pb[r][c] = R(0.5) * (B[3 * r + c] + B[3 * c + r]);
    }
#pragma unroll
  for (int r = 0; r < M; ++r) {
#pragma unroll
    for (int c = 0; c < M; ++c) S[r][c] = P[r][c] + a.Rm[r][c];
    v[r] = yo[r] - m[r];
  }
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int c = 0; c < M; ++c) Sb[r][c] = R(0.5) * (S[r][c] + S[c][r]) + ((r == c) ? R(1e-9) : R(0));
  lpe_invm<R, M>(S, Sinv);
  lpe_invm<R, M>(Sb, Sbinv);
  R X[M][3], SX[M][3], Kb[M][3], Ub[M][3], XP[M][3], Sbar[M][M];
#pragma unroll
  for (int r = 0; r < M; ++r) {
    R acc_w = R(0), acc_v = R(0);
#pragma unroll
    for (int c = 0; c < M; ++c) acc_w = rfma(Sinv[r][c], v[c], acc_w);
    w[r] = acc_w;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      R x = R(0);
#pragma unroll
      for (int k = 0; k < M; ++k) x = rfma(Sbinv[r][k], P[k][c], x);  // H P = the first M rows of P
      X[r][c] = x;
      acc_v = rfma(x, mb[c], acc_v);
    }
    vbar[r] = acc_v - w[r];
  }
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      R sx = R(0), xp = R(0);
#pragma unroll
      for (int k = 0; k < M; ++k) sx = rfma(S[r][k], X[k][c], sx);
#pragma unroll
      for (int k = 0; k < 3; ++k) xp = rfma(X[r][k], Pb[k][c], xp);
      SX[r][c] = sx;
      XP[r][c] = xp;
    }
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      R t1 = R(0);
#pragma unroll
      for (int k = 0; k < 3; ++k) t1 = rfma(SX[r][k], Pb[k][c], t1);
      Kb[r][c] = rfma(R(-2), t1, v[r] * mb[c]);
    }
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      R u = R(0);
#pragma unroll
      for (int k = 0; k < M; ++k) u = rfma(Sbinv[r][k], Kb[k][c], u);
      Ub[r][c] = u;
    }
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int c = 0; c < M; ++c) {
      R q = R(0), xu = R(0), ux = R(0);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        q = rfma(XP[r][k], X[c][k], q);
        xu = rfma(X[r][k], Ub[c][k], xu);
        ux = rfma(X[c][k], Ub[r][k], ux);
      }
      Sbar[r][c] = rfma(R(-0.5), xu + ux, rfma(R(0.5), rfma(w[r], w[c], -Sinv[r][c]), -q));
    }
  R out = R(0);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      R pn = Pb[r][c];
      if (c < M) pn = rfma(R(0.5), Ub[c < M ? c : 0][r], pn);  // sym(Ubar^T H): (Ubar^T H)_rc = Ubar_cr for c < M
      if (r < M) pn = rfma(R(0.5), Ub[r < M ? r : 0][c], pn);
      if (r < M && c < M) pn += Sbar[r < M ? r : 0][c < M ? c : 0];
      out = (i == r && j == c) ? pn : out;
    }
    out = (i == r && j == 3) ? ((r < M) ? mb[r] - vbar[r < M ? r : 0] : mb[r]) : out;
  }
  if constexpr (ALL) {
#pragma unroll
    for (int r = 0; r < M; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        R h = -(vbar[r] * m[c]);
#pragma unroll
        for (int k = 0; k < M; ++k) h = rfma(R(2) * Sbar[r][k], P[k][c], h);
#pragma unroll
        for (int k = 0; k < 3; ++k) h = rfma(Ub[r][k], P[k][c], h);
        acc.h += (i == r && j == c) ? h : R(0);
        if (c < M) acc.r += (i == r && j == c) ? Sbar[r][c < M ? c : 0] : R(0);
      }
      acc.b -= (i == r) ? vbar[r] : R(0);
    }
  }
  return out;
}

// M: emission dimension (H = I[:M], symmetric R).  GRID: the reverse update in the grid (lpe_update_adj: M = 3); otherwise per lane.
// ALL: also grad_model [N, 21 + 3 M + M + M^2] = m0 [3] | P0 [3,3] | LQL [3,3] | H [M,3] | h_bias [M] | R [M,M] (include/cdkf.h,
// cdkf_ekf_loglik_grad_all_*)
template <typename R, int M, bool GRID, bool ALL>
__global__ __launch_bounds__(64) void grad_lpe_l63_kernel(const RegArgs<R, 3, M, DriftLorenz63<R, 3>> a, R* __restrict__ grad,
                                                          R* __restrict__ grad_model) {
  static_assert(!GRID || M == 3, "the in-grid reverse update is written for H = I");
  constexpr int D = 3;
  __shared__ R starts[kLpeGradWin][64];
  __shared__ R coarse[kLpeGradCoarse][64];
  const int lane = threadIdx.x, l = lane & 15, i = l >> 2, j = l & 3;
  constexpr int sh = lpe_xcd_shift<R>();
  const long b = blockIdx.x;
  const long grp = ((b >> (3 + sh)) << (3 + sh)) + ((b & 7) << sh) + ((b >> 3) & ((1 << sh) - 1));
  if (grp * 4 >= a.N) return;
  const long n_raw = grp * 4 + (lane >> 4);
  const bool live = n_raw < a.N;
  const long n = live ? n_raw : a.N - 1;
  const bool cov = i < 3 && j < 3, mean = i < 3 && j == 3;

  LpeRhs<R, false> rhs;
  rhs.init(i, j, a.drift.sigma, a.drift.rho, a.drift.beta, a.LQL);
  LpeAdjRhs<R> adj;
  adj.init(i, j, a.drift.sigma, a.drift.rho, a.drift.beta);
  const auto C = TabSel<R, false>::get(a);

  // per-lane input streams, read backwards with running pointers.  `pp`: the predicted moments at observation k (array row k - 1),
  // y_k (row 3), t_k (lane 15); `fp`: the filtered moments at k - 1, t_{k-1} (lane 15).  Both are loaded a whole step ahead.
  const R* pp;
  const R* fp;
  long stride;
  const long last = a.T - 1, lastm = last > 0 ? last - 1 : 0;
  if (cov) {
    stride = a.P_sk;
    pp = a.pP + n * a.P_sn + (i * D + j) * a.P_si + lastm * stride;
    fp = a.fP + n * a.P_sn + (i * D + j) * a.P_si + lastm * stride;
  } else if (mean) {
    stride = a.m_sk;
    pp = a.pm + n * a.m_sn + i * a.m_si + lastm * stride;
    fp = a.fm + n * a.m_sn + i * a.m_si + lastm * stride;
  } else if (j < 3) {  // (a lane without an observation component mirrors component 0)
    stride = a.y_sk;
    pp = a.y + n * a.y_sn + (j < M ? j : 0) * a.y_si + last * stride;
    fp = pp;
  } else {
    stride = a.t_sk;
    pp = a.t + n * a.t_sn + last * stride;
    fp = a.t + n * a.t_sn + lastm * stride;
  }
  const long stride3 = (i == 3) ? stride : 0;
  const R own0 = cov ? a.P0[sidx<D>(i, j < 3 ? j : 0)] : (mean ? a.m0[i < 3 ? i : 0] : R(0));
  auto advance = [&](R& tp, R& tq, const R t1) {  // the fixed-step loop's next interval (lpe_integrate)
    tp = rmin(tq, t1);
    const R tn = tq + a.dt0;
    tq = (tn > t1 - Tol<R>::v) ? t1 : tn;
  };

  R vb = R(0), th = R(0);
  LpeModelAcc<R> acc;
  R pv = pp[0], fv = fp[0];
  for (long k = last; k >= 0; --k) {
    R vp = pv;
    const R vf = fv;
    if (k >= 1) {  // the rows of observation k - 1 (array rows k - 2 of the moments: none for k = 1, the prior is an argument)
      pp -= (k >= 2) ? stride : stride3;
      if (k >= 2) fp -= stride;
      pv = pp[0];
      fv = fp[0];
    }
    if (k == 0 && i < 3) vp = own0;
    if constexpr (GRID)
      vb = lpe_update_adj<ALL, R>(a, vp, vb, i, j, acc);
    else
      vb = lpe_update_adj_gen<ALL, M, R>(a, vp, vb, i, j, acc);
    if (k == 0) break;
    // ---- the predict from k-1 to k, reversed ----
    const R t1 = lpe_bcast<15>(vp), t0 = lpe_bcast<15>(vf);
    const R y0 = (i == 3) ? R(0) : vf;
    R tp = t0, tq = rmin(t0 + a.dt0, t1);
    const R dt_0 = (t0 < t1) ? tq - tp : R(0);  // (no step where the forward sweep took none: t_k <= t_{k-1})
    advance(tp, tq, t1);
    // more than one Runge-Kutta step in this interval: the starts of the steps 1, 2, ... (lanes of grid row 3 carry no state: lane 15's
    // slot holds the step size / the time the step starts at, lane 14's the end of that step) are parked in LDS on the way forward
    if (tp < t1 && a.max_steps > 1) {
      constexpr int W = kLpeGradWin, NC = kLpeGradCoarse;
      int nrest = 0;  // the steps behind the first one (time arithmetic only; the forward sweep stops at max_steps as well, and raises the flag)
      {
        R up = tp, uq = tq;
        while (up < t1 && nrest + 1 < a.max_steps) {
          advance(up, uq, t1);
          ++nrest;
        }
      }
      const int Cs = (nrest <= W * NC) ? W : (nrest + NC - 1) / NC;  // steps per coarse segment
      const int G = (nrest + Cs - 1) / Cs;                            // segments (<= NC)
      // the sub-block the window holds after the forward pass: the last one of the last segment
      const int gl_first = 1 + (G - 1) * Cs, gl_len = nrest - gl_first + 1;
      const int wl_first = gl_first + ((gl_len - 1) / W) * W;
      R y = y0;
      lpe_step<R>(y, dt_0, rhs, C);
      {
        R up = tp, uq = tq;
        for (int s = 1; s <= nrest; ++s) {  // y: the start of step s; (up, uq): its interval
          const R dt = uq - up;
          const int q = s - 1;
          if (q - (q / Cs) * Cs == 0) coarse[q / Cs][lane] = (l == 15) ? up : ((l == 14) ? uq : y);
          if (s >= wl_first) starts[s - wl_first][lane] = (l == 15) ? dt : y;
          if (s < nrest) {
            lpe_step<R>(y, dt, rhs, C);
            advance(up, uq, t1);
          }
        }
      }
      for (int g = G - 1; g >= 0; --g) {
        const int sfirst = 1 + g * Cs, slast = (sfirst + Cs - 1 < nrest) ? sfirst + Cs - 1 : nrest;
        for (int bfirst = sfirst + ((slast - sfirst) / W) * W; bfirst >= sfirst; bfirst -= W) {
          const int blast = (bfirst + W - 1 < slast) ? bfirst + W - 1 : slast;
          if (bfirst != wl_first) {  // refill the window: from the segment's coarse start up to the end of this sub-block
            const R raw = coarse[g][lane];
            R up = lpe_bcast<15>(raw), uq = lpe_bcast<14>(raw);
            R yy = (i == 3) ? R(0) : raw;
            for (int s = sfirst; s <= blast; ++s) {
              const R dt = uq - up;
              if (s >= bfirst) starts[s - bfirst][lane] = (l == 15) ? dt : yy;
              if (s < blast) {
                lpe_step<R>(yy, dt, rhs, C);
                advance(up, uq, t1);
              }
            }
          }
          for (int s = blast; s >= bfirst; --s) {
            const R raw = starts[s - bfirst][lane];
            const R dt = lpe_bcast<15>(raw);
            const R ys = (i == 3) ? R(0) : raw;
            lpe_step_adj<ALL, R>(rhs, adj, C, ys, dt, vb, th, acc.lql);
          }
        }
      }
    }
    lpe_step_adj<ALL, R>(rhs, adj, C, y0, dt_0, vb, th, acc.lql);
  }
  // row p holds the shares of parameter p
  const R g0 = (lpe_bcast<0>(th) + lpe_bcast<1>(th)) + (lpe_bcast<2>(th) + lpe_bcast<3>(th));
  const R g1 = (lpe_bcast<4>(th) + lpe_bcast<5>(th)) + (lpe_bcast<6>(th) + lpe_bcast<7>(th));
  const R g2 = (lpe_bcast<8>(th) + lpe_bcast<9>(th)) + (lpe_bcast<10>(th) + lpe_bcast<11>(th));
  if (live && l < 3) grad[n * 3 + l] = (l == 0) ? g0 : ((l == 1) ? g1 : g2);
  if constexpr (ALL) {
    // the symmetric blocks as a symmetric parametrisation pairs with them: averaged with the transpose partner
    const R P0b = R(0.5) * (vb + lpe_transpose(vb, l)), Lb = R(0.5) * (acc.lql + lpe_transpose(acc.lql, l)),
            Rb = R(0.5) * (acc.r + lpe_transpose(acc.r, l));
    R* gm = grad_model + n * (21 + 4 * M + M * M);
    if (live && cov) {
      gm[3 + i * 3 + j] = P0b;
      gm[12 + i * 3 + j] = Lb;
      if (i < M) gm[21 + i * 3 + j] = acc.h;
      if (i < M && j < M) gm[21 + 4 * M + i * M + j] = Rb;
    }
    if (live && mean) {
      gm[i] = vb;
      if (i < M) gm[21 + 3 * M + i] = acc.b;
    }
  }
}

}  // namespace cdkf
