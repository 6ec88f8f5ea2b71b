// cdkf_rts1_kernels.h -- the linear model's smoother type 1 ("cd_smoother_1", the reference's default): discrete RTS on the
// pushed-forward transition pair (A, Q) of every interval.  state_dim <= 8, wavefront kernels on 8 x 8 tiles.
//
// Reference (src/continuous_discrete_linear_gaussian_ssm/inference.py):
//   compute_pushforward :105-143   A' = F A, Q' = F Q + Q F^T + L Qc L^T from (I, 0) over [t_k, t_{k+1}], Dopri5 dt0 = 0.01
//   _step_1             :746-773   C = psd_solve(Q + A P_f A^T, A P_f)^T;  m_s = m_f + C (m_s' - A m_f);
//                                  P_s = P_f + C (P_s' - A P_f A^T - Q) C^T;  cross = C P_s' + m_s m_s'^T
//
// Two launches after the filter sweep:
//   pushforward_wave8_kernel : one wavefront per (trajectory, interval) -- the pairs do not depend on the data, so all
//                              N (T-1) of them integrate concurrently (lane (i, j) owns A_ij and Q_ij and their slopes);
//   rts1_wave8_kernel        : one wavefront per trajectory walks k = T-2 .. 0.
#pragma once
#include "cdkf_adjoint_kernels.h"

namespace cdkf {

struct Rts1Off {  // per wavefront, in reals
  static constexpr int A = 0, Q = 64, F = 128, T0 = 192, T1 = 256, T2 = 320, S2 = 384, P = 448;
  static constexpr int v0 = 512, v1 = 520;
  static constexpr int end = 528;
};
constexpr int kRts1Waves = 4;

// (A, Q) for interval k of trajectory n, stored as two d x d blocks at AQ[((n (T-1) + k) 2 + {0,1}) d d]
template <typename R>
__global__ __launch_bounds__(64 * kRts1Waves) void pushforward_wave8_kernel(const WgArgs<R> a, R* __restrict__ AQ) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  R* W = reinterpret_cast<R*>(smem_raw) + (threadIdx.x >> 6) * Rts1Off::end;
  const int lane = threadIdx.x & 63, i = lane >> 3, j = lane & 7;
  const int d = a.d;
  const long item = (long)blockIdx.x * kRts1Waves + (threadIdx.x >> 6);
  const long per = a.T - 1;
  if (item >= a.N * per) return;
  const long n = item / per, k = item - n * per;
  const bool in = (i < d) && (j < d);
  const R Fij = in ? (a.par + a.o_theta)[i * d + j] : R(0);
  const R lql = in ? (a.par + a.o_LQL)[i * d + j] : R(0);
  W[Rts1Off::F + lane] = Fij;
  wave_sync();
  auto mm = [&](int TA, int TB) {
    R s = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) s = rfma(W[TA + i * 8 + q], W[TB + q * 8 + j], s);
    return s;
  };
  auto rhs = [&](R As, R Qs, R& kA, R& kQ) {
    W[Rts1Off::A + lane] = As;
    W[Rts1Off::Q + lane] = Qs;
    wave_sync();
    kA = mm(Rts1Off::F, Rts1Off::A);
    const R fq = mm(Rts1Off::F, Rts1Off::Q);
    W[Rts1Off::T0 + lane] = fq;
    wave_sync();
    kQ = (fq + W[Rts1Off::T0 + j * 8 + i]) + lql;
    wave_sync();
  };
  // the Runge-Kutta method of opts.solver (a.rk, run-time tableau in the kernel arguments; static stage indices after unrolling),
  // fixed steps of dt0 or diffrax.PIDController around the embedded pair -- the error norm then runs over the (A, Q) pytree
  const RkTab<R>& tb = a.rk;
  auto stage_in = [&](int si, R y0, const R (&ks)[6], R dt) {
    R s = 0;
#pragma unroll
    for (int jj = 0; jj < 5; ++jj)
      if (jj < si) s = rfma(tb.a[si][jj], ks[jj], s);
    return rfma(dt, s, y0);
  };
  const R* tp = a.t + n * a.t_sn;
  const R t0 = tp[k * a.t_sk], t1 = tp[(k + 1) * a.t_sk];
  R Aij = (in && i == j) ? R(1) : R(0), Qij = 0;
  // (adaptive: the first size clipped to [dtmin, dtmax], a step at dtmin kept -- integrate_adaptive, cdkf_math.h)
  const R dt_first = tb.adaptive ? rmin(a.dt0, tb.dtmax) : a.dt0;
  bool at_min = tb.adaptive && dt_first <= tb.dtmin;
  R tprev = t0, tnext = rmin(t0 + (tb.adaptive ? rmax(dt_first, tb.dtmin) : a.dt0), t1);
  R inv1 = R(1), inv2 = R(1);
  long steps = 0;
  while (tprev < t1 && steps < a.max_steps) {
    const R dt = tnext - tprev;
    R kA[6] = {0, 0, 0, 0, 0, 0}, kQ[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 6; ++s)
      if (s < tb.stages) rhs(stage_in(s, Aij, kA, dt), stage_in(s, Qij, kQ, dt), kA[s], kQ[s]);  // uniform
    R sa = 0, sq = 0;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      sa = rfma(tb.b[s], kA[s], sa);
      sq = rfma(tb.b[s], kQ[s], sq);
    }
    const R An = rfma(dt, sa, Aij), Qn = rfma(dt, sq, Qij);
    if (!tb.adaptive) {
      Aij = An;
      Qij = Qn;
      tprev = rmin(tnext, t1);
      const R tn = tnext + a.dt0;
      tnext = (tn > t1 - Tol<R>::v) ? t1 : tn;
    } else {  // integrate_adaptive (cdkf_math.h), the RMS over the 2 d^2 entries this wavefront holds
      R k7A = 0, k7Q = 0;
      if (tb.fsal) rhs(An, Qn, k7A, k7Q);
      R eA = tb.berr[6] * k7A, eQ = tb.berr[6] * k7Q;
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        eA = rfma(tb.berr[s], kA[s], eA);
        eQ = rfma(tb.berr[s], kQ[s], eQ);
      }
      const R sA = (dt * eA) / rfma(rmax(rabs(Aij), rabs(An)), tb.rtol, tb.atol);
      const R sQ = (dt * eQ) / rfma(rmax(rabs(Qij), rabs(Qn)), tb.rtol, tb.atol);
      double ssum = in ? (double)(sA * sA) + (double)(sQ * sQ) : 0.0;
#pragma unroll
      for (int o_ = 32; o_ >= 1; o_ >>= 1) ssum += __shfl_xor(ssum, o_);
      const R scaled = rsqrt_((R)ssum / R(2 * d * d));
      const bool keep = scaled < R(1) || at_min;
      const R inv = (scaled == R(0)) ? R(__builtin_huge_val()) : R(1) / scaled;
      R factor = tb.safety * rpow(inv, tb.c1);
      if (tb.c2 != R(0)) factor *= rpow(inv1, tb.c2);
      if (tb.c3 != R(0)) factor *= rpow(inv2, tb.c3);
      factor = rmin(rmax(factor, keep ? R(1) : tb.fmin), tb.fmax);
      const R nt0 = keep ? tnext : tprev;
      R dtn = rmin(dt * factor, tb.dtmax);
      at_min = dtn <= tb.dtmin;
      dtn = rmax(dtn, tb.dtmin);
      const R nt1 = nt0 + dtn;
      if (keep) {
        Aij = An;
        Qij = Qn;
        inv2 = inv1;
        inv1 = inv;
      }
      tprev = rmin(nt0, t1);
      tnext = (nt1 > t1 - Tol<R>::v) ? (keep ? t1 : rfma(R(0.5), t1 - tprev, tprev)) : nt1;
    }
    ++steps;
  }
  if (tprev < t1 && a.status) atomicOr(&a.status[n], kStatusMaxSteps);
  if (in) {
    R* o = AQ + item * 2 * d * d;
    o[i * d + j] = Aij;
    o[d * d + i * d + j] = Qij;
  }
}

// backward pass; sm / sP / cross follow the strides of the filtered arrays (cross: entries k = 0 .. T-2)
template <typename R>
__global__ __launch_bounds__(64 * kRts1Waves) void rts1_wave8_kernel(const WgArgs<R> a, const R* __restrict__ AQ,
                                                                     R* __restrict__ cross) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  R* W = reinterpret_cast<R*>(smem_raw) + (threadIdx.x >> 6) * Rts1Off::end;
  const int lane = threadIdx.x & 63, i = lane >> 3, j = lane & 7;
  const int d = a.d;
  const long n = (long)blockIdx.x * kRts1Waves + (threadIdx.x >> 6);
  if (n >= a.N) return;
  const bool in = (i < d) && (j < d);
  auto mm = [&](int TA, int TB) {
    R s = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) s = rfma(W[TA + i * 8 + q], W[TB + q * 8 + j], s);
    return s;
  };
  auto mm_nt = [&](int TA, int TB) {
    R s = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) s = rfma(W[TA + i * 8 + q], W[TB + j * 8 + q], s);
    return s;
  };
  const long per = a.T - 1;
  auto mo = [&](long k) { return n * a.m_sn + k * a.m_sk + lane * a.m_si; };
  auto po = [&](long k) { return n * a.P_sn + k * a.P_sk + (long)(i * d + j) * a.P_si; };
  // smoothed moments at T-1 = filtered
  R smn = (lane < d) ? a.fm[mo(a.T - 1)] : R(0);
  R sPn = in ? a.fP[po(a.T - 1)] : R(0);
  if (lane < d) a.sm[mo(a.T - 1)] = smn;
  if (in) a.sP[po(a.T - 1)] = sPn;
  int st = 0;
  for (long k = a.T - 2; k >= 0; --k) {
    const R* aq = AQ + (n * per + k) * 2 * d * d;
    const R Aij = in ? aq[i * d + j] : R(0), Qij = in ? aq[d * d + i * d + j] : R(0);
    const R mf = (lane < d) ? a.fm[mo(k)] : R(0);
    const R Pf = in ? a.fP[po(k)] : R(0);
    W[Rts1Off::A + lane] = Aij;
    W[Rts1Off::P + lane] = Pf;
    if (lane < 8) {
      W[Rts1Off::v0 + lane] = mf;
      W[Rts1Off::v1 + lane] = smn;
    }
    wave_sync();
    const R ap = mm(Rts1Off::A, Rts1Off::P);  // A P_f
    W[Rts1Off::T0 + lane] = ap;
    wave_sync();
    const R ppred = mm_nt(Rts1Off::T0, Rts1Off::A) + Qij;  // A P_f A^T + Q
    W[Rts1Off::T1 + lane] = ppred;
    wave_sync();
    // psd_solve: symmetrize + 1e-9 I, Cholesky (padded with the identity), solve for C^T = Sb^-1 (A P_f)
    R s2 = in ? R(0.5) * (ppred + W[Rts1Off::T1 + j * 8 + i]) + (i == j ? R(1e-9) : R(0)) : (i == j ? R(1) : R(0));
    R inv2[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      W[Rts1Off::S2 + lane] = s2;
      wave_sync();
      const R p2 = W[Rts1Off::S2 + p * 8 + p];
      if (p < d && !(p2 > R(0))) st |= kStatusNotPd;
      const R r2 = rrsqrt(p2);
      inv2[p] = r2;
      const R l2i = W[Rts1Off::S2 + i * 8 + p] * r2, l2j = W[Rts1Off::S2 + j * 8 + p] * r2;
      wave_sync();
      if (j == p && i >= p)
        s2 = (i == p) ? p2 * r2 : l2i;
      else if (i > p && j > p && j <= i)
        s2 = rfma(-l2i, l2j, s2);
    }
    W[Rts1Off::S2 + lane] = s2;
    wave_sync();
    R col[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) col[r] = W[Rts1Off::T0 + r * 8 + j];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      R w = col[r];
#pragma unroll
      for (int c = 0; c < r; ++c) w = rfma(-W[Rts1Off::S2 + r * 8 + c], col[c], w);
      col[r] = w * inv2[r];
    }
#pragma unroll
    for (int r = 7; r >= 0; --r) {
      R w = col[r];
#pragma unroll
      for (int c = r + 1; c < 8; ++c) w = rfma(-W[Rts1Off::S2 + c * 8 + r], col[c], w);
      col[r] = w * inv2[r];
    }
    R ct = 0;  // C^T[i][j]
#pragma unroll
    for (int r = 0; r < 8; ++r)
      if (r == i) ct = col[r];
    if (!in) ct = 0;
    W[Rts1Off::T2 + lane] = ct;      // C^T
    W[Rts1Off::Q + lane] = sPn;      // P_s'
    // dm = m_s' - A m_f (lanes < 8)
    if (lane < 8) {
      R s = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) s = rfma(W[Rts1Off::A + lane * 8 + q], W[Rts1Off::v0 + q], s);
      W[Rts1Off::v0 + lane] = smn - s;
    }
    wave_sync();
    // D = P_s' - Ppred;  C D = sum_q C[i][q] D[q][j] with C[i][q] = C^T[q][i]
    W[Rts1Off::F + lane] = sPn - ppred;
    wave_sync();
    R cd = 0, cps = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const R c = W[Rts1Off::T2 + q * 8 + i];
      cd = rfma(c, W[Rts1Off::F + q * 8 + j], cd);
      cps = rfma(c, W[Rts1Off::Q + q * 8 + j], cps);  // (C P_s')[i][j]
    }
    wave_sync();
    W[Rts1Off::T0 + lane] = cd;
    wave_sync();
    // P_s = P_f + (C D) C^T:  sum_q (C D)[i][q] C[j][q] = sum_q CD[i][q] C^T[q][j]
    const R sPk = Pf + mm(Rts1Off::T0, Rts1Off::T2);
    R smk = 0;
    if (lane < 8) {
      R s = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) s = rfma(W[Rts1Off::T2 + q * 8 + lane], W[Rts1Off::v0 + q], s);  // (C dm)[lane]
      smk = (lane < d) ? mf + s : R(0);
      W[Rts1Off::v0 + lane] = smk;
    }
    wave_sync();
    if (lane < d) a.sm[mo(k)] = smk;
    if (in) {
      a.sP[po(k)] = sPk;
      if (cross) cross[po(k)] = rfma(W[Rts1Off::v0 + i], W[Rts1Off::v1 + j], cps);
    }
    wave_sync();
    smn = smk;
    sPn = in ? sPk : R(0);
  }
  if (st && lane == 0 && a.status) a.status[n] |= st;
}

}  // namespace cdkf
