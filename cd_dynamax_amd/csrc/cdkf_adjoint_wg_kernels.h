// cdkf_adjoint_wg_kernels.h -- reverse sweep (discrete adjoint) of the EKF log-likelihood for state dimensions beyond the wavefront
// kernel's eight (cdkf_adjoint_kernels.h): d ll / d theta for the drift parameters (Lorenz-96: the forcing; linear: W and b) and, on
// request, for every other parameter of the model (m0, P0, L Qc L^T, H, bias, R), state_dim and emission_dim up to what nine q x q
// matrices of LDS allow (q = max(d, m): 41 in fp64, 58 in fp32) -- BASELINE config 4's Lorenz-96 at d = 40 among them.
//
// The reference gets this from jax.value_and_grad through the filter (ssm_temissions.py:550-568; reverse mode through diffrax with
// RecursiveCheckpointAdjoint, diffrax_utils.py:49) for a model of any size; same quantity here, written out
// (oracle/cdkf_oracle.py ekf_loglik_grad_adjoint restates it line by line):
//   forward  : the filter sweep (wavefront- or workgroup-per-trajectory) stores predicted and filtered moments at every observation;
//   backward : k = T-1 .. 0:  (1) adjoint of the measurement update + log-likelihood term at k (inference_ekf.py:153-199, 285-286),
//                             (2) adjoint of the Runge-Kutta steps over [t_{k-1}, t_k], re-integrated from the filtered moments at
//                                 k-1 (inference_ekf.py:76-123; the steps' starts kept per chunk, the stages recomputed).
//
// Mapping: ONE WORKGROUP (256 threads) per trajectory, every matrix of the step in LDS, thread <-> entries e = tid, tid + 256, ...
// Straightforward loops: this kernel is the shape-generic one (any d, any H, any R) -- a trajectory's reverse step is ~30 small
// dense products and three factorisations between barriers; the specialised sweeps of cdkf_adjoint_kernels.h / cdkf_lpe_grad_kernels.h
// keep the small shapes.  Slopes and stage cotangents of the step in hand live in a per-trajectory global scratch (L2-resident:
// 12 (d^2 + d) reals), each entry read back only by the thread that wrote it.
#pragma once
#include "cdkf_wg2_kernels.h"

namespace cdkf {

constexpr int kAwgSlots = 9;   // q x ld matrices in LDS
constexpr int kAwgVecs = 23;   // 64-entry vectors in LDS
constexpr int kAwgThreads = 256;
__host__ __device__ inline int awg_ld(int q) { return q | 1; }
__host__ __device__ inline long awg_lds_reals(int d, int m) {
  const int q = d > m ? d : m, ld = awg_ld(q);
  return (long)kAwgSlots * q * ld + (long)m * ld + 64L * kAwgVecs;
}
// layout of the optional model-gradient block (per trajectory): m0 [d] | P0 [d,d] | LQL [d,d] | H [m,d] | bias [m] | R [m,m]
// (the same as adj_model_grad_size of cdkf_adjoint_kernels.h)
__host__ __device__ inline long awg_model_grad_size(int d, int m) { return (long)d + 2L * d * d + (long)m * d + m + (long)m * m; }
// per-trajectory global scratch in reals: six slopes, six stage cotangents, `cap` step starts and their step sizes
__host__ __device__ inline long awg_scratch_reals(int d, int cap) {
  const long sz = (long)d * d + d;
  return 12 * sz + (long)cap * sz + cap;
}

template <typename R>
__global__ __launch_bounds__(kAwgThreads) void ekf_adjoint_wg_kernel(const WgArgs<R> a, R* __restrict__ grad, R* __restrict__ grad_model,
                                                                    R* __restrict__ ws, long ws_stride, int cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  R* sm = reinterpret_cast<R*>(smem_raw);
  const int tid = threadIdx.x, NT = blockDim.x;
  const long n = blockIdx.x;
  const int d = a.d, m = a.m, q = d > m ? d : m, ld = awg_ld(q), SL = q * ld;
  auto slot = [&](int s) { return sm + (long)s * SL; };
  R* Pb = slot(0);                            // cotangent of the covariance (persistent)
  R* Hs = sm + (long)kAwgSlots * SL;          // H [m][ld]
  R* vec = Hs + (long)m * ld;
  R *x0 = vec, *xs = vec + 64, *lam = vec + 128, *mb = vec + 192, *vv = vec + 256, *wv = vec + 320, *vb = vec + 384, *fv = vec + 448,
    *g1 = vec + 512, *g2 = vec + 576, *g3 = vec + 640;
  R* km = vec + 704;  // mean parts of the step's slopes   [6][64]
  R* ym = km + 384;   // mean parts of the stage cotangents [6][64]
  const R* par = a.par;
  const R* th = par + a.o_theta;
  const R* LQL = par + a.o_LQL;
  const R* Rm = par + a.o_R;
  const R* hb = par + a.o_hb;
  const bool lin = a.kind == kDriftLinear;
  const int nst = a.rk.stages;
  const long sz = (long)d * d + d;
  R* wsb = ws + n * ws_stride;
  R* ksP = wsb;               // [6][d*d] then [6][d] unused (the mean parts live in LDS)
  R* ybP = wsb + 6 * sz;      // [6][d*d]
  R* starts = wsb + 12 * sz;  // [cap][d*d + d]
  R* dts = starts + (long)cap * sz;
  const long ntheta = lin ? (long)d * d + d : 1;
  R* g = grad + n * ntheta;
  R* gm = grad_model ? grad_model + n * awg_model_grad_size(d, m) : nullptr;
  R* gP0 = gm ? gm + d : nullptr;
  R* gQ = gm ? gm + d + (long)d * d : nullptr;
  R* gH = gm ? gm + d + 2L * d * d : nullptr;
  R* gBias = gm ? gH + (long)m * d : nullptr;
  R* gR = gm ? gBias + m : nullptr;

#define AWG_FOR(e, cnt) for (int e = tid; e < (cnt); e += NT)
  // ---- start: H into LDS, accumulators to zero -------------------------------------------------------------------------
  AWG_FOR(e, m * d) {
    const int r = fdiv(e, d), c = e - r * d;
    Hs[r * ld + c] = (par + a.o_H)[e];
  }
  AWG_FOR(e, d * d) {
    const int i = fdiv(e, d), j = e - i * d;
    Pb[i * ld + j] = R(0);
  }
  if (tid < 64) mb[tid] = R(0);
  AWG_FOR(e, (int)ntheta) g[e] = R(0);
  if (gm) AWG_FOR(e, (int)awg_model_grad_size(d, m)) gm[e] = R(0);
  R gForcing = R(0);  // Lorenz-96: thread 0 accumulates d ll / d F
  int st = 0;
  __syncthreads();

  // ---- drift: dense Jacobian F(x) into a slot, f(x) into fv; x in LDS (synchronised by the caller before AND after) ------------
  auto drift_eval = [&](const R* xv, R* F) {
    AWG_FOR(e, d * d) {
      const int i = fdiv(e, d), j = e - i * d;
      R v;
      if (lin) {
        v = th[e];
      } else {  // Lorenz-96: f_i = (x_{i+1} - x_{i-2}) x_{i-1} - x_i + F
        const int ip1 = (i + 1 >= d) ? 0 : i + 1, im1 = (i == 0) ? d - 1 : i - 1, im2 = (im1 == 0) ? d - 1 : im1 - 1;
        v = R(0);
        if (j == ip1) v += xv[im1];
        if (j == im2) v -= xv[im1];
        if (j == im1) v += xv[ip1] - xv[im2];
        if (j == i) v -= R(1);
      }
      F[i * ld + j] = v;
    }
    if (tid < d) {
      const int i = tid;
      R f;
      if (lin) {
        f = th[d * d + i];
        for (int k = 0; k < d; ++k) f = rfma(th[i * d + k], xv[k], f);
      } else {
        const int ip1 = (i + 1 >= d) ? 0 : i + 1, im1 = (i == 0) ? d - 1 : i - 1, im2 = (im1 == 0) ? d - 1 : im1 - 1;
        f = rfma(xv[ip1] - xv[im2], xv[im1], th[0] - xv[i]);
      }
      fv[i] = f;
    }
  };
  // ---- lower Cholesky factor of the n x n matrix in A (in place, lower triangle), then (L L^T) X = B for the columns of B -------
  auto chol = [&](R* A, int nn) {
    for (int p = 0; p < nn; ++p) {
      __syncthreads();
      const R piv = A[p * ld + p];
      if (!(piv > R(0))) st |= kStatusNotPd;
      const R r = R(1) / rsqrt_(piv);
      __syncthreads();
      for (int i = p + tid; i < nn; i += NT) A[i * ld + p] = (i == p) ? piv * r : A[i * ld + p] * r;
      __syncthreads();
      const int w = nn - p - 1;
      AWG_FOR(e, w * w) {
        const int ii = fdiv(e, w), i = p + 1 + ii, j = p + 1 + (e - ii * w);
        if (j <= i) A[i * ld + j] = rfma(-A[i * ld + p], A[j * ld + p], A[i * ld + j]);
      }
    }
    __syncthreads();
  };
  auto chol_solve = [&](const R* L, int nn, R* B, int ncols) {  // B [nn][ld], in place; one thread per column
    if (tid < ncols) {
      const int c = tid;
      for (int r = 0; r < nn; ++r) {
        R w = B[r * ld + c];
        for (int k = 0; k < r; ++k) w = rfma(-L[r * ld + k], B[k * ld + c], w);
        B[r * ld + c] = w / L[r * ld + r];
      }
      for (int r = nn - 1; r >= 0; --r) {
        R w = B[r * ld + c];
        for (int k = r + 1; k < nn; ++k) w = rfma(-L[k * ld + r], B[k * ld + c], w);
        B[r * ld + c] = w / L[r * ld + r];
      }
    }
    __syncthreads();
  };
  // A <- 0.5 (A + A^T) for a d x d matrix, through a scratch slot
  auto symmetrize = [&](R* A, R* tmp) {
    AWG_FOR(e, d * d) {
      const int i = fdiv(e, d), j = e - i * d;
      tmp[i * ld + j] = R(0.5) * (A[i * ld + j] + A[j * ld + i]);
    }
    __syncthreads();
    AWG_FOR(e, d * d) {
      const int i = fdiv(e, d), j = e - i * d;
      A[i * ld + j] = tmp[i * ld + j];
    }
    __syncthreads();
  };

  // ---- Runge-Kutta stages of one step from (x0, P0s): slopes k_i = (km[i], ksP[i]) -----------------------------------------------
  // stage value i into (xs, Ps): y0 + dt sum_{j < i} a_ij k_j   (own entries of the global slopes)
  auto stage_value = [&](int si, const R* P0s, R* Ps, R dt) {
    AWG_FOR(e, d * d) {
      const int i = fdiv(e, d), j = e - i * d;
      R s = R(0);
      for (int jj = 0; jj < si; ++jj) s = rfma(a.rk.a[si][jj], ksP[(long)jj * d * d + e], s);
      Ps[i * ld + j] = rfma(dt, s, P0s[i * ld + j]);
    }
    if (tid < d) {
      R s = R(0);
      for (int jj = 0; jj < si; ++jj) s = rfma(a.rk.a[si][jj], km[64 * jj + tid], s);
      xs[tid] = rfma(dt, s, x0[tid]);
    }
    __syncthreads();
  };
  auto stages_fwd = [&](const R* P0s, R* Ps, R* F, R dt) {
    for (int si = 0; si < nst; ++si) {
      stage_value(si, P0s, Ps, dt);
      drift_eval(xs, F);
      __syncthreads();
      AWG_FOR(e, d * d) {  // k_P = F Ps + (F Ps)^T + L Qc L^T
        const int i = fdiv(e, d), j = e - i * d;
        R sa = R(0), sb = R(0);
        for (int k = 0; k < d; ++k) {
          sa = rfma(F[i * ld + k], Ps[k * ld + j], sa);
          sb = rfma(F[j * ld + k], Ps[k * ld + i], sb);
        }
        ksP[(long)si * d * d + e] = (sa + sb) + LQL[e];
      }
      if (tid < d) km[64 * si + tid] = fv[tid];
      __syncthreads();
    }
  };
  // y <- y + dt sum_i b_i k_i  (x0 and the matrix in P0s)
  auto step_end = [&](R* P0s, R dt) {
    AWG_FOR(e, d * d) {
      const int i = fdiv(e, d), j = e - i * d;
      R s = R(0);
      for (int si = 0; si < nst; ++si) s = rfma(a.rk.b[si], ksP[(long)si * d * d + e], s);
      P0s[i * ld + j] = rfma(dt, s, P0s[i * ld + j]);
    }
    if (tid < d) {
      R s = R(0);
      for (int si = 0; si < nst; ++si) s = rfma(a.rk.b[si], km[64 * si + tid], s);
      x0[tid] = rfma(dt, s, x0[tid]);
    }
    __syncthreads();
  };

  const R* tp = a.t + n * a.t_sn;
  const R* yp = a.y + n * a.y_sn;
  for (long k = a.T - 1; k >= 0; --k) {
    // ================= (1) measurement update + log-likelihood term at k, reversed ===============================================
    R* Pp = slot(1);   // predicted covariance
    R* HP = slot(2);   // H P            [m][d]
    R* S = slot(3);    // H P H^T + R    [m][m]; later Sbar
    R* L1 = slot(4);   // chol(S); later Kb -> Ub [m][d]
    R* Si = slot(5);   // S^-1; later Sbar H [m][d]
    R* L2 = slot(6);   // chol(sym(S) + 1e-9 I)
    R* X = slot(7);    // (sym(S) + 1e-9 I)^-1 H P   [m][d]
    R* T1 = slot(8);   // X Pbar         [m][d]
    AWG_FOR(e, d * d) {
      const int i = fdiv(e, d), j = e - i * d;
      Pp[i * ld + j] = (k == 0) ? R(0.5) * ((par + a.o_P0)[i * d + j] + (par + a.o_P0)[j * d + i])
                                : a.pP[n * a.P_sn + (k - 1) * a.P_sk + (long)e * a.P_si];
    }
    if (tid < d) x0[tid] = (k == 0) ? (par + a.o_m0)[tid] : a.pm[n * a.m_sn + (k - 1) * a.m_sk + tid * a.m_si];
    __syncthreads();
    AWG_FOR(e, m * d) {
      const int r = fdiv(e, d), c = e - r * d;
      R s = R(0);
      for (int kk = 0; kk < d; ++kk) s = rfma(Hs[r * ld + kk], Pp[kk * ld + c], s);
      HP[r * ld + c] = s;
    }
    if (tid < m) {
      R s = hb[tid];
      for (int kk = 0; kk < d; ++kk) s = rfma(Hs[tid * ld + kk], x0[kk], s);
      vv[tid] = yp[k * a.y_sk + tid * a.y_si] - s;
    }
    __syncthreads();
    AWG_FOR(e, m * m) {
      const int r = fdiv(e, m), c = e - r * m;
      R s = Rm[e];
      for (int kk = 0; kk < d; ++kk) s = rfma(HP[r * ld + kk], Hs[c * ld + kk], s);
      S[r * ld + c] = s;
    }
    __syncthreads();
    AWG_FOR(e, m * m) {
      const int r = fdiv(e, m), c = e - r * m;
      L1[r * ld + c] = S[r * ld + c];
      L2[r * ld + c] = R(0.5) * (S[r * ld + c] + S[c * ld + r]) + (r == c ? R(1e-9) : R(0));
      Si[r * ld + c] = (r == c) ? R(1) : R(0);
    }
    AWG_FOR(e, m * d) {
      const int r = fdiv(e, d), c = e - r * d;
      X[r * ld + c] = HP[r * ld + c];
    }
    chol(L1, m);
    chol(L2, m);
    chol_solve(L1, m, Si, m);
    chol_solve(L2, m, X, d);
    symmetrize(Pb, T1);
    AWG_FOR(e, m * d) {  // T1 = X Pbar
      const int r = fdiv(e, d), c = e - r * d;
      R s = R(0);
      for (int kk = 0; kk < d; ++kk) s = rfma(X[r * ld + kk], Pb[kk * ld + c], s);
      T1[r * ld + c] = s;
    }
    if (tid < m) {  // w = S^-1 v;  vbar = X mbar - w
      R w = R(0), s = R(0);
      for (int c = 0; c < m; ++c) w = rfma(Si[tid * ld + c], vv[c], w);
      for (int c = 0; c < d; ++c) s = rfma(X[tid * ld + c], mb[c], s);
      wv[tid] = w;
      vb[tid] = s - w;
    }
    __syncthreads();
    AWG_FOR(e, m * d) {  // Kb = v mbar^T - 2 S (X Pbar)   (cotangent of K^T), over the dead factor L1
      const int r = fdiv(e, d), c = e - r * d;
      R s = R(0);
      for (int kk = 0; kk < m; ++kk) s = rfma(S[r * ld + kk], T1[kk * ld + c], s);
      L1[r * ld + c] = rfma(R(-2), s, vv[r] * mb[c]);
    }
    __syncthreads();
    R* Ub = L1;
    chol_solve(L2, m, Ub, d);  // Ub = (sym(S) + 1e-9 I)^-1 Kb
    AWG_FOR(e, m * m) {  // Sbar = -(X Pbar) X^T + w w^T / 2 - S^-1 / 2 - sym(X Ub^T), over the dead S
      const int r = fdiv(e, m), c = e - r * m;
      R s1 = R(0), s2 = R(0), s3 = R(0);
      for (int kk = 0; kk < d; ++kk) {
        s1 = rfma(T1[r * ld + kk], X[c * ld + kk], s1);
        s2 = rfma(X[r * ld + kk], Ub[c * ld + kk], s2);
        s3 = rfma(X[c * ld + kk], Ub[r * ld + kk], s3);
      }
      const R sbar = -s1 + R(0.5) * wv[r] * wv[c] - R(0.5) * Si[r * ld + c] - R(0.5) * (s2 + s3);
      S[r * ld + c] = sbar;  // (S itself is dead: Kb has been formed, two barriers ago)
    }
    __syncthreads();
    R* Sbar = S;
    if (gm) {  // model block: dR += Sbar; dH += 2 Sbar (H P) - vbar m^T + Ub P; dbias -= vbar
      AWG_FOR(e, m * m) {
        const int r = fdiv(e, m), c = e - r * m;
        gR[e] += Sbar[r * ld + c];
      }
      AWG_FOR(e, m * d) {
        const int r = fdiv(e, d), c = e - r * d;
        R s1 = R(0), s2 = R(0);
        for (int kk = 0; kk < m; ++kk) s1 = rfma(Sbar[r * ld + kk], HP[kk * ld + c], s1);
        for (int kk = 0; kk < d; ++kk) s2 = rfma(Ub[r * ld + kk], Pp[kk * ld + c], s2);
        gH[e] += R(2) * s1 - vb[r] * x0[c] + s2;
      }
      if (tid < m) gBias[tid] -= vb[tid];
    }
    R* SH = Si;  // Sbar H [m][d], over the dead S^-1 (every thread is past its last read of it: the barrier above)
    AWG_FOR(e, m * d) {
      const int r = fdiv(e, d), c = e - r * d;
      R s = R(0);
      for (int kk = 0; kk < m; ++kk) s = rfma(Sbar[r * ld + kk], Hs[kk * ld + c], s);
      SH[r * ld + c] = s;
    }
    __syncthreads();
    AWG_FOR(e, d * d) {  // Pbar <- Pbar + sym(Ub^T H) + H^T Sbar H
      const int i = fdiv(e, d), j = e - i * d;
      R s1 = R(0), s2 = R(0), s3 = R(0);
      for (int r = 0; r < m; ++r) {
        s1 = rfma(Ub[r * ld + i], Hs[r * ld + j], s1);
        s2 = rfma(Ub[r * ld + j], Hs[r * ld + i], s2);
        s3 = rfma(Hs[r * ld + i], SH[r * ld + j], s3);
      }
      Pb[i * ld + j] += R(0.5) * (s1 + s2) + s3;
    }
    R mbn = R(0);
    if (tid < d) {  // mbar <- mbar - H^T vbar
      R s = R(0);
      for (int r = 0; r < m; ++r) s = rfma(Hs[r * ld + tid], vb[r], s);
      mbn = mb[tid] - s;
    }
    __syncthreads();
    if (tid < d) mb[tid] = mbn;
    __syncthreads();
    if (k == 0) break;

    // ================= (2) predict k-1 -> k: the Runge-Kutta steps of the interval, reversed ========================================
    const R t0 = tp[(k - 1) * a.t_sk], t1 = tp[k * a.t_sk];
    long Ssteps = 0;
    {
      R tprev = t0, tnext = rmin(t0 + a.dt0, t1);
      while (tprev < t1 && Ssteps < a.max_steps) {
        tprev = rmin(tnext, t1);
        const R tn = tnext + a.dt0;
        tnext = (tn > t1 - Tol<R>::v) ? t1 : tn;
        ++Ssteps;
      }
    }
    R* P0s = slot(1);  // start of the step in hand
    R* Ps = slot(2);   // stage value
    R* Lt = slot(3);   // cotangent of the stage slope before / after symmetrisation
    R* Lam = slot(4);
    R* F = slot(5);
    R* G = slot(6);    // 2 Lam Ps (linear drift: its weight gradient)
    for (long cs = ((Ssteps - 1) / cap) * cap; cs >= 0; cs -= cap) {
      const long ce = (cs + cap < Ssteps) ? cs + cap : Ssteps;
      // replay the interval from the filtered moments at k-1 up to the last start of this chunk, keeping the chunk's starts
      AWG_FOR(e, d * d) {
        const int i = fdiv(e, d), j = e - i * d;
        P0s[i * ld + j] = a.fP[n * a.P_sn + (k - 1) * a.P_sk + (long)e * a.P_si];
      }
      if (tid < d) x0[tid] = a.fm[n * a.m_sn + (k - 1) * a.m_sk + tid * a.m_si];
      __syncthreads();
      {
        R tprev = t0, tnext = rmin(t0 + a.dt0, t1);
        for (long s = 0; s < ce; ++s) {
          const R dt = tnext - tprev;
          if (s >= cs) {
            R* sv = starts + (s - cs) * sz;
            AWG_FOR(e, d * d) {
              const int i = fdiv(e, d), j = e - i * d;
              sv[e] = P0s[i * ld + j];
            }
            if (tid < d) sv[(long)d * d + tid] = x0[tid];
            if (tid == 0) dts[s - cs] = dt;
          }
          if (s + 1 < ce) {
            stages_fwd(P0s, Ps, F, dt);
            step_end(P0s, dt);
          }
          tprev = rmin(tnext, t1);
          const R tn = tnext + a.dt0;
          tnext = (tn > t1 - Tol<R>::v) ? t1 : tn;
        }
      }
      __syncthreads();
      for (long s = ce - 1; s >= cs; --s) {
        const R* sv = starts + (s - cs) * sz;
        AWG_FOR(e, d * d) {  // (each thread reads back the entries it wrote)
          const int i = fdiv(e, d), j = e - i * d;
          P0s[i * ld + j] = sv[e];
        }
        if (tid < d) x0[tid] = sv[(long)d * d + tid];
        __syncthreads();
        const R dt = dts[s - cs];
        stages_fwd(P0s, Ps, F, dt);
        for (int si = nst - 1; si >= 0; --si) {
          // cotangent of slope i: dt (b_i ybar + sum_{r > i} a_ri Ybar_r), symmetrised
          AWG_FOR(e, d * d) {
            const int i = fdiv(e, d), j = e - i * d;
            R s2 = a.rk.b[si] * Pb[i * ld + j];
            for (int r = nst - 1; r > si; --r) s2 = rfma(a.rk.a[r][si], ybP[(long)r * d * d + e], s2);
            Lt[i * ld + j] = dt * s2;
          }
          if (tid < d) {
            R s2 = a.rk.b[si] * mb[tid];
            for (int r = nst - 1; r > si; --r) s2 = rfma(a.rk.a[r][si], ym[64 * r + tid], s2);
            lam[tid] = dt * s2;
          }
          stage_value(si, P0s, Ps, dt);  // (synchronises)
          AWG_FOR(e, d * d) {
            const int i = fdiv(e, d), j = e - i * d;
            Lam[i * ld + j] = R(0.5) * (Lt[i * ld + j] + Lt[j * ld + i]);
          }
          drift_eval(xs, F);
          __syncthreads();
          // Ybar_P = F^T Lam + Lam F;  G = 2 Lam Ps where the drift's parameters / state derivative want it
          AWG_FOR(e, d * d) {
            const int i = fdiv(e, d), j = e - i * d;
            R sa = R(0), sb = R(0);
            for (int kk = 0; kk < d; ++kk) {
              sa = rfma(F[kk * ld + i], Lam[kk * ld + j], sa);
              sb = rfma(Lam[i * ld + kk], F[kk * ld + j], sb);
            }
            ybP[(long)si * d * d + e] = sa + sb;
            if (gQ) gQ[e] += Lam[i * ld + j];
            if (lin) {
              R sg = R(0);
              for (int kk = 0; kk < d; ++kk) sg = rfma(Lam[i * ld + kk], Ps[kk * ld + j], sg);
              g[e] += rfma(lam[i], xs[j], R(2) * sg);  // dW += lam x^T + G
            }
          }
          if (!lin && tid < d) {  // Lorenz-96: the three entries of row i of G the state derivative of F touches
            const int i = tid;
            const int ip1 = (i + 1 >= d) ? 0 : i + 1, im1 = (i == 0) ? d - 1 : i - 1, im2 = (im1 == 0) ? d - 1 : im1 - 1;
            R s1 = R(0), s2 = R(0), s3 = R(0);
            for (int kk = 0; kk < d; ++kk) {
              const R l = Lam[i * ld + kk];
              s1 = rfma(l, Ps[kk * ld + ip1], s1);
              s2 = rfma(l, Ps[kk * ld + im2], s2);
              s3 = rfma(l, Ps[kk * ld + im1], s3);
            }
            g1[i] = R(2) * s1;  // G[i][i+1]
            g2[i] = R(2) * s2;  // G[i][i-2]
            g3[i] = R(2) * s3;  // G[i][i-1]
          }
          __syncthreads();
          if (tid < d) {  // Ybar_m = F^T lam (+ the Jacobian's own state derivative contracted with G)
            const int c = tid;
            R s2 = R(0);
            for (int r = 0; r < d; ++r) s2 = rfma(F[r * ld + c], lam[r], s2);
            if (lin) {
              g[d * d + c] += lam[c];
            } else {
              // xbar[i-1] += G[i][i+1] - G[i][i-2];  xbar[i+1] += G[i][i-1];  xbar[i-2] -= G[i][i-1]
              const int cp1 = (c + 1 >= d) ? 0 : c + 1, cm1 = (c == 0) ? d - 1 : c - 1, cp2 = (cp1 + 1 >= d) ? 0 : cp1 + 1;
              s2 += g1[cp1] - g2[cp1];
              s2 += g3[cm1];
              s2 -= g3[cp2];
            }
            ym[64 * si + c] = s2;
          }
          if (!lin && tid == 0) {
            R s2 = R(0);
            for (int r = 0; r < d; ++r) s2 += lam[r];
            gForcing += s2;
          }
          __syncthreads();
        }
        // cotangent of the step's start
        AWG_FOR(e, d * d) {
          const int i = fdiv(e, d), j = e - i * d;
          R s2 = Pb[i * ld + j];
          for (int si = 0; si < nst; ++si) s2 += ybP[(long)si * d * d + e];
          Pb[i * ld + j] = s2;
        }
        if (tid < d) {
          R s2 = mb[tid];
          for (int si = 0; si < nst; ++si) s2 += ym[64 * si + tid];
          mb[tid] = s2;
        }
        __syncthreads();
        symmetrize(Pb, Lt);
      }
    }
  }
  // ---- results ------------------------------------------------------------------------------------------------------------------
  if (gm) {
    if (tid < d) gm[tid] = mb[tid];
    AWG_FOR(e, d * d) {
      const int i = fdiv(e, d), j = e - i * d;
      gP0[e] = R(0.5) * (Pb[i * ld + j] + Pb[j * ld + i]);
    }
  }
  if (!lin && tid == 0) g[0] = gForcing;
  if (st && tid == 0 && a.status) atomicOr(&a.status[n], st);
#undef AWG_FOR
}

}  // namespace cdkf
