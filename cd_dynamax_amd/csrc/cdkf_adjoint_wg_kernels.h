// cdkf_adjoint_wg_kernels.h -- reverse sweep (discrete adjoint) of the EKF log-likelihood for state dimensions beyond the wavefront
// kernel's eight (cdkf_adjoint_kernels.h): d ll / d theta for the drift parameters (Lorenz-96: the forcing; linear: W and b) and, on
// request, for every other parameter of the model (m0, P0, L Qc L^T, H, bias, R), state_dim and emission_dim up to what nine q x q
// matrices of LDS allow (q = max(d, m): 43 in fp64, 62 in fp32) -- BASELINE config 4's Lorenz-96 at d = 40 among them.
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
constexpr int kAwgVecs = 24;   // 64-entry vectors in LDS
constexpr int kAwgThreads = 256;  // (512 -- two wavefronts per SIMD at 256 registers each -- spills 266 registers even at four entries per thread: not pursued)
#ifndef CDKF_AWG_SOLVE_PANEL
#define CDKF_AWG_SOLVE_PANEL 8  // (16 -- six substitution panel steps per system instead of ten -- measured 20.9 ms against 20.5 at d = 40)
#endif
__host__ __device__ inline int awg_ld(int q) { return q | 1; }
__host__ __device__ inline long awg_lds_reals(int d, int m) {
  const int q = d > m ? d : m, ld = awg_ld(q);
  return (long)kAwgSlots * q * ld + (long)m * ld + 64L * kAwgVecs;
}
// MLP drift (round 4; d -> h1 -> h2 -> d, tanh): behind the vectors the weights W2 [h2][h1 | 1], W1 [h1][d | 1], W3 [d][h2 | 1], twenty
// 64-entry work vectors and four [64][d | 1] images (the tangent pass U = D1 W1, T = W2 U and the cotangents of the two)
constexpr int kAwgMlpVecs = 20;
__host__ __device__ inline long awg_mlp_lds_reals(int d, int h1, int h2) {
  return (long)h2 * (h1 | 1) + (long)h1 * (d | 1) + (long)d * (h2 | 1) + 64L * kAwgMlpVecs + 4L * 64 * (d | 1);
}
// layout of the optional model-gradient block (per trajectory): m0 [d] | P0 [d,d] | LQL [d,d] | H [m,d] | bias [m] | R [m,m]
// (the same as adj_model_grad_size of cdkf_adjoint_kernels.h)
__host__ __device__ inline long awg_model_grad_size(int d, int m) { return (long)d + 2L * d * d + (long)m * d + m + (long)m * m; }
// per-trajectory global scratch in reals: `cap` step starts of a replay chunk and their step sizes
__host__ __device__ inline long awg_scratch_reals(int d, int cap) {
  const long sz = (long)d * d + d;
  return (long)cap * sz + 2 * (long)cap;  // (+ the sizes AND the start times of the chunk's steps: a drift given as source may read t)
}
// covariance entries per thread on the (column, row group) map of a d x d matrix
__host__ __device__ inline int awg_entries_per_thread(int d) {
  const int rs = kAwgThreads / d;
  return (d + rs - 1) / rs;
}

// 1 / sqrt(x): the hardware estimate and one third-order correction (fp64: to the last bit or two, as w40_rsqrt of the wavefront kernels)
__device__ __forceinline__ double awg_rsqrt(double x) {
  const double y0 = __builtin_amdgcn_rsq(x);
  const double e = rfma(-(x * y0), y0, 1.0);
  return rfma(y0 * e, rfma(e, 0.375, 0.5), y0);
}
__device__ __forceinline__ float awg_rsqrt(float x) { return 1.0f / sqrtf(x); }

#ifdef CDKF_AWG_PROFILE  // local diagnostic build: cycles per phase (s_memtime), printed by trajectory 0
static __device__ long long awg_prof[16];
#define AWG_TICK(i)                                                  \
  {                                                                  \
    const long long awg_now = clock64();                             \
    if (threadIdx.x == 0 && blockIdx.x == 0) awg_prof[i] += awg_now - awg_last; \
    awg_last = clock64();                                            \
  }
#define AWG_TICK2(i) AWG_TICK(i)
#else
#define AWG_TICK(i)
#define AWG_TICK2(i)
#endif

#ifdef CDKF_AWG_CUSTOM
// A drift given as C source (launch_custom.hip compiles this header at run time together with the definitions of these two, which
// differentiate the source by dual numbers, cdkf_dual.h; CDKF_AWG_CUSTOM = the drift's number of parameters):
//   awg_custom_column:   column j of the Jacobian at x into F (leading dimension ld) and, if fv is non-null, f(x) into fv;
//   awg_custom_contract: sum_i G[i][j] d2 f_i / dx_j dz  (+ sum_i lam_i d f_i / dz if lam is non-null), z < d: the state component z,
//                        z >= d: the parameter z - d -- what the reverse of F(m, theta) Ps + (F Ps)^T contributes to the cotangent of z.
template <typename R>
__device__ void awg_custom_column(const R* th, const R* x, int j, R* F, int ld, R* fv, const R* uin, R tin);
template <typename R>
__device__ R awg_custom_contract(const R* th, const R* x, const R* G, int ld, int j, int z, const R* lam, const R* uin, R tin);
// state_order 'second' (CDKF_AWG_CUSTOM_SECOND: grad(div f) registered as "auto"): the mean's slope carries 0.5 Ps g(x), g = grad(div f)
//   awg_custom_divpair: d2 f_i / dx_i dx_k  (g_k is its sum over i);
//   awg_custom_third:   sum_k u_k d3 f_i / dx_i dx_k dz -- what the reverse of u . g(x, theta), u = 0.5 Ps^T lam, contributes to the
//                       cotangent of the state component / parameter z (its sum over i)
template <typename R>
__device__ R awg_custom_divpair(const R* th, const R* x, int i, int k, const R* uin, R tin);
template <typename R>
__device__ R awg_custom_third(const R* th, const R* x, const R* u, int i, int z, const R* uin, R tin);
#ifndef CDKF_AWG_CUSTOM_SECOND
#define CDKF_AWG_CUSTOM_SECOND 0
#endif
#endif

// NE: covariance entries a thread owns at most (rows i0, i0 + rs, ... of its column on the d x d map): 8 up to d = 42, else 16
// MLP: the instantiation that carries the network's forward / reverse passes (a separate one: in the Lorenz-96 / linear / source-drift
// instantiation that code would only cost registers -- 560 -> 788 B of scratch per lane in fp64 when it was compiled in)
template <typename R, int NE, bool MLP = false>
__global__ __launch_bounds__(kAwgThreads) void ekf_adjoint_wg_kernel(const WgArgs<R> a, R* __restrict__ grad, R* __restrict__ grad_model,
                                                                    R* __restrict__ ws, long ws_stride, int cap) {
#ifdef CDKF_WG_STATIC_LDS  // run-time compiled for one shape (no dynamic-LDS cap to raise on a module function)
  __shared__ __attribute__((aligned(16))) unsigned char smem_raw[CDKF_WG_STATIC_LDS];
#else
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
#endif
  R* sm = reinterpret_cast<R*>(smem_raw);
  const int tid = threadIdx.x, NT = blockDim.x;
#ifdef CDKF_AWG_PROFILE
  long long awg_last = clock64();
#endif
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
  R* lam2 = ym + 384; // second copy of lam (stages alternate)
  const R* par = a.par;
  const R* th = par + a.o_theta;
  const R* LQL = par + a.o_LQL;
  const R* Rm = par + a.o_R;
  const R* hb = par + a.o_hb;
  const bool lin = a.kind == kDriftLinear;
  const int nst = a.rk.stages;
  const long sz = (long)d * d + d;
  R* wsb = ws + n * ws_stride;
  R* starts = wsb;            // [cap][d*d + d]
  R* dts = starts + (long)cap * sz;
  R* tss = dts + cap;         // start time of each kept step (stage times t + c_i dt of f(x, u, t): inference_ekf.py:95)
#ifdef CDKF_AWG_CUSTOM
  const bool custom = a.kind >= kDriftCustomBase;
  const long ntheta = custom ? (long)CDKF_AWG_CUSTOM : (lin ? (long)d * d + d : 1);
  const bool second = custom && CDKF_AWG_CUSTOM_SECOND && a.order == 2;
  // the (group, z) tasks of the second-derivative contraction and where their partial sums meet: a free slot, or -- for the smallest
  // shapes, whose slots hold less than d + n_theta reals -- a free 64-entry vector
  const int cZ = d + (int)ntheta;
  R* const cpart = (SL >= cZ) ? slot(8) : (vec + 640);  // (vec + 640: g3, a Lorenz-96 vector)
  const int ccap = (SL >= cZ) ? SL : 64;
  const int cNG = (NT / cZ < ccap / cZ) ? NT / cZ : ccap / cZ;
#else
  constexpr bool custom = false;
  // (round 4) the MLP drift beyond eight state dimensions: hidden sizes <= 64, both state orders -- the reverse of a right-hand side
  // is the oracle's MLP branch of the drift's vector-Jacobian product and divgrad_vjp, line by line, on the workgroup's threads
  constexpr bool mlp = MLP;
  const bool second = mlp && a.order == 2;
  const int h1 = a.h1, h2 = a.h2;
  const long ntheta = lin ? (long)d * d + d : (mlp ? (long)h1 * d + h1 + (long)h2 * h1 + h2 + (long)d * h2 + d : 1);
#endif
  R* g = grad + n * ntheta;
  R* gm = grad_model ? grad_model + n * awg_model_grad_size(d, m) : nullptr;
  R* gP0 = gm ? gm + d : nullptr;
  R* gQ = gm ? gm + d + (long)d * d : nullptr;
  R* gH = gm ? gm + d + 2L * d * d : nullptr;
  R* gBias = gm ? gH + (long)m * d : nullptr;
  R* gR = gm ? gBias + m : nullptr;

#define AWG_FOR(e, cnt) for (int e = tid; e < (cnt); e += NT)
  // Element-wise passes over a rows x cols matrix: thread = (column, row group) with NT / cols row groups taking the rows in turn -- no
  // index arithmetic per entry (the maps for cols = d and cols = m are formed once) -- in batches of four rows per thread: the new
  // values of a batch are formed (LDS / global reads in flight together) before any of them is written; a lone wavefront per SIMD has
  // nothing else to cover the round trips with.  Every pass over a d x d matrix uses the same map, so an entry of the global scratch
  // is always read back by the thread that wrote it.
  struct Map {
    int j, i0, rs;  // column, first row, row stride; i0 >= rs: the thread sits out
  };
  auto make_map = [&](int cols) {
    Map M;
    M.rs = fdiv(NT, cols);
    M.i0 = fdiv(tid, cols);
    M.j = tid - M.i0 * cols;
    if (M.i0 >= M.rs) M.i0 = 1 << 20;
    return M;
  };
  const Map map_d = make_map(d), map_m = make_map(m);
  auto map_for = [&](int cols) { return cols == d ? map_d : (cols == m ? map_m : make_map(cols)); };
  auto rows2d = [&](int rows, int cols, auto&& value, auto&& store) {
    const Map M = map_for(cols);
    for (int ib = M.i0; ib < rows; ib += 4 * M.rs) {
      R v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = ib + u * M.rs;
        v[u] = (i < rows) ? value(i, M.j) : R(0);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = ib + u * M.rs;
        if (i < rows) store(i, M.j, v[u]);
      }
    }
  };
  // Dense product in 2 x 4 register tiles, one tile per thread: out(i, j, sum_k A(i, k) B(k, j)) for i < rows, j < cols.  Twelve LDS
  // reads per sixteen multiply-adds (a dot product per entry takes two per multiply-add and waits for each).
  auto gemm = [&](int rows, int cols, int K, auto&& Aat, auto&& Bat, auto&& out) {
#ifdef CDKF_AWG_MFMA_GEMM
    // The same product on the matrix cores (16 x 16 output tiles dealt to the four wavefronts, v_mfma_{f64,f32}_16x16x4, operands of
    // eight k-steps in flight): measured, not used -- 31.5 ms against the register tiles' 30.3 per 256 x 100 slice at d = 40 in fp64
    // (33.6 without the operand prefetch; fp32 26.1 against 26.4): nine tiles over four wavefronts are three rounds, 44 % of a
    // 48 x 48 tile grid is padding, and the f64 matrix rate is the vector rate on this part.  Kept for the record (-DCDKF_AWG_MFMA_GEMM).
    {
      const int wave = tid >> 6, lane = tid & 63, lm = lane & 15, lg = lane >> 4;
      const int tcn = (cols + 15) >> 4, ntile = ((rows + 15) >> 4) * tcn;
      for (int tile = wave; tile < ntile; tile += (NT >> 6)) {
        const int ti = fdiv(tile, tcn), tj = tile - ti * tcn;
        const int ai = 16 * ti + lm, bj = 16 * tj + lm;
        const bool aok = ai < rows, bok = bj < cols;
        const int aic = aok ? ai : 0, bjc = bok ? bj : 0;
        typename WgAcc<R>::type acc = {0, 0, 0, 0};
        // the operands of eight k-steps in flight before the first of their products (a lone wavefront per SIMD has nothing else to cover
        // the LDS round trips with)
        for (int k0 = 0; k0 < K; k0 += 32) {
          R av[8], bv[8];
#pragma unroll
          for (int s8 = 0; s8 < 8; ++s8) {
            const int kk = k0 + 4 * s8 + lg;
            const bool kok = kk < K;
            const int kc = kok ? kk : 0;
            const R a_ = Aat(aic, kc), b_ = Bat(kc, bjc);
            av[s8] = (aok && kok) ? a_ : R(0);
            bv[s8] = (bok && kok) ? b_ : R(0);
          }
#pragma unroll
          for (int s8 = 0; s8 < 8; ++s8)
            if (k0 + 4 * s8 < K) acc = wg_mfma(av[s8], bv[s8], acc);
        }
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int row = 16 * ti + WgAcc<R>::row(lane, r4);
          if (row < rows && bok) out(row, bj, acc[r4]);
        }
      }
      return;
    }
#endif
    const int tr = (rows + 1) >> 1, tc = (cols + 3) >> 2;
    for (int tile = tid; tile < tr * tc; tile += NT) {
      const int ti = fdiv(tile, tc), tj = tile - ti * tc;
      const int i0 = 2 * ti, j0 = 4 * tj;
      const int i1 = (i0 + 1 < rows) ? i0 + 1 : i0;
      int jj[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) jj[u] = (j0 + u < cols) ? j0 + u : cols - 1;
      R acc0[4] = {0, 0, 0, 0}, acc1[4] = {0, 0, 0, 0};
      int k = 0;
      for (; k + 2 <= K; k += 2) {
        const R a00 = Aat(i0, k), a01 = Aat(i0, k + 1), a10 = Aat(i1, k), a11 = Aat(i1, k + 1);
        R b0[4], b1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          b0[u] = Bat(k, jj[u]);
          b1[u] = Bat(k + 1, jj[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc0[u] = rfma(a01, b1[u], rfma(a00, b0[u], acc0[u]));
          acc1[u] = rfma(a11, b1[u], rfma(a10, b0[u], acc1[u]));
        }
      }
      if (k < K) {
        const R a00 = Aat(i0, k), a10 = Aat(i1, k);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const R b = Bat(k, jj[u]);
          acc0[u] = rfma(a00, b, acc0[u]);
          acc1[u] = rfma(a10, b, acc1[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (j0 + u < cols) {
          out(i0, j0 + u, acc0[u]);
          if (i0 + 1 < rows) out(i0 + 1, j0 + u, acc1[u]);
        }
    }
  };
  // sum_k A(k) B(k), four independent chains
  auto dot = [&](int K, auto&& Aat, auto&& Bat) {
    R s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int k = 0;
    for (; k + 4 <= K; k += 4) {
      const R a0 = Aat(k), a1 = Aat(k + 1), a2 = Aat(k + 2), a3 = Aat(k + 3);
      const R b0 = Bat(k), b1 = Bat(k + 1), b2 = Bat(k + 2), b3 = Bat(k + 3);
      s0 = rfma(a0, b0, s0);
      s1 = rfma(a1, b1, s1);
      s2 = rfma(a2, b2, s2);
      s3 = rfma(a3, b3, s3);
    }
    for (; k < K; ++k) s0 = rfma(Aat(k), Bat(k), s0);
    return (s0 + s1) + (s2 + s3);
  };
  // The same product for TWO operand sets (system 0: cols0 columns, system 1: cols1, zero = absent) with the tiles of both dealt to
  // the threads in one round -- the factorisations' and substitutions' rank-8 updates of two systems in lockstep are ~ 130 tiles
  // each: one after the other they left half the workgroup idle twice.  The operand lambdas take the system as their first argument.
  auto gemm2 = [&](int rows, int cols0, int cols1, int K, auto&& Aat, auto&& Bat, auto&& out) {
    const int tr = (rows + 1) >> 1, tc0 = (cols0 + 3) >> 2, tcs = tc0 + ((cols1 + 3) >> 2);
    for (int tile = tid; tile < tr * tcs; tile += NT) {
      const int ti = fdiv(tile, tcs), tjj = tile - ti * tcs;
      const int sy = tjj >= tc0 ? 1 : 0, tj = sy ? tjj - tc0 : tjj, cols = sy ? cols1 : cols0;
      const int i0 = 2 * ti, j0 = 4 * tj;
      const int i1 = (i0 + 1 < rows) ? i0 + 1 : i0;
      int jj[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) jj[u] = (j0 + u < cols) ? j0 + u : cols - 1;
      R acc0[4] = {0, 0, 0, 0}, acc1[4] = {0, 0, 0, 0};
      int k = 0;
      for (; k + 2 <= K; k += 2) {
        const R a00 = Aat(sy, i0, k), a01 = Aat(sy, i0, k + 1), a10 = Aat(sy, i1, k), a11 = Aat(sy, i1, k + 1);
        R b0[4], b1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          b0[u] = Bat(sy, k, jj[u]);
          b1[u] = Bat(sy, k + 1, jj[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc0[u] = rfma(a01, b1[u], rfma(a00, b0[u], acc0[u]));
          acc1[u] = rfma(a11, b1[u], rfma(a10, b0[u], acc1[u]));
        }
      }
      if (k < K) {
        const R a00 = Aat(sy, i0, k), a10 = Aat(sy, i1, k);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const R b = Bat(sy, k, jj[u]);
          acc0[u] = rfma(a00, b, acc0[u]);
          acc1[u] = rfma(a10, b, acc1[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (j0 + u < cols) {
          out(sy, i0, j0 + u, acc0[u]);
          if (i0 + 1 < rows) out(sy, i0 + 1, j0 + u, acc1[u]);
        }
    }
  };
  // ---- start: H into LDS, accumulators to zero -------------------------------------------------------------------------
  AWG_FOR(e, m * d) {
    const int r = fdiv(e, d), c = e - r * d;
    Hs[r * ld + c] = (par + a.o_H)[e];
  }
  rows2d(d, d, [&](int, int) { return R(0); }, [&](int i, int j, R v) { Pb[i * ld + j] = v; });
  if (tid < 64) mb[tid] = R(0);
  // Does the emission pick state components (every row of H a unit vector, no two alike, no bias -- H = I, H = I[:m], any subset in
  // any order: BASELINE config 4's H = I among them)?  Then the five products of the update's adjoint with H as a factor are copies
  // of the other factor's rows / columns: obs_s[r] = the component row r observes, inv_s[c] = the row that observes component c or -1.
  __shared__ unsigned char obs_s[64];  // (bytes: the launchers raise the dynamic-LDS cap to all but 256 bytes of the CU's 160 KB)
  __shared__ signed char inv_s[64];
  __shared__ int sel_s;  // (a flag of our own: __syncthreads_and brings 256 bytes of static LDS with it)
  if (tid == 0) sel_s = 1;
  __syncthreads();
  int sel_ok = 1;
  if (tid < m) {
    int col = -1, bad_row = (hb[tid] != R(0)) ? 1 : 0;
    for (int c = 0; c < d; ++c) {
      const R h = Hs[tid * ld + c];
      if (h == R(0)) continue;
      if (h != R(1) || col >= 0) bad_row = 1;
      col = c;
    }
    if (col < 0) bad_row = 1;
    obs_s[tid] = (unsigned char)(col < 0 ? 0 : col);
    sel_ok = !bad_row;
  }
  __syncthreads();
  if (tid < d) {
    int who = -1, cnt = 0;
    for (int r = 0; r < m; ++r)
      if (obs_s[r] == tid) {
        who = r;
        ++cnt;
      }
    inv_s[tid] = (signed char)who;
    if (cnt > 1) sel_ok = 0;
  }
  if (!sel_ok) atomicAnd(&sel_s, 0);  // (several threads may clear the flag: atomically, so that the host build's ThreadSanitizer sees no race)
  __syncthreads();
  const bool hsel = sel_s != 0 && m <= d;
  AWG_FOR(e, (int)ntheta) g[e] = R(0);
  if (gm) AWG_FOR(e, (int)awg_model_grad_size(d, m)) gm[e] = R(0);
  R gForcing = R(0);  // Lorenz-96: thread 64 accumulates d ll / d F
  int st = 0;
  __syncthreads();

#ifndef CDKF_AWG_CUSTOM
  // ---- MLP drift (round 4): weights, work vectors and the tangent pass's images behind the vectors (awg_mlp_lds_reals) ---------------
  const int l1 = h1 | 1, l2 = h2 | 1, ldd = d | 1;
  R* const mx = lam2 + 64;
  R* const W2s = mx;                              // [h2][l1]
  R* const W1s = W2s + (long)h2 * l1;             // [h1][ldd]
  R* const W3s = W1s + (long)h1 * ldd;            // [d][l2]
  R* const mv = W3s + (long)d * l2;               // kAwgMlpVecs vectors of 64
  R *const va1 = mv, *const vd1 = mv + 64, *const va2 = mv + 128, *const vd2 = mv + 192, *const vz2b = mv + 256, *const vz1b = mv + 320,
    *const vs = mv + 384, *const vs2 = mv + 448, *const vtd = mv + 512, *const vtq = mv + 576, *const vr = mv + 640, *const vtdb = mv + 704,
    *const vtcb = mv + 768, *const vs2b = mv + 832, *const vsb = mv + 896, *const vd2b = mv + 960, *const vz2c = mv + 1024,
    *const vz1c = mv + 1088;
  R* const Ui = mv + 64L * kAwgMlpVecs;           // [h1][ldd]  U = D1 W1
  R* const Ti = Ui + 64L * ldd;                   // [h2][ldd]  T = W2 U  (tangent of z2)
  R* const C2 = Ti + 64L * ldd;                   // [h2][ldd]  cotangent of D2 T, then of T
  R* const C1 = C2 + 64L * ldd;                   // [h1][ldd]  cotangent of U; 'second': E1 = W2^T diag(d2) W3^T on the way
  const long oW1 = 0, ob1 = oW1 + (long)h1 * d, oW2 = ob1 + h1, ob2 = oW2 + (long)h2 * h1, oW3 = ob2 + h2, ob3 = oW3 + (long)d * h2;
  if (mlp) {
    AWG_FOR(e, h2 * h1) { const int p_ = fdiv(e, h1); W2s[p_ * l1 + (e - p_ * h1)] = th[oW2 + e]; }
    AWG_FOR(e, h1 * d) { const int q_ = fdiv(e, d); W1s[q_ * ldd + (e - q_ * d)] = th[oW1 + e]; }
    AWG_FOR(e, d * h2) { const int i_ = fdiv(e, h2); W3s[i_ * l2 + (e - i_ * h2)] = th[oW3 + e]; }
    __syncthreads();
  }
  // f(xv) into fv, F(xv) = W3 D2 W2 D1 W1 into F; leaves a1, d1, a2, d2, U, T for mlp_g / mlp_bwd.  Synchronises inside; the caller
  // synchronises before and after as for the other drifts.
  auto mlp_fwd = [&](const R* xv, R* F) {
    AWG_FOR(q_, h1) {
      R z = th[ob1 + q_];
      for (int i_ = 0; i_ < d; ++i_) z = rfma(W1s[q_ * ldd + i_], xv[i_], z);
      const R a1_ = rtanh_fast(z);
      va1[q_] = a1_;
      vd1[q_] = R(1) - a1_ * a1_;
    }
    __syncthreads();
    AWG_FOR(e, h1 * d) {
      const int q_ = fdiv(e, d), i_ = e - q_ * d;
      Ui[q_ * ldd + i_] = vd1[q_] * W1s[q_ * ldd + i_];
    }
    AWG_FOR(p_, h2) {
      R z = th[ob2 + p_];
      for (int q_ = 0; q_ < h1; ++q_) z = rfma(W2s[p_ * l1 + q_], va1[q_], z);
      const R a2_ = rtanh_fast(z);
      va2[p_] = a2_;
      vd2[p_] = R(1) - a2_ * a2_;
    }
    __syncthreads();
    AWG_FOR(e, h2 * d) {
      const int p_ = fdiv(e, d), i_ = e - p_ * d;
      R s_ = R(0);
      for (int q_ = 0; q_ < h1; ++q_) s_ = rfma(W2s[p_ * l1 + q_], Ui[q_ * ldd + i_], s_);
      Ti[p_ * ldd + i_] = s_;
    }
    __syncthreads();
    AWG_FOR(e, d * d) {
      const int i_ = fdiv(e, d), j_ = e - i_ * d;
      R s_ = R(0);
      for (int p_ = 0; p_ < h2; ++p_) s_ = rfma(W3s[i_ * l2 + p_] * vd2[p_], Ti[p_ * ldd + j_], s_);
      F[i_ * ld + j_] = s_;
    }
    if (tid < d) {
      R s_ = th[ob3 + tid];
      for (int p_ = 0; p_ < h2; ++p_) s_ = rfma(W3s[tid * l2 + p_], va2[p_], s_);
      fv[tid] = s_;
    }
  };
  // g = grad(div f) at the point mlp_fwd was last called for, into gout (oracle: MLPDrift.divgrad): td = G^T d2, s = G d1, s2 = -2 a2 d2 s,
  // tc = W2^T s2, tq = d1 (-2 a1 td + tc), g = W1^T tq with G = ((W1 W3)^T o W2) through its rank-d factors -- E1 = W2^T diag(d2) W3^T
  // gives td_q = sum_i W1[q][i] E1[q][i], the tangent T gives s_p = sum_i W3[i][p] T[p][i].  Synchronises before it reads and after.
  auto mlp_g = [&](R* gout) {
    __syncthreads();
    AWG_FOR(p_, h2) {
      R s_ = R(0);
      for (int i_ = 0; i_ < d; ++i_) s_ = rfma(W3s[i_ * l2 + p_], Ti[p_ * ldd + i_], s_);
      vs[p_] = s_;
      vs2[p_] = R(-2) * va2[p_] * vd2[p_] * s_;
    }
    AWG_FOR(e, h1 * d) {
      const int q_ = fdiv(e, d), i_ = e - q_ * d;
      R s_ = R(0);
      for (int p_ = 0; p_ < h2; ++p_) s_ = rfma(W2s[p_ * l1 + q_] * vd2[p_], W3s[i_ * l2 + p_], s_);
      C1[q_ * ldd + i_] = s_;
    }
    __syncthreads();
    AWG_FOR(q_, h1) {
      R td_ = R(0), tc_ = R(0);
      for (int i_ = 0; i_ < d; ++i_) td_ = rfma(W1s[q_ * ldd + i_], C1[q_ * ldd + i_], td_);
      for (int p_ = 0; p_ < h2; ++p_) tc_ = rfma(vs2[p_], W2s[p_ * l1 + q_], tc_);
      vtd[q_] = td_;
      vtq[q_] = vd1[q_] * rfma(R(-2) * va1[q_], td_, tc_);
    }
    __syncthreads();
    if (tid < d) {
      R s_ = R(0);
      for (int q_ = 0; q_ < h1; ++q_) s_ = rfma(W1s[q_ * ldd + tid], vtq[q_], s_);
      gout[tid] = s_;
    }
    __syncthreads();
  };
  // Reverse of a right-hand side through the network: the gradient of  lam . f(x) + <G2, F(x)>  (+ u . g(x) for 'second', u non-null;
  // mlp_g must have run for this point) with respect to x -- into xb -- and to the weights -- added to the trajectory's gradient g.
  // oracle/cdkf_oracle.py: the "mlp" branch of the drift's vector-Jacobian product and divgrad_vjp, same intermediate names.  Every
  // entry of g is always updated by the same thread (one AWG_FOR index space per weight block): program order, no atomics.
  auto M_of = [&](int p_, int q_) {  // M_pq = sum_i W3[i][p] W1[q][i]  ((W1 W3)^T, never stored)
    R s_ = R(0);
    for (int i_ = 0; i_ < d; ++i_) s_ = rfma(W3s[i_ * l2 + p_], W1s[q_ * ldd + i_], s_);
    return s_;
  };
  auto mlp_bwd = [&](const R* xv, const R* lamv, const R* G2, const R* u, R* xb) {
    __syncthreads();
    AWG_FOR(e, h2 * d) {  // c2 = W3^T G2
      const int p_ = fdiv(e, d), j_ = e - p_ * d;
      R s_ = R(0);
      for (int i_ = 0; i_ < d; ++i_) s_ = rfma(W3s[i_ * l2 + p_], G2[i_ * ld + j_], s_);
      C2[p_ * ldd + j_] = s_;
    }
    if (tid < d) g[ob3 + tid] += lamv[tid];
    __syncthreads();
    AWG_FOR(e, d * h2) {  // dW3 += lam a2^T + G2 (D2 T)^T
      const int i_ = fdiv(e, h2), p_ = e - i_ * h2;
      R s_ = R(0);
      for (int j_ = 0; j_ < d; ++j_) s_ = rfma(G2[i_ * ld + j_], Ti[p_ * ldd + j_], s_);
      g[oW3 + e] += rfma(lamv[i_], va2[p_], vd2[p_] * s_);
    }
    AWG_FOR(p_, h2) {  // a2b, z2b
      R s_ = R(0), w_ = R(0);
      for (int i_ = 0; i_ < d; ++i_) s_ = rfma(W3s[i_ * l2 + p_], lamv[i_], s_);
      for (int j_ = 0; j_ < d; ++j_) w_ = rfma(Ti[p_ * ldd + j_], C2[p_ * ldd + j_], w_);
      vz2b[p_] = vd2[p_] * rfma(R(-2) * va2[p_], w_, s_);
    }
    __syncthreads();
    AWG_FOR(e, h2 * d) {  // zt2 = D2 c2
      const int p_ = fdiv(e, d);
      C2[p_ * ldd + (e - p_ * d)] *= vd2[p_];
    }
    __syncthreads();
    AWG_FOR(e, h2 * h1) {  // dW2 += z2b a1^T + zt2 U^T
      const int p_ = fdiv(e, h1), q_ = e - p_ * h1;
      R s_ = vz2b[p_] * va1[q_];
      for (int j_ = 0; j_ < d; ++j_) s_ = rfma(C2[p_ * ldd + j_], Ui[q_ * ldd + j_], s_);
      g[oW2 + e] += s_;
    }
    AWG_FOR(p_, h2) g[ob2 + p_] += vz2b[p_];
    AWG_FOR(e, h1 * d) {  // c1 = W2^T zt2
      const int q_ = fdiv(e, d), j_ = e - q_ * d;
      R s_ = R(0);
      for (int p_ = 0; p_ < h2; ++p_) s_ = rfma(W2s[p_ * l1 + q_], C2[p_ * ldd + j_], s_);
      C1[q_ * ldd + j_] = s_;
    }
    __syncthreads();
    AWG_FOR(q_, h1) {  // a1b, z1b
      R s_ = R(0), w_ = R(0);
      for (int p_ = 0; p_ < h2; ++p_) s_ = rfma(W2s[p_ * l1 + q_], vz2b[p_], s_);
      for (int j_ = 0; j_ < d; ++j_) w_ = rfma(W1s[q_ * ldd + j_], C1[q_ * ldd + j_], w_);
      const R z_ = vd1[q_] * rfma(R(-2) * va1[q_], w_, s_);
      vz1b[q_] = z_;
      g[ob1 + q_] += z_;
    }
    __syncthreads();
    AWG_FOR(e, h1 * d) {  // dW1 += z1b x^T + D1 c1
      const int q_ = fdiv(e, d), i_ = e - q_ * d;
      g[oW1 + e] += rfma(vz1b[q_], xv[i_], vd1[q_] * C1[q_ * ldd + i_]);
    }
    if (tid < d) {
      R s_ = R(0);
      for (int q_ = 0; q_ < h1; ++q_) s_ = rfma(W1s[q_ * ldd + tid], vz1b[q_], s_);
      xb[tid] = s_;
    }
    __syncthreads();
    if (!u) return;
    // ---- 'second': the gradient of u . g(x) (divgrad_vjp) --------------------------------------------------------------------------
    AWG_FOR(q_, h1) {
      R r_ = R(0);
      for (int l_ = 0; l_ < d; ++l_) r_ = rfma(W1s[q_ * ldd + l_], u[l_], r_);
      const R a1_ = va1[q_], d1_ = vd1[q_];
      vr[q_] = r_;
      vtdb[q_] = R(-2) * a1_ * d1_ * r_;
      vtcb[q_] = d1_ * r_;
      vz1c[q_] = r_ * (R(-2) * a1_ * vtq[q_] - R(2) * d1_ * d1_ * vtd[q_]);
    }
    AWG_FOR(e, h1 * d) {
      const int q_ = fdiv(e, d);
      g[oW1 + e] += vtq[q_] * u[e - q_ * d];
    }
    __syncthreads();
    AWG_FOR(p_, h2) {
      R d2b_ = R(0), s2b_ = R(0);
      for (int q_ = 0; q_ < h1; ++q_) {
        const R w_ = W2s[p_ * l1 + q_];
        d2b_ = rfma(w_ * M_of(p_, q_), vtdb[q_], d2b_);
        s2b_ = rfma(w_, vtcb[q_], s2b_);
      }
      const R a2_ = va2[p_], d2_ = vd2[p_];
      vs2b[p_] = s2b_;
      vsb[p_] = R(-2) * a2_ * d2_ * s2b_;
      vd2b[p_] = d2b_;
      vz2c[p_] = s2b_ * (R(-2) * vs[p_]) * d2_ * (R(1) - R(3) * a2_ * a2_) + d2b_ * (R(-2) * a2_ * d2_);
    }
    AWG_FOR(e, h2 * h1) {
      const int p_ = fdiv(e, h1);
      g[oW2 + e] += vs2[p_] * vtcb[e - p_ * h1];
    }
    __syncthreads();
    AWG_FOR(q_, h1) {  // z1_b += (s_b G)(-2 a1 d1)
      R s_ = R(0);
      for (int p_ = 0; p_ < h2; ++p_) s_ = rfma(vsb[p_] * W2s[p_ * l1 + q_], M_of(p_, q_), s_);
      vz1c[q_] += s_ * (R(-2) * va1[q_] * vd1[q_]);
    }
    AWG_FOR(e, h2 * h1) {  // dW2 += G_b o M + z2_b a1^T,  G_b = d2 td_b^T + s_b d1^T
      const int p_ = fdiv(e, h1), q_ = e - p_ * h1;
      const R gb_ = rfma(vd2[p_], vtdb[q_], vsb[p_] * vd1[q_]);
      g[oW2 + e] += rfma(gb_, M_of(p_, q_), vz2c[p_] * va1[q_]);
    }
    AWG_FOR(e, h1 * d) {  // dW1 += (G_b o W2)^T W3^T
      const int q_ = fdiv(e, d), i_ = e - q_ * d;
      R s_ = R(0);
      for (int p_ = 0; p_ < h2; ++p_) s_ = rfma(rfma(vd2[p_], vtdb[q_], vsb[p_] * vd1[q_]) * W2s[p_ * l1 + q_], W3s[i_ * l2 + p_], s_);
      g[oW1 + e] += s_;
    }
    AWG_FOR(e, d * h2) {  // dW3 += W1^T (G_b o W2)^T
      const int i_ = fdiv(e, h2), p_ = e - i_ * h2;
      R s_ = R(0);
      for (int q_ = 0; q_ < h1; ++q_) s_ = rfma(W1s[q_ * ldd + i_], rfma(vd2[p_], vtdb[q_], vsb[p_] * vd1[q_]) * W2s[p_ * l1 + q_], s_);
      g[oW3 + e] += s_;
    }
    AWG_FOR(p_, h2) g[ob2 + p_] += vz2c[p_];
    __syncthreads();
    AWG_FOR(q_, h1) {  // z1_b += (W2^T z2_b) d1
      R s_ = R(0);
      for (int p_ = 0; p_ < h2; ++p_) s_ = rfma(W2s[p_ * l1 + q_], vz2c[p_], s_);
      const R z_ = rfma(s_, vd1[q_], vz1c[q_]);
      vz1c[q_] = z_;
      g[ob1 + q_] += z_;
    }
    __syncthreads();
    AWG_FOR(e, h1 * d) {
      const int q_ = fdiv(e, d);
      g[oW1 + e] += vz1c[q_] * xv[e - q_ * d];
    }
    if (tid < d) {
      R s_ = xb[tid];
      for (int q_ = 0; q_ < h1; ++q_) s_ = rfma(W1s[q_ * ldd + tid], vz1c[q_], s_);
      xb[tid] = s_;
    }
    __syncthreads();
  };
#endif
  // what a drift given as source sees beside x and theta: the inputs row of the interval in hand and the time of the evaluation in hand
  // (every thread the same values; the registry drifts ignore both)
#ifdef CDKF_AWG_CUSTOM
  R cur_u[CDKF_AWG_CUSTOM_DU > 0 ? CDKF_AWG_CUSTOM_DU : 1];
  cur_u[0] = R(0);
#endif
  R cur_t = R(0);
  (void)cur_t;
  // ---- drift: dense Jacobian F(x) into a slot, f(x) into fv; x in LDS (synchronised by the caller before AND after) ------------
  auto drift_eval = [&](const R* xv, R* F) {
#ifdef CDKF_AWG_CUSTOM
    if (custom) {  // jacfwd with the directions spread over the workgroup: thread j carries e_j
      AWG_FOR(j, d) awg_custom_column<R>(th, xv, j, F, ld, j == 0 ? fv : (R*)nullptr, cur_u, cur_t);
      return;
    }
#endif
#ifndef CDKF_AWG_CUSTOM
    if (mlp) {
      mlp_fwd(xv, F);
      return;
    }
#endif
    if (lin)  // (Lorenz-96: the four non-zeros of a row of the Jacobian are formed where they are used -- l96_FPs, l96_LamF below)
    rows2d(d, d,
           [&](int i, int j) {
             if (lin) return th[i * d + j];
             // Lorenz-96: f_i = (x_{i+1} - x_{i-2}) x_{i-1} - x_i + F
             const int ip1 = (i + 1 >= d) ? 0 : i + 1, im1 = (i == 0) ? d - 1 : i - 1, im2 = (im1 == 0) ? d - 1 : im1 - 1;
             R v = R(0);
             if (j == ip1) v += xv[im1];
             if (j == im2) v -= xv[im1];
             if (j == im1) v += xv[ip1] - xv[im2];
             if (j == i) v -= R(1);
             return v;
           },
           [&](int i, int j, R v) { F[i * ld + j] = v; });
    if (tid < d) {
      const int i = tid;
      R f;
      if (lin) {
        f = th[d * d + i] + dot(d, [&](int k) { return th[i * d + k]; }, [&](int k) { return xv[k]; });
      } else {
        const int ip1 = (i + 1 >= d) ? 0 : i + 1, im1 = (i == 0) ? d - 1 : i - 1, im2 = (im1 == 0) ? d - 1 : im1 - 1;
        f = rfma(xv[ip1] - xv[im2], xv[im1], th[0] - xv[i]);
      }
      fv[i] = f;
    }
  };
  // Lorenz-96: F has four entries per row (F_i,i+1 = x_{i-1}, F_i,i-2 = -x_{i-1}, F_i,i-1 = x_{i+1} - x_{i-2}, F_ii = -1) -- the two dense
  // products of a stage with it are four-term stencils over the LDS image of the other factor (they were 2 x 6.5 k of a stage's ~ 22 k
  // cycles as d^3 products against a dense copy of F)
#ifdef CDKF_AWG_CUSTOM
  constexpr bool mlp = false;
#endif
  const bool l96 = !lin && !custom && !mlp;
  auto wrap = [&](int i) { return i < 0 ? i + d : (i >= d ? i - d : i); };
  auto l96_FPs = [&](const R* xv, const R* Ps_, R* A_) {  // A = F(x) Ps
    rows2d(d, d,
           [&](int i, int j) {
             const int ip1 = wrap(i + 1), im1 = wrap(i - 1), im2 = wrap(i - 2);
             R v = -xv[im1] * Ps_[im2 * ld + j];
             v = rfma(xv[ip1] - xv[im2], Ps_[im1 * ld + j], v);
             v -= Ps_[i * ld + j];
             return rfma(xv[im1], Ps_[ip1 * ld + j], v);
           },
           [&](int i, int j, R v) { A_[i * ld + j] = v; });
  };
  auto l96_LamF = [&](const R* xv, const R* Lam_, R* G_) {  // G = Lam F(x): column j of F holds rows j-1, j+2, j+1, j
    rows2d(d, d,
           [&](int i, int j) {
             const int jp1 = wrap(j + 1), jp2 = wrap(j + 2), jm1 = wrap(j - 1), jm2 = wrap(j - 2);
             R v = Lam_[i * ld + jm1] * xv[jm2];
             v = rfma(-Lam_[i * ld + jp2], xv[jp1], v);
             v = rfma(Lam_[i * ld + jp1], xv[jp2] - xv[jm1], v);
             return v - Lam_[i * ld + j];
           },
           [&](int i, int j, R v) { G_[i * ld + j] = v; });
  };
#if defined(CDKF_AWG_CUSTOM) && CDKF_AWG_CUSTOM_SECOND
  // g = grad(div f) at xv into gout (an LDS vector): one nested-dual evaluation per pair (i, k) into a free slot, column sums behind a
  // barrier (deterministic); synchronises before and after the sums
  auto custom_g = [&](const R* xv, R* gout) {
    R* W = slot(7);
    AWG_FOR(e, d * d) {
      const int i = fdiv(e, d), kk = e - i * d;
      W[i * ld + kk] = awg_custom_divpair<R>(th, xv, i, kk, cur_u, cur_t);
    }
    __syncthreads();
    if (tid < d) {
      R sg = R(0);
      for (int i = 0; i < d; ++i) sg += W[i * ld + tid];
      gout[tid] = sg;
    }
    __syncthreads();
  };
#endif
  // ---- lower Cholesky factors of one or two n x n matrices in lockstep (in place, lower triangles), right-looking in PANELS of eight
  // columns, two barriers per panel: (A) every thread of the system's wavefront factorises the 8 x 8 block on the diagonal in
  // registers (redundantly: broadcast reads, no exchange) and solves its own row of the panel against it; (B) the trailing matrix
  // takes the panel's rank-8 update as a tiled product.  iv0 / iv1 (LDS vectors) receive 1 / L_pp for the substitutions.
  // (the rank-one form -- a barrier, a pivot read, a division and two LDS round trips per COLUMN -- cost 138 k cycles per update at
  //  m = 40, the three substitutions 260 k: a third of the reverse step)
  constexpr int NB = 8;
  auto chol2 = [&](R* A0, R* iv0, R* A1, R* iv1, int nn) {
    const int sys = tid >> 6, ln = tid & 63;
    R* As = sys == 0 ? A0 : A1;
    R* ivs = sys == 0 ? iv0 : iv1;
    const bool act = sys == 0 || (sys == 1 && A1);
    for (int p0 = 0; p0 < nn; p0 += NB) {
      __syncthreads();
      const int wdt = (nn - p0 < NB) ? nn - p0 : NB;
      AWG_TICK2(12)
      if (act) {
        R L[NB][NB], ivl[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r)
#pragma unroll
          for (int c = 0; c <= r; ++c) L[r][c] = (r < wdt) ? As[(p0 + r) * ld + p0 + c] : (r == c ? R(1) : R(0));
        // (wave-synchronous: every lane of the wavefront has read the block before the lanes that own its rows write them back below.
        //  The hardware runs the wavefront's reads before its writes anyway; the barrier -- no instruction on the GPU -- says so to the
        //  compiler and to the host build under ThreadSanitizer, tests/test_hostsim.py)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < NB; ++c) {
          R sd = L[c][c];
#pragma unroll
          for (int kk = 0; kk < c; ++kk) sd = rfma(-L[c][kk], L[c][kk], sd);
          if (!(sd > R(0))) st |= kStatusNotPd;
          const R rinv = awg_rsqrt(sd);  // (a square root and a division in fp64 are ~ 100 instructions on the panel's critical path)
          L[c][c] = sd * rinv;
          ivl[c] = rinv;
#pragma unroll
          for (int r = c + 1; r < NB; ++r) {
            R tt = L[r][c];
#pragma unroll
            for (int kk = 0; kk < c; ++kk) tt = rfma(-L[r][kk], L[c][kk], tt);
            L[r][c] = tt * rinv;
          }
        }
        const int i = p0 + ln;
        if (i < nn) {
          if (ln < wdt) {  // a row of the block itself (select chain: no run-time register index)
#pragma unroll
            for (int r = 0; r < NB; ++r)
              if (r == ln) {
#pragma unroll
                for (int c = 0; c <= r; ++c) As[i * ld + p0 + c] = L[r][c];
                ivs[i] = ivl[r];
              }
          } else {  // a row below it: x L_dd^T = a
            R x[NB];
#pragma unroll
            for (int c = 0; c < NB; ++c) {
              R tt = (c < wdt) ? As[i * ld + p0 + c] : R(0);
#pragma unroll
              for (int kk = 0; kk < c; ++kk) tt = rfma(-x[kk], L[c][kk], tt);
              x[c] = tt * ivl[c];
            }
#pragma unroll
            for (int c = 0; c < NB; ++c)
              if (c < wdt) As[i * ld + p0 + c] = x[c];
          }
        }
      }
      AWG_TICK2(13)
      __syncthreads();
      AWG_TICK2(14)
      const int rem = nn - p0 - NB, q0 = p0 + NB;
      if (rem > 0) {
        gemm2(rem, rem, A1 ? rem : 0, NB, [&](int sy, int i, int kk) { return (sy ? A1 : A0)[(q0 + i) * ld + p0 + kk]; },
              [&](int sy, int kk, int j) { return (sy ? A1 : A0)[(q0 + j) * ld + p0 + kk]; },
              [&](int sy, int i, int j, R v) { if (j <= i) (sy ? A1 : A0)[(q0 + i) * ld + q0 + j] -= v; });
      }
      AWG_TICK2(15)
    }
    __syncthreads();
  };
  // (L L^T) X = B in place for the columns of one or two right-hand-side matrices B [nn][ld] (each with its own factor), in blocks of
  // eight unknowns: (A) a thread per column substitutes through the 8 x 8 triangle in registers, (B) the remaining rows take the
  // block's contribution as a tiled product; two barriers per block and direction.
  constexpr int NBS = CDKF_AWG_SOLVE_PANEL;  // unknowns per substitution panel (the factorisation's panels stay at eight: its diagonal block lives in registers)
  auto solve2 = [&](const R* La, const R* iva, R* Ba, int nca, const R* Lb, const R* ivb, R* Bb, int ncb, int nn) {
    const int sys = tid >> 6, c = tid & 63;
    const R* Ls = sys == 0 ? La : Lb;
    const R* ivs = sys == 0 ? iva : ivb;
    R* Bs = sys == 0 ? Ba : Bb;
    const bool act = (sys == 0 || (sys == 1 && Bb)) && c < (sys == 0 ? nca : ncb);
    for (int p0 = 0; p0 < nn; p0 += NBS) {  // forward: L y = b
      __syncthreads();
      const int wdt = (nn - p0 < NBS) ? nn - p0 : NBS;
      if (act) {
        R x[NBS];
#pragma unroll
        for (int r = 0; r < NBS; ++r) {
          R tt = (r < wdt) ? Bs[(p0 + r) * ld + c] : R(0);
#pragma unroll
          for (int kk = 0; kk < r; ++kk) tt = rfma(-((r < wdt) ? Ls[(p0 + r) * ld + p0 + kk] : R(0)), x[kk], tt);
          x[r] = (r < wdt) ? tt * ivs[p0 + r] : R(0);
        }
#pragma unroll
        for (int r = 0; r < NBS; ++r)
          if (r < wdt) Bs[(p0 + r) * ld + c] = x[r];
      }
      __syncthreads();
      const int rem = nn - p0 - NBS, q0 = p0 + NBS;
      if (rem > 0) {
        gemm2(rem, nca, Bb ? ncb : 0, NBS, [&](int sy, int i, int kk) { return (sy ? Lb : La)[(q0 + i) * ld + p0 + kk]; },
              [&](int sy, int kk, int j) { return (sy ? Bb : Ba)[(p0 + kk) * ld + j]; },
              [&](int sy, int i, int j, R v) { (sy ? Bb : Ba)[(q0 + i) * ld + j] -= v; });
      }
    }
    for (int p0 = ((nn - 1) / NBS) * NBS; p0 >= 0; p0 -= NBS) {  // backward: L^T x = y
      __syncthreads();
      const int wdt = (nn - p0 < NBS) ? nn - p0 : NBS;
      if (act) {
        R x[NBS];
#pragma unroll
        for (int r = NBS - 1; r >= 0; --r) {
          R tt = (r < wdt) ? Bs[(p0 + r) * ld + c] : R(0);
#pragma unroll
          for (int kk = r + 1; kk < NBS; ++kk) tt = rfma(-((kk < wdt) ? Ls[(p0 + kk) * ld + p0 + r] : R(0)), x[kk], tt);
          x[r] = (r < wdt) ? tt * ivs[p0 + r] : R(0);
        }
#pragma unroll
        for (int r = 0; r < NBS; ++r)
          if (r < wdt) Bs[(p0 + r) * ld + c] = x[r];
      }
      __syncthreads();
      if (p0 > 0) {
        gemm2(p0, nca, Bb ? ncb : 0, wdt, [&](int sy, int i, int kk) { return (sy ? Lb : La)[(p0 + kk) * ld + i]; },
              [&](int sy, int kk, int j) { return (sy ? Bb : Ba)[(p0 + kk) * ld + j]; },
              [&](int sy, int i, int j, R v) { (sy ? Bb : Ba)[i * ld + j] -= v; });
      }
    }
    __syncthreads();
  };
  // A <- A + 0.5 (T + T^T) for d x d matrices (T fully written and synchronised); A == nullptr: T <- 0.5 (T + T^T) is not needed here
  auto add_sym = [&](R* A, const R* Tm, bool assign) {
    rows2d(d, d,
            [&](int i, int j) {
              const R sy = R(0.5) * (Tm[i * ld + j] + Tm[j * ld + i]);
              return assign ? sy : A[i * ld + j] + sy;
            },
            [&](int i, int j, R v) {
              A[i * ld + j] = v;
            });
    __syncthreads();
  };
  // A <- 0.5 (A + A^T) through a scratch slot
  auto symmetrize = [&](R* A, R* tmp) {
    rows2d(d, d, [&](int i, int j) { return A[i * ld + j]; }, [&](int i, int j, R v) { tmp[i * ld + j] = v; });
    __syncthreads();
    add_sym(A, tmp, true);
  };

  // ---- Runge-Kutta stages of one step from (x0, P0s) ------------------------------------------------------------------------------
  // The covariance parts of the six slopes and of the six stage cotangents of the step in hand stay in the REGISTERS of the thread
  // that owns the entry (NE entries per thread on the d x d map): a stage's combination is a sum over all six with the coefficient
  // zero where the tableau has none, a stage's result is written through a select -- every register index is a compile-time constant
  // although the stage loops stay rolled.  (In a global scratch they cost an L2 round trip per batch and pass: ~40 % of a stage.)
  R kP[6][NE], yP[6][NE], gQacc[NE];  // (gQacc: d ll / d (L Qc L^T) of the owned entries, accumulated over the whole sweep)
#pragma unroll
  for (int s6 = 0; s6 < 6; ++s6)
#pragma unroll
    for (int u = 0; u < NE; ++u) kP[s6][u] = yP[s6][u] = R(0);
#pragma unroll
  for (int u = 0; u < NE; ++u) gQacc[u] = R(0);
  auto slots = [&](auto&& value, auto&& store) __attribute__((always_inline)) {
#pragma unroll
    for (int u0 = 0; u0 < NE; u0 += 4) {
      R v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = map_d.i0 + (u0 + u) * map_d.rs;
        v[u] = (i < d) ? value(i, map_d.j, u0 + u) : R(0);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = map_d.i0 + (u0 + u) * map_d.rs;
        if (i < d) store(i, map_d.j, u0 + u, v[u]);
      }
    }
  };
  auto rka = [&](int r, int c) { return (c < r && c < 5) ? a.rk.a[r][c] : R(0); };  // (rows / columns the method does not have: zero)
  // k_P of stage sv from the product A = F Ps left in LDS (pending: the slope is formed where it is first needed -- no barrier of its own)
  auto take_slope = [&](int sv, const R* A) __attribute__((always_inline)) {
    slots([&](int i, int j, int) { return (A[i * ld + j] + A[j * ld + i]) + LQL[i * d + j]; },
          [&](int, int, int u, R v) {
#pragma unroll
            for (int s6 = 0; s6 < 6; ++s6) kP[s6][u] = (s6 == sv) ? v : kP[s6][u];
          });
  };
  // stage value si into (xs, Ps): y0 + dt sum_{j < si} a_ij k_j
  auto stage_value = [&](int si, const R* P0s, R* Ps, R dt) __attribute__((always_inline)) {
    R cf[6];
#pragma unroll
    for (int s6 = 0; s6 < 6; ++s6) cf[s6] = rka(si, s6);
    slots([&](int i, int j, int u) {
            R s2 = R(0);
#pragma unroll
            for (int s6 = 0; s6 < 5; ++s6) s2 = rfma(cf[s6], kP[s6][u], s2);
            return rfma(dt, s2, P0s[i * ld + j]);
          },
          [&](int i, int j, int, R v) { Ps[i * ld + j] = v; });
    if (tid < d) {
      R s2 = R(0);
      for (int jj = 0; jj < si; ++jj) s2 = rfma(a.rk.a[si][jj], km[64 * jj + tid], s2);
      xs[tid] = rfma(dt, s2, x0[tid]);
    }
    __syncthreads();
  };
  // (A: a free slot for F Ps; three barriers per stage)
  auto stages_fwd = [&](const R* P0s, R* Ps, R* F, R* A, R dt, R tstep) {
    for (int si = 0; si < nst; ++si) {
      if (si) take_slope(si - 1, A);
      stage_value(si, P0s, Ps, dt);
      cur_t = rfma(rk_stage_c(a.rk, si), dt, tstep);
      drift_eval(xs, F);
      __syncthreads();
      if (l96)
        l96_FPs(xs, Ps, A);
      else
        gemm(d, d, d, [&](int i, int k) { return F[i * ld + k]; }, [&](int k, int j) { return Ps[k * ld + j]; },
             [&](int i, int j, R v) { A[i * ld + j] = v; });
#if defined(CDKF_AWG_CUSTOM) && CDKF_AWG_CUSTOM_SECOND
      if (second) {  // dm/dt = f + 0.5 Ps g  (inference_ekf.py:108-116)
        custom_g(xs, g1);
        if (tid < d) km[64 * si + tid] = rfma(R(0.5), dot(d, [&](int kk) { return Ps[tid * ld + kk]; }, [&](int kk) { return g1[kk]; }), fv[tid]);
      } else
#elif !defined(CDKF_AWG_CUSTOM)
      if (second) {  // (MLP) dm/dt = f + 0.5 Ps g  (inference_ekf.py:108-116)
        mlp_g(g1);
        if (tid < d) km[64 * si + tid] = rfma(R(0.5), dot(d, [&](int kk) { return Ps[tid * ld + kk]; }, [&](int kk) { return g1[kk]; }), fv[tid]);
      } else
#endif
      if (tid < d) {
        R kmv = fv[tid];
        if (a.ukf && l96)  // the unscented filter's curvature term of the quadratic drift (oracle: ukf_curvature)
          kmv += Ps[wrap(tid + 1) * ld + wrap(tid - 1)] - Ps[wrap(tid - 2) * ld + wrap(tid - 1)];
        km[64 * si + tid] = kmv;
      }
      __syncthreads();
    }
    take_slope(nst - 1, A);
  };
  // y <- y + dt sum_i b_i k_i  (x0 and the matrix in P0s)
  auto step_end = [&](R* P0s, R dt) {
    R cf[6];
#pragma unroll
    for (int s6 = 0; s6 < 6; ++s6) cf[s6] = (s6 < nst) ? a.rk.b[s6] : R(0);
    slots([&](int i, int j, int u) {
            R s2 = R(0);
#pragma unroll
            for (int s6 = 0; s6 < 6; ++s6) s2 = rfma(cf[s6], kP[s6][u], s2);
            return rfma(dt, s2, P0s[i * ld + j]);
          },
          [&](int i, int j, int, R v) { P0s[i * ld + j] = v; });
    if (tid < d) {
      R s2 = R(0);
      for (int si = 0; si < nst; ++si) s2 = rfma(a.rk.b[si], km[64 * si + tid], s2);
      x0[tid] = rfma(dt, s2, x0[tid]);
    }
    __syncthreads();
  };

  const R* tp = a.t + n * a.t_sn;
  const R* yp = a.y + n * a.y_sn;
#ifdef CDKF_AWG_PROFILE
  awg_last = clock64();
#endif
  for (long k = a.T - 1; k >= 0; --k) {
    AWG_TICK(0)
    // ================= (1) measurement update + log-likelihood term at k, reversed ===============================================
    R* Pp = slot(1);   // predicted covariance
    R* HP = slot(2);   // H P            [m][d]
    R* S = slot(3);    // H P H^T + R    [m][m]; later (X Pbar) X^T, then Sbar
    R* L1 = slot(4);   // chol(S); later Kb [m][d]
    R* Si = slot(5);   // S^-1; later Sbar H [m][d]
    R* L2 = slot(6);   // chol(sym(S) + 1e-9 I); later X = W2 H P [m][d]
    R* T1 = slot(8);   // X Pbar [m][d]; later Ub = W2 Kb [m][d]
    rows2d(d, d,
            [&](int i, int j) {
              return (k == 0) ? R(0.5) * ((par + a.o_P0)[i * d + j] + (par + a.o_P0)[j * d + i])
                              : a.pP[n * a.P_sn + (k - 1) * a.P_sk + (long)(i * d + j) * a.P_si];
            },
            [&](int i, int j, R v) {
              Pp[i * ld + j] = v;
            });
    if (tid < d) x0[tid] = (k == 0) ? (par + a.o_m0)[tid] : a.pm[n * a.m_sn + (k - 1) * a.m_sk + tid * a.m_si];
    __syncthreads();
    if (hsel) {
      rows2d(m, d, [&](int r, int c) { return Pp[obs_s[r] * ld + c]; }, [&](int r, int c, R v) { HP[r * ld + c] = v; });
      rows2d(m, m, [&](int r, int c) { return Pp[obs_s[r] * ld + obs_s[c]] + Rm[r * m + c]; }, [&](int r, int c, R v) { S[r * ld + c] = v; });
      if (tid < m) vv[tid] = yp[k * a.y_sk + tid * a.y_si] - (hb[tid] + x0[obs_s[tid]]);
    } else {
      gemm(m, d, d, [&](int r, int kk) { return Hs[r * ld + kk]; }, [&](int kk, int c) { return Pp[kk * ld + c]; },
           [&](int r, int c, R v) { HP[r * ld + c] = v; });
      if (tid < m)
        vv[tid] = yp[k * a.y_sk + tid * a.y_si] - (hb[tid] + dot(d, [&](int kk) { return Hs[tid * ld + kk]; }, [&](int kk) { return x0[kk]; }));
      __syncthreads();
      gemm(m, m, d, [&](int r, int kk) { return HP[r * ld + kk]; }, [&](int kk, int c) { return Hs[c * ld + kk]; },
           [&](int r, int c, R v) { S[r * ld + c] = v + Rm[r * m + c]; });
    }
    __syncthreads();
    // psd_solve's matrix is used twice (X = W2 H P, Ub = W2 Kb, W2 = (sym(S) + 1e-9 I)^-1): its inverse is formed ONCE, in lockstep with
    // S^-1 (which the log-likelihood's gradient needs anyway), and the two applications are dense products -- a third substitution
    // sweep with one system (49 k cycles at m = 40, half the workgroup idle) against two products of 6.5 k
    R* W2 = slot(7);   // (sym(S) + 1e-9 I)^-1 [m][m]; later X Ub^T [m][m], then Ub^T H [d][d]
    R* X = slot(6);    // W2 H P [m][d], over the dead factor L2
    rows2d(m, m, [&](int r, int c) { return R(0.5) * (S[r * ld + c] + S[c * ld + r]) + (r == c ? R(1e-9) : R(0)); },
           [&](int r, int c, R v) {
             L1[r * ld + c] = S[r * ld + c];
             L2[r * ld + c] = v;
             Si[r * ld + c] = (r == c) ? R(1) : R(0);
             W2[r * ld + c] = (r == c) ? R(1) : R(0);
           });
    AWG_TICK(1)  // loads, H P, S
    chol2(L1, g1, L2, g2, m);               // (g1, g2: the reciprocal diagonals; free vectors during the update)
    AWG_TICK(2)  // factorisations
    solve2(L1, g1, Si, m, L2, g2, W2, m, m);
    gemm(m, d, m, [&](int r, int kk) { return W2[r * ld + kk]; }, [&](int kk, int c) { return HP[kk * ld + c]; },
         [&](int r, int c, R v) { X[r * ld + c] = v; });
    AWG_TICK(3)  // S^-1, W2, X
    symmetrize(Pb, T1);  // (synchronises: X is complete behind it)
    gemm(m, d, d, [&](int r, int kk) { return X[r * ld + kk]; }, [&](int kk, int c) { return Pb[kk * ld + c]; },
         [&](int r, int c, R v) { T1[r * ld + c] = v; });
    if (tid < m) {  // w = S^-1 v;  vbar = X mbar - w
      const R w = dot(m, [&](int c) { return Si[tid * ld + c]; }, [&](int c) { return vv[c]; });
      const R s2 = dot(d, [&](int c) { return X[tid * ld + c]; }, [&](int c) { return mb[c]; });
      wv[tid] = w;
      vb[tid] = s2 - w;
    }
    __syncthreads();
    // Kb = v mbar^T - 2 S (X Pbar)   (cotangent of K^T), over the dead factor L1
    R* Kb = L1;
    gemm(m, d, m, [&](int r, int kk) { return S[r * ld + kk]; }, [&](int kk, int c) { return T1[kk * ld + c]; },
         [&](int r, int c, R v) { Kb[r * ld + c] = rfma(R(-2), v, vv[r] * mb[c]); });
    __syncthreads();
    AWG_TICK(4)  // X Pbar, w, vbar, Kb
    // Sbar = -(X Pbar) X^T + w w^T / 2 - S^-1 / 2 - sym(X Ub^T): the first product over the dead S; Ub = W2 Kb over the then dead
    // X Pbar; X Ub^T over the then dead W2
    gemm(m, m, d, [&](int r, int kk) { return T1[r * ld + kk]; }, [&](int kk, int c) { return X[c * ld + kk]; },
         [&](int r, int c, R v) { S[r * ld + c] = v; });
    __syncthreads();
    R* Ub = T1;
    gemm(m, d, m, [&](int r, int kk) { return W2[r * ld + kk]; }, [&](int kk, int c) { return Kb[kk * ld + c]; },
         [&](int r, int c, R v) { Ub[r * ld + c] = v; });
    __syncthreads();
    AWG_TICK(5)  // Ub
    R* XU = W2;
    gemm(m, m, d, [&](int r, int kk) { return X[r * ld + kk]; }, [&](int kk, int c) { return Ub[c * ld + kk]; },
         [&](int r, int c, R v) { XU[r * ld + c] = v; });
    __syncthreads();
    R* Sbar = S;
    rows2d(m, m,
            [&](int r, int c) {
              return -S[r * ld + c] + R(0.5) * wv[r] * wv[c] - R(0.5) * Si[r * ld + c] - R(0.5) * (XU[r * ld + c] + XU[c * ld + r]);
            },
            [&](int r, int c, R v) {
              Sbar[r * ld + c] = v;
            });
    __syncthreads();
    if (gm) {  // model block: dR += Sbar; dH += 2 Sbar (H P) - vbar m^T + Ub P; dbias -= vbar   (a thread owns the same tiles in both products)
      rows2d(m, m, [&](int r, int c) { return gR[r * m + c] + Sbar[r * ld + c]; }, [&](int r, int c, R v) { gR[r * m + c] = v; });
      gemm(m, d, m, [&](int r, int kk) { return Sbar[r * ld + kk]; }, [&](int kk, int c) { return HP[kk * ld + c]; },
           [&](int r, int c, R v) { gH[r * d + c] += R(2) * v - vb[r] * x0[c]; });
      gemm(m, d, d, [&](int r, int kk) { return Ub[r * ld + kk]; }, [&](int kk, int c) { return Pp[kk * ld + c]; },
           [&](int r, int c, R v) { gH[r * d + c] += v; });
      if (tid < m) gBias[tid] -= vb[tid];
    }
    R* UH = XU;  // Ub^T H [d][d], over the dead X Ub^T
    R mbn = R(0);
    if (hsel) {  // Pbar <- Pbar + sym(Ub^T H) + H^T Sbar H: column j of Ub^T H is row inv[j] of Ub (or zero), H^T Sbar H scatters Sbar
      rows2d(d, d, [&](int i, int j) { return inv_s[j] >= 0 ? Ub[inv_s[j] * ld + i] : R(0); }, [&](int i, int j, R v) { UH[i * ld + j] = v; });
      rows2d(m, m, [&](int r, int c) { return Pb[obs_s[r] * ld + obs_s[c]] + Sbar[r * ld + c]; },
             [&](int r, int c, R v) { Pb[obs_s[r] * ld + obs_s[c]] = v; });
      if (tid < d) mbn = mb[tid] - (inv_s[tid] >= 0 ? vb[inv_s[tid]] : R(0));
    } else {
      R* SH = Si;  // Sbar H [m][d], over the dead S^-1
      gemm(m, d, m, [&](int r, int kk) { return Sbar[r * ld + kk]; }, [&](int kk, int c) { return Hs[kk * ld + c]; },
           [&](int r, int c, R v) { SH[r * ld + c] = v; });
      // Pbar <- Pbar + sym(Ub^T H) + H^T Sbar H
      gemm(d, d, m, [&](int i, int kk) { return Ub[kk * ld + i]; }, [&](int kk, int j) { return Hs[kk * ld + j]; },
           [&](int i, int j, R v) { UH[i * ld + j] = v; });
      __syncthreads();
      gemm(d, d, m, [&](int i, int kk) { return Hs[kk * ld + i]; }, [&](int kk, int j) { return SH[kk * ld + j]; },
           [&](int i, int j, R v) { Pb[i * ld + j] += v; });
      if (tid < d) mbn = mb[tid] - dot(m, [&](int r) { return Hs[r * ld + tid]; }, [&](int r) { return vb[r]; });  // mbar <- mbar - H^T vbar
    }
    __syncthreads();
    if (tid < d) mb[tid] = mbn;
    add_sym(Pb, UH, false);
    AWG_TICK(6)  // Sbar, model block, Pbar, mbar
    if (k == 0) break;

    // ================= (2) predict k-1 -> k: the Runge-Kutta steps of the interval, reversed ========================================
    // (the slope / cotangent registers start every interval at zero: written through selects, they would otherwise count as live
    //  across the update's adjoint above -- 24 NE registers the products and factorisations there then spill around)
#pragma unroll
    for (int s6 = 0; s6 < 6; ++s6)
#pragma unroll
      for (int u = 0; u < NE; ++u) kP[s6][u] = yP[s6][u] = R(0);
    const R t0 = tp[(k - 1) * a.t_sk], t1 = tp[k * a.t_sk];
#ifdef CDKF_AWG_CUSTOM
    for (int iu = 0; iu < CDKF_AWG_CUSTOM_DU; ++iu)  // u = inputs[t0_idx] of the interval k-1 -> k (inference_ekf.py:277)
      cur_u[iu] = a.u ? a.u[n * a.u_sn + (k - 1) * a.u_sk + iu * a.u_si] : R(0);
#endif
    // an adaptive solve: the forward (workgroup) sweep logged the step sizes it accepted in this interval; the reverse of the solve
    // treats them as constants -- the controller's factor carries no derivative, as in the reference's reverse mode through diffrax
    const R* dtl = a.dtlog ? a.dtlog + (n * (a.T - 1) + (k - 1)) * (1 + a.dtlog_cap) : nullptr;
    long Ssteps = 0;
    if (dtl) {
      Ssteps = (long)dtl[0];
      if (Ssteps > a.dtlog_cap) {  // more accepted steps than the log holds: the gradient of this trajectory is not valid
        Ssteps = a.dtlog_cap;
        st |= kStatusMaxSteps;
      }
    } else {
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
    R* Lt = slot(3);   // cotangent of the stage slope before symmetrisation
    R* Lam = slot(4);  // ... and after
    R* F = slot(5);
    R* G = slot(6);    // F Ps (forward) / Lam F (reverse)
    for (long cs = ((Ssteps - 1) / cap) * cap; cs >= 0; cs -= cap) {
      const long ce = (cs + cap < Ssteps) ? cs + cap : Ssteps;
      // replay the interval from the filtered moments at k-1 up to the last start of this chunk, keeping the chunk's starts
      rows2d(d, d, [&](int i, int j) { return a.fP[n * a.P_sn + (k - 1) * a.P_sk + (long)(i * d + j) * a.P_si]; },
             [&](int i, int j, R v) { P0s[i * ld + j] = v; });
      if (tid < d) x0[tid] = a.fm[n * a.m_sn + (k - 1) * a.m_sk + tid * a.m_si];
      __syncthreads();
      {
        R tprev = t0, tnext = rmin(t0 + a.dt0, t1);
        R tlog = t0;  // (adaptive: the start times follow from the logged sizes)
        for (long s = 0; s < ce; ++s) {
          const R dt = dtl ? dtl[1 + s] : tnext - tprev;
          const R tstart = dtl ? tlog : tprev;
          if (s >= cs) {
            R* sv = starts + (s - cs) * sz;
            rows2d(d, d, [&](int i, int j) { return P0s[i * ld + j]; }, [&](int i, int j, R v) { sv[i * d + j] = v; });
            if (tid < d) sv[(long)d * d + tid] = x0[tid];
            if (tid == 0) {
              dts[s - cs] = dt;
              tss[s - cs] = tstart;
            }
          }
          if (s + 1 < ce) {
            stages_fwd(P0s, Ps, F, G, dt, tstart);
            step_end(P0s, dt);
          }
          tlog += dt;
          tprev = rmin(tnext, t1);
          const R tn = tnext + a.dt0;
          tnext = (tn > t1 - Tol<R>::v) ? t1 : tn;
        }
      }
      __syncthreads();
      for (long s = ce - 1; s >= cs; --s) {
        const R* sv = starts + (s - cs) * sz;
        rows2d(d, d, [&](int i, int j) { return sv[i * d + j]; },  // (each thread reads back the entries it wrote)
               [&](int i, int j, R v) { P0s[i * ld + j] = v; });
        if (tid < d) x0[tid] = sv[(long)d * d + tid];
        __syncthreads();
        const R dt = dts[s - cs];
        const R tstart = tss[s - cs];
        AWG_TICK(7)  // replay, step start
        stages_fwd(P0s, Ps, F, G, dt, tstart);
        AWG_TICK(8)  // the step's stages forward
        // (lam in two copies by the stage's parity: a stage's last phase still reads it while the next stage's first phase writes)
        auto take_cotangent = [&](int sv) __attribute__((always_inline)) {  // Ybar_P of stage sv = G + G^T from the product left in LDS
          slots([&](int i, int j, int) { return G[i * ld + j] + G[j * ld + i]; },
                [&](int, int, int u, R v) {
#pragma unroll
                  for (int s6 = 0; s6 < 6; ++s6) yP[s6][u] = (s6 == sv) ? v : yP[s6][u];
                });
        };
        for (int si = nst - 1; si >= 0; --si) {
          R* lamv = (si & 1) ? lam2 : lam;
          if (si + 1 < nst) take_cotangent(si + 1);
          // cotangent of slope si: dt (b_si ybar + sum_{r > si} a_r,si Ybar_r); the stage value again
          {
            R cf[6];
#pragma unroll
            for (int s6 = 0; s6 < 6; ++s6) cf[s6] = (s6 > si && s6 < nst) ? rka(s6, si) : R(0);
            const R bsi = a.rk.b[si];
            slots([&](int i, int j, int u) {
                    R s2 = bsi * Pb[i * ld + j];
#pragma unroll
                    for (int s6 = 1; s6 < 6; ++s6) s2 = rfma(cf[s6], yP[s6][u], s2);
                    return dt * s2;
                  },
                  [&](int i, int j, int, R v) { Lt[i * ld + j] = v; });
          }
          if (tid < d) {
            R s2 = a.rk.b[si] * mb[tid];
            for (int r = nst - 1; r > si; --r) s2 = rfma(a.rk.a[r][si], ym[64 * r + tid], s2);
            lamv[tid] = dt * s2;
          }
          stage_value(si, P0s, Ps, dt);  // (synchronises)
          rows2d(d, d, [&](int i, int j) { return R(0.5) * (Lt[i * ld + j] + Lt[j * ld + i]); }, [&](int i, int j, R v) { Lam[i * ld + j] = v; });
          cur_t = rfma(rk_stage_c(a.rk, si), dt, tstart);
          drift_eval(xs, F);
          __syncthreads();
#if defined(CDKF_AWG_CUSTOM) && CDKF_AWG_CUSTOM_SECOND
          if (second) {  // g(xs) again, and u = 0.5 Ps^T lam: the cotangent of g in the mean's slope
            custom_g(xs, g1);
            if (tid < d) g2[tid] = R(0.5) * dot(d, [&](int i) { return Ps[i * ld + tid]; }, [&](int i) { return lamv[i]; });
            __syncthreads();
          }
#elif !defined(CDKF_AWG_CUSTOM)
          if (second) {  // (MLP) g(xs) again, and u = 0.5 Ps^T lam: the cotangent of g in the mean's slope
            mlp_g(g1);
            if (tid < d) g2[tid] = R(0.5) * dot(d, [&](int i) { return Ps[i * ld + tid]; }, [&](int i) { return lamv[i]; });
            __syncthreads();
          }
#endif
          AWG_TICK(9)  // stage cotangent, stage value, drift
          // Ybar_P = F^T Lam + Lam F = (Lam F) + (Lam F)^T;  G = 2 Lam Ps where the drift's parameters / state derivative want it
          if (l96)
            l96_LamF(xs, Lam, G);
          else
            gemm(d, d, d, [&](int i, int kk) { return Lam[i * ld + kk]; }, [&](int kk, int j) { return F[kk * ld + j]; },
                 [&](int i, int j, R v) { G[i * ld + j] = second ? rfma(R(0.25) * lamv[i], g1[j], v) : v; });  // (Ybar_P = G + G^T gains sym(0.5 lam g^T))
          if (lin) {
            gemm(d, d, d, [&](int i, int kk) { return Lam[i * ld + kk]; }, [&](int kk, int j) { return Ps[kk * ld + j]; },
                 [&](int i, int j, R v) { g[i * d + j] += rfma(lamv[i], xs[j], R(2) * v); });  // dW += lam x^T + G
#ifdef CDKF_AWG_CUSTOM
          } else if (custom) {
            // G2 = 2 Lam Ps in full; then c_z = sum_ij G2_ij d F_ij / dz (+ lam . df/dz for a parameter) for every state component and
            // parameter z, one nested-dual evaluation per (column j, z): the workgroup's threads as (group, z), a group takes every
            // NG-th column, the partial sums meet in a free slot
            R* G2 = slot(7);
            R* part = cpart;
            gemm(d, d, d, [&](int i, int kk) { return Lam[i * ld + kk]; }, [&](int kk, int j) { return Ps[kk * ld + j]; },
                 [&](int i, int j, R v) { G2[i * ld + j] = R(2) * v; });
            __syncthreads();
            const int Z = cZ, NG = cNG;
            if (tid < NG * Z) {
              const int grp = tid / Z, z = tid - grp * Z;
              R s2 = R(0);
              for (int j = grp; j < d; j += NG) {
                s2 += awg_custom_contract<R>(th, xs, G2, ld, j, z, (j == 0 && z >= d) ? lamv : (const R*)nullptr, cur_u, cur_t);
#if CDKF_AWG_CUSTOM_SECOND
                if (second) s2 += awg_custom_third<R>(th, xs, g2, j, z, cur_u, cur_t);  // (j in the role of the divergence's index i)
#endif
              }
              part[grp * Z + z] = s2;
            }
#else
          } else if (mlp) {  // G2 = 2 Lam Ps in full, then the network's reverse pass (weights' gradient into g, state's into g3)
            R* G2 = slot(7);
            gemm(d, d, d, [&](int i, int kk) { return Lam[i * ld + kk]; }, [&](int kk, int j) { return Ps[kk * ld + j]; },
                 [&](int i, int j, R v) { G2[i * ld + j] = R(2) * v; });
            mlp_bwd(xs, lamv, G2, second ? g2 : (const R*)nullptr, g3);
#endif
          } else if (tid < 3 * d && 3 * d <= NT) {  // Lorenz-96: the three entries of row i of G the state derivative of F touches, a thread each
            const int which = fdiv(tid, d), i = tid - which * d;
            const int ip1 = (i + 1 >= d) ? 0 : i + 1, im1 = (i == 0) ? d - 1 : i - 1, im2 = (im1 == 0) ? d - 1 : im1 - 1;
            const int col = which == 0 ? ip1 : (which == 1 ? im2 : im1);  // G[i][i+1], G[i][i-2], G[i][i-1]
            R* gw = which == 0 ? g1 : (which == 1 ? g2 : g3);
            gw[i] = R(2) * dot(d, [&](int kk) { return Lam[i * ld + kk]; }, [&](int kk) { return Ps[kk * ld + col]; });
          } else if (tid < d && 3 * d > NT) {
            const int i = tid;
            const int ip1 = (i + 1 >= d) ? 0 : i + 1, im1 = (i == 0) ? d - 1 : i - 1, im2 = (im1 == 0) ? d - 1 : im1 - 1;
            auto lrow = [&](int kk) { return Lam[i * ld + kk]; };
            g1[i] = R(2) * dot(d, lrow, [&](int kk) { return Ps[kk * ld + ip1]; });  // G[i][i+1]
            g2[i] = R(2) * dot(d, lrow, [&](int kk) { return Ps[kk * ld + im2]; });  // G[i][i-2]
            g3[i] = R(2) * dot(d, lrow, [&](int kk) { return Ps[kk * ld + im1]; });  // G[i][i-1]
          }
          if (gQ) slots([&](int i, int j, int u) { return gQacc[u] + Lam[i * ld + j]; }, [&](int, int, int u, R v) { gQacc[u] = v; });
          __syncthreads();
          if (tid < d) {  // Ybar_m = F^T lam (+ the Jacobian's own state derivative contracted with G)
            const int c = tid;
            R s2;
            if (l96) {  // column c of F: rows c-1, c+2, c+1, c
              const int cp1 = wrap(c + 1), cp2 = wrap(c + 2), cm1 = wrap(c - 1), cm2 = wrap(c - 2);
              s2 = lamv[cm1] * xs[cm2];
              s2 = rfma(-lamv[cp2], xs[cp1], s2);
              s2 = rfma(lamv[cp1], xs[cp2] - xs[cm1], s2);
              s2 -= lamv[c];
            } else {
              s2 = dot(d, [&](int r) { return F[r * ld + c]; }, [&](int r) { return lamv[r]; });
            }
            if (lin) {
              g[d * d + c] += lamv[c];
#ifdef CDKF_AWG_CUSTOM
            } else if (custom) {
              for (int grp = 0; grp < cNG; ++grp) s2 += cpart[grp * cZ + c];
#else
            } else if (mlp) {
              s2 = g3[c];  // (the network's reverse pass returns the whole state gradient, F^T lam included)
#endif
            } else {
              // xbar[i-1] += G[i][i+1] - G[i][i-2];  xbar[i+1] += G[i][i-1];  xbar[i-2] -= G[i][i-1]
              const int cp1 = (c + 1 >= d) ? 0 : c + 1, cm1 = (c == 0) ? d - 1 : c - 1, cp2 = (cp1 + 1 >= d) ? 0 : cp1 + 1;
              s2 += g1[cp1] - g2[cp1];
              s2 += g3[cm1];
              s2 -= g3[cp2];
            }
            ym[64 * si + c] = s2;
          }
#ifdef CDKF_AWG_CUSTOM
          if (custom) {
            const int Z = cZ;
            if (tid >= d && tid < Z) {
              // (straight into the result: as a register of these few lanes, kept across the steps like the Lorenz-96 forcing's, the
              //  sum came back holding only the LAST step's share in three of four builds of the run-time compiled kernel at -O2 / -O3
              //  -- scripts/gpu_fuzz_custom.py)
              R s3 = R(0);
              for (int grp = 0; grp < cNG; ++grp) s3 += cpart[grp * Z + tid];
              g[tid - d] += s3;
            }
          } else
#endif
          if (l96 && tid == 64) gForcing += dot(d, [&](int r) { return lamv[r]; }, [&](int) { return R(1); });  // (a thread of another wavefront)
          if (a.ukf && l96) {  // Ybar_P = G + G^T gains the cotangent of Ps through lam . curvature(Ps) (oracle: ukf_curvature_vjp)
            if (tid < d) {
              G[wrap(tid + 1) * ld + wrap(tid - 1)] += R(0.5) * lamv[tid];
              G[wrap(tid - 2) * ld + wrap(tid - 1)] -= R(0.5) * lamv[tid];
            }
            __syncthreads();
          }
          AWG_TICK(10)  // right-hand-side adjoint products
        }
        take_cotangent(0);
        __syncthreads();
        // cotangent of the step's start
        {
          R cf[6];
#pragma unroll
          for (int s6 = 0; s6 < 6; ++s6) cf[s6] = (s6 < nst) ? R(1) : R(0);
          slots([&](int i, int j, int u) {
                  R s2 = Pb[i * ld + j];
#pragma unroll
                  for (int s6 = 0; s6 < 6; ++s6) s2 = rfma(cf[s6], yP[s6][u], s2);
                  return s2;
                },
                [&](int i, int j, int, R v) { Lt[i * ld + j] = v; });
        }
        if (tid < d) {
          R s2 = mb[tid];
          for (int si = 0; si < nst; ++si) s2 += ym[64 * si + tid];
          mb[tid] = s2;
        }
        __syncthreads();
        add_sym(Pb, Lt, true);
        AWG_TICK(11)  // step end
      }
    }
  }
  // ---- results ------------------------------------------------------------------------------------------------------------------
  if (gm) {
    slots([&](int, int, int u) { return gQacc[u]; }, [&](int i, int j, int, R v) { gQ[i * d + j] = v; });
    if (tid < d) gm[tid] = mb[tid];
    rows2d(d, d, [&](int i, int j) { return R(0.5) * (Pb[i * ld + j] + Pb[j * ld + i]); }, [&](int i, int j, R v) { gP0[i * d + j] = v; });
  }
  if (l96 && tid == 64) g[0] = gForcing;
  if (st && tid == 0 && a.status) atomicOr(&a.status[n], st);
#ifdef CDKF_AWG_PROFILE
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    printf("awg cycles/obs-step (sizeof real %d, d %d):", (int)sizeof(R), d);
    for (int q2 = 0; q2 < 16; ++q2) printf(" [%d] %lld", q2, awg_prof[q2] / a.T);
    printf("\n");
    for (int q2 = 0; q2 < 16; ++q2) awg_prof[q2] = 0;
  }
#endif
#undef AWG_FOR
}

}  // namespace cdkf
