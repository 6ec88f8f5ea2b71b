// Diagnostic build (never shipped): where does one observation step of the reg EKF kernel spend its cycles?
// s_memtime stamps around update / store / predict / store, summed per wave, plus ablations.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "../../cd_dynamax_amd/csrc/cdkf_reg_kernels.h"
using namespace cdkf;

__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

template <typename R, int D, int M, typename Drift, bool HSEL, bool STORES, bool STAMPS, int LPW = 64>
__global__ __launch_bounds__(64, 1) void ekf_diag(const RegArgs<R, D, M, Drift> a, unsigned long long* __restrict__ cyc) {
  constexpr int NS = Dims<D>::NS; constexpr int NP = Dims<D>::NP;
  if (LPW < 64 && threadIdx.x >= LPW) return;
  const long gid = (long)blockIdx.x * LPW + threadIdx.x;
  const bool live = gid < a.N;
  const long n = live ? gid : a.N - 1;
  const R* __restrict__ tp = a.t + n * a.t_sn;
  const R* __restrict__ yp = a.y + n * a.y_sn;
  long moff = n * a.m_sn, poff = n * a.P_sn;
  R ys[NS];
  for (int i = 0; i < D; ++i) ys[i] = a.m0[i];
  for (int e = 0; e < NP; ++e) ys[D + e] = a.P0[e];
  LlAcc ll; int st = 0; Dp5V<R> C; C.init();
  EkfRhs<R, D, Drift> rhs{a.drift, a.LQL, a.order};
  R tcur = tp[0];
  if (a.T > 1) tp += a.t_sk;
  R tnext_obs = tp[0];
  R ycur[M];
  for (int r = 0; r < M; ++r) ycur[r] = yp[r * a.y_si];
  unsigned long long c_upd = 0, c_st1 = 0, c_rk = 0, c_st2 = 0, nrk = 0;
  for (long k = 0; k < a.T; ++k) {
    unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
    if (STAMPS) s0 = stamp();
    ekf_update<R, D, M, HSEL>(a, ys, ycur, ll, st);
    if (STAMPS) s1 = stamp();
    if (STORES && live) store_moments<R, D>(a.fm, a.fP, moff, poff, a.m_si, a.P_si, ys);
    const R t1 = (k + 1 < a.T) ? tnext_obs : tcur + a.dt_final;
    if (k + 1 < a.T) yp += a.y_sk;
    if (k + 2 < a.T) tp += a.t_sk;
    R ynext[M];
    for (int r = 0; r < M; ++r) ynext[r] = yp[r * a.y_si];
    const R tnn = tp[0];
    if (STAMPS) s2 = stamp();
    {
      R tprev = tcur; R tnext = rmin(tcur + a.dt0, t1);
      while (tprev < t1) {
        dopri5_step<R, NS>(ys, tnext - tprev, rhs, C);
        tprev = rmin(tnext, t1);
        R tn = tnext + a.dt0;
        tnext = (tn > t1 - Tol<R>::v) ? t1 : tn;
        if (STAMPS) nrk += 1;
      }
    }
    if (STAMPS) s3 = stamp();
    if (STORES && live) store_moments<R, D>(a.pm, a.pP, moff, poff, a.m_si, a.P_si, ys);
    if (STAMPS) s4 = stamp();
    moff += a.m_sk; poff += a.P_sk;
    tcur = tnext_obs; tnext_obs = tnn;
    for (int r = 0; r < M; ++r) ycur[r] = ynext[r];
    c_upd += s1 - s0; c_st1 += s2 - s1; c_rk += s3 - s2; c_st2 += s4 - s3;
  }
  ll.flush();
  if (live) a.ll[n] = (R)ll.ll;
  if (STAMPS && threadIdx.x == 0) {
    cyc[blockIdx.x * 8 + 0] = c_upd; cyc[blockIdx.x * 8 + 1] = c_st1; cyc[blockIdx.x * 8 + 2] = c_rk;
    cyc[blockIdx.x * 8 + 3] = c_st2; cyc[blockIdx.x * 8 + 4] = nrk;
  }
}

template <typename K>
float timeit(K launch, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}

int main() {
  using R = double;
  const long N = 4096, T = 1000;
  std::vector<R> t(N * T), y(N * T * 3);
  unsigned long long seed = 12345;
  auto rnd = [&]() { seed = seed * 6364136223846793005ULL + 1442695040888963407ULL; return (double)((seed >> 11) & ((1ULL << 53) - 1)) / (double)(1ULL << 53); };
  for (long n = 0; n < N; ++n) {
    std::vector<double> u(T); double s = 0;
    for (long k = 0; k < T; ++k) { u[k] = rnd(); s += u[k]; }
    double acc = 0;
    for (long k = 0; k < T; ++k) { acc += u[k]; t[k * N + n] = acc / s * (0.005 * T); }   // TCN layout: t[T,N]
  }
  for (long k = 0; k < T; ++k) for (int i = 0; i < 3; ++i) for (long n = 0; n < N; ++n) y[(k * 3 + i) * N + n] = (rnd() - 0.5) * 20.0;
  R *dt_, *dy, *dll, *fm, *fP, *pm, *pP; unsigned long long* cyc;
  hipMalloc(&dt_, t.size() * sizeof(R)); hipMalloc(&dy, y.size() * sizeof(R)); hipMalloc(&dll, N * sizeof(R));
  hipMalloc(&fm, N * T * 3 * sizeof(R)); hipMalloc(&pm, N * T * 3 * sizeof(R)); hipMalloc(&fP, N * T * 9 * sizeof(R)); hipMalloc(&pP, N * T * 9 * sizeof(R));
  hipMalloc(&cyc, 64 * 8 * sizeof(unsigned long long));
  hipMemcpy(dt_, t.data(), t.size() * sizeof(R), hipMemcpyHostToDevice); hipMemcpy(dy, y.data(), y.size() * sizeof(R), hipMemcpyHostToDevice);
  RegArgs<R, 3, 3, DriftLorenz63<R, 3>> a{};
  a.drift.sigma = 10; a.drift.rho = 28; a.drift.beta = R(8.0 / 3.0);
  R eye[6] = {1, 0, 0, 1, 0, 1};
  for (int e = 0; e < 6; ++e) { a.LQL[e] = eye[e]; a.LQLz[e] = eye[e]; a.P0[e] = 5 * eye[e]; }
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) { a.H[i][j] = i == j; a.Rm[i][j] = i == j; } a.hb[i] = 0; a.m0[i] = 0; }
  a.dt0 = R(0.01); a.dt_final = R(1e-10); a.max_steps = 100000; a.order = 2; a.num_iter = 1; a.N = N; a.T = T;
  a.t_sn = 1; a.t_sk = N; a.y_sn = a.m_sn = a.P_sn = 1; a.y_sk = N * 3; a.m_sk = N * 3; a.P_sk = N * 9; a.y_si = a.m_si = a.P_si = N;
  a.t = dt_; a.y = dy; a.ll = dll; a.status = nullptr; a.fm = fm; a.fP = fP; a.pm = pm; a.pP = pP;
  using Dr = DriftLorenz63<R, 3>;
  dim3 g(N / 64), b(64);
  printf("HSEL stores      : %.3f ms\n", timeit([&] { hipLaunchKernelGGL((ekf_diag<R, 3, 3, Dr, true, true, false>), g, b, 0, 0, a, cyc); }, 5));
  printf("HSEL no stores   : %.3f ms\n", timeit([&] { hipLaunchKernelGGL((ekf_diag<R, 3, 3, Dr, true, false, false>), g, b, 0, 0, a, cyc); }, 5));
  printf("general stores   : %.3f ms\n", timeit([&] { hipLaunchKernelGGL((ekf_diag<R, 3, 3, Dr, false, true, false>), g, b, 0, 0, a, cyc); }, 5));
  printf("HSEL stamps      : %.3f ms\n", timeit([&] { hipLaunchKernelGGL((ekf_diag<R, 3, 3, Dr, true, true, true>), g, b, 0, 0, a, cyc); }, 2));
  printf("HSEL stores lpw32: %.3f ms\n", timeit([&] { hipLaunchKernelGGL((ekf_diag<R, 3, 3, Dr, true, true, false, 32>), dim3(N / 32), b, 0, 0, a, cyc); }, 5));
  printf("HSEL stores lpw16: %.3f ms\n", timeit([&] { hipLaunchKernelGGL((ekf_diag<R, 3, 3, Dr, true, true, false, 16>), dim3(N / 16), b, 0, 0, a, cyc); }, 5));
  printf("HSEL stores lpw8 : %.3f ms\n", timeit([&] { hipLaunchKernelGGL((ekf_diag<R, 3, 3, Dr, true, true, false, 8>), dim3(N / 8), b, 0, 0, a, cyc); }, 5));
  printf("HSEL stores lpw4 : %.3f ms\n", timeit([&] { hipLaunchKernelGGL((ekf_diag<R, 3, 3, Dr, true, true, false, 4>), dim3(N / 4), b, 0, 0, a, cyc); }, 5));
  hipLaunchKernelGGL((ekf_diag<R, 3, 3, Dr, true, true, true, 16>), dim3(64), b, 0, 0, a, cyc); hipDeviceSynchronize();
  {
    std::vector<unsigned long long> h(64 * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s[5] = {0, 0, 0, 0, 0};
    for (int w = 0; w < 64; ++w) for (int j = 0; j < 5; ++j) s[j] += h[w * 8 + j] / 64.0;
    printf("lpw16 per step: update %.0f | store_f+prefetch %.0f | rk %.0f | store_p %.0f\n", s[0] / T, s[1] / T, s[2] / T, s[3] / T);
  }
  hipLaunchKernelGGL((ekf_diag<R, 3, 3, Dr, true, true, true>), g, b, 0, 0, a, cyc); hipDeviceSynchronize();
  std::vector<unsigned long long> h(64 * 8);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double s[5] = {0, 0, 0, 0, 0};
  for (int w = 0; w < 64; ++w) for (int j = 0; j < 5; ++j) s[j] += h[w * 8 + j] / 64.0;
  printf("per step (mean over waves), s_memtime ticks: update %.0f | store_f+prefetch %.0f | rk %.0f (%.3f rk passes/step) | store_p %.0f\n",
         s[0] / T, s[1] / T, s[2] / T, s[4] / T, s[3] / T);
  return 0;
}
