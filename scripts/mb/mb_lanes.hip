// micro-benchmark: does a partially filled wave (EXEC low lanes only) issue fp64 faster? and what does a lone
// wave per SIMD sustain?  Dev aid only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../cd_dynamax_amd/csrc/cdkf_reg_kernels.h"
using namespace cdkf;

template <typename R, int D, int M, typename Drift, int LPW>
__global__ __launch_bounds__(64) void ekf_lpw(const RegArgs<R, D, M, Drift> a) {
  constexpr int NS = Dims<D>::NS; constexpr int NP = Dims<D>::NP;
  if (threadIdx.x >= LPW) return;
  const long gid = (long)blockIdx.x * LPW + threadIdx.x;
  if (gid >= a.N) return;
  const long n = gid;
  const R* __restrict__ tp = a.t + n * a.t_stride;
  const R* __restrict__ yp = a.y + n * a.T * M;
  R ys[NS];
  for (int i = 0; i < D; ++i) ys[i] = a.m0[i];
  for (int e = 0; e < NP; ++e) ys[D + e] = a.P0[e];
  double ll = 0.0; int st = 0;
  EkfRhs<R, D, Drift> rhs{a.drift, a.LQL, a.order};
  R tcur = tp[0];
  R ycur[M];
  for (int r = 0; r < M; ++r) ycur[r] = yp[r];
  for (long k = 0; k < a.T; ++k) {
    const long kn = (k + 1 < a.T) ? k + 1 : k;
    R tnext_obs = tp[kn]; R ynext[M];
    for (int r = 0; r < M; ++r) ynext[r] = yp[kn * M + r];
    ekf_update<R, D, M>(a, ys, ycur, ll, st);
    const long row = n * a.T + k;
    store_moments<R, D>(a.fm, a.fP, row, ys);
    const R t1 = (k + 1 < a.T) ? tnext_obs : tcur + a.dt_final;
    integrate<R, NS>(ys, tcur, t1, a.dt0, a.max_steps, rhs);
    store_moments<R, D>(a.pm, a.pP, row, ys);
    tcur = tnext_obs;
    for (int r = 0; r < M; ++r) ycur[r] = ynext[r];
  }
  a.ll[n] = (R)ll;
}

template <typename R, int LPW>
float run(RegArgs<R, 3, 3, DriftLorenz63<R, 3>> a, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  unsigned blocks = (a.N + LPW - 1) / LPW;
  hipLaunchKernelGGL((ekf_lpw<R, 3, 3, DriftLorenz63<R, 3>, LPW>), dim3(blocks), dim3(64), 0, 0, a);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((ekf_lpw<R, 3, 3, DriftLorenz63<R, 3>, LPW>), dim3(blocks), dim3(64), 0, 0, a);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}

template <typename R>
void bench(long N, long T) {
  std::vector<R> t(N * T), y(N * T * 3);
  for (long n = 0; n < N; ++n) { double acc = 0; for (long k = 0; k < T; ++k) { acc += 0.0025 + 0.005 * ((n * 7 + k * 13) % 97) / 97.0; t[n * T + k] = R(acc); } }
  for (size_t i = 0; i < y.size(); ++i) y[i] = R(((i * 2654435761u) % 2000) / 100.0 - 10.0);
  R *dt_, *dy, *dll, *fm, *fP, *pm, *pP;
  hipMalloc(&dt_, t.size() * sizeof(R)); hipMalloc(&dy, y.size() * sizeof(R)); hipMalloc(&dll, N * sizeof(R));
  hipMalloc(&fm, N * T * 3 * sizeof(R)); hipMalloc(&pm, N * T * 3 * sizeof(R)); hipMalloc(&fP, N * T * 9 * sizeof(R)); hipMalloc(&pP, N * T * 9 * sizeof(R));
  hipMemcpy(dt_, t.data(), t.size() * sizeof(R), hipMemcpyHostToDevice); hipMemcpy(dy, y.data(), y.size() * sizeof(R), hipMemcpyHostToDevice);
  RegArgs<R, 3, 3, DriftLorenz63<R, 3>> a{};
  a.drift.sigma = 10; a.drift.rho = 28; a.drift.beta = R(8.0 / 3.0);
  R eye[6] = {1, 0, 0, 1, 0, 1};
  for (int e = 0; e < 6; ++e) { a.LQL[e] = eye[e]; a.LQLz[e] = eye[e]; a.P0[e] = 5 * eye[e]; }
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) { a.H[i][j] = i == j; a.Rm[i][j] = i == j; } a.hb[i] = 0; a.m0[i] = 0; }
  a.dt0 = R(0.01); a.dt_final = R(1e-10); a.max_steps = 100000; a.order = 2; a.num_iter = 1; a.N = N; a.T = T; a.t_stride = T;
  a.t = dt_; a.y = dy; a.ll = dll; a.status = nullptr;
  for (int full = 0; full < 2; ++full) {
    a.fm = full ? fm : nullptr; a.fP = full ? fP : nullptr; a.pm = full ? pm : nullptr; a.pP = full ? pP : nullptr;
    printf("%s N=%ld full=%d: lpw64 %.3f ms, lpw32 %.3f ms, lpw16 %.3f ms, lpw8 %.3f ms\n", sizeof(R) == 8 ? "f64" : "f32", N, full,
           run<R, 64>(a, 3), run<R, 32>(a, 3), run<R, 16>(a, 3), run<R, 8>(a, 3));
  }
  hipFree(dt_); hipFree(dy); hipFree(dll); hipFree(fm); hipFree(fP); hipFree(pm); hipFree(pP);
}

int main() {
  bench<double>(4096, 1000); bench<float>(4096, 1000);
  bench<double>(65536, 1000); bench<double>(131072, 1000); bench<double>(262144, 1000);
  return 0;
}
