// Micro-benchmark (never shipped): v_mfma_f64_4x4x4_4b_f64 on gfx950 -- (1) which lane holds which element of A, B and D,
// (2) what a lone wavefront pays for it: back-to-back independent, chained through C, and in the MFMA -> FMA -> MFMA pattern
// of a Runge-Kutta stage of the sixteen-lane sweep.  Build: hipcc --offload-arch=gfx950 -O3 -o mb_mfma64 mb_mfma64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void layout_kernel(double* out) {
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double a = (lane == la) ? 1.0 : 0.0, b = (lane == lb) ? 1.0 : 0.0;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      out[(la * 64 + lb) * 64 + lane] = d;
    }
}

__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

// MODE 0: chain through C (d = mfma(a, b, d));  1: four independent accumulators;  2: mfma -> fma -> mfma(A = result);
// 3: fma chain only (same count as mode 2's fmas);  4: two chained mfma + 3 dependent fma + 3 independent fma (a stage)
template <int MODE>
__global__ __launch_bounds__(64) void time_kernel(double* out, unsigned long long* cyc, int iters) {
  const int lane = threadIdx.x;
  double a = 1.0 + 1e-3 * lane, b = 1.0 - 1e-3 * lane, d0 = 0.1 * lane, d1 = 0.2, d2 = 0.3, d3 = 0.4;
  double c0 = 1e-3, c1 = 0.999;
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c0), "+v"(c1));
  const unsigned long long t0 = stamp();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 8; ++u) d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d0, 0, 0, 0);
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d1, 0, 0, 0);
        d2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d2, 0, 0, 0);
        d3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d3, 0, 0, 0);
      }
    } else if constexpr (MODE == 2) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d1, 0, 0, 0);
        a = __builtin_fma(d0, c0, c1);
      }
    } else if constexpr (MODE == 3) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a = __builtin_fma(a, c0, c1);
    } else if constexpr (MODE == 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        // "stage": three dependent fmas build the F entry from the stage value, two chained mfmas, one fma forms the next stage value
        double f = __builtin_fma(c0, d0, c1);
        f = __builtin_fma(c0, f, d1);
        f = __builtin_fma(c1, f, d2);
        double k = __builtin_amdgcn_mfma_f64_4x4x4f64(f, d0, d3, 0, 0, 0);
        k = __builtin_amdgcn_mfma_f64_4x4x4f64(d0, f, k, 0, 0, 0);
        d0 = __builtin_fma(c0, k, d0);
      }
    } else if constexpr (MODE == 5) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        // the same stage with the two products independent and one add
        double f = __builtin_fma(c0, d0, c1);
        f = __builtin_fma(c0, f, d1);
        f = __builtin_fma(c1, f, d2);
        const double k1 = __builtin_amdgcn_mfma_f64_4x4x4f64(f, d0, d3, 0, 0, 0);
        const double k2 = __builtin_amdgcn_mfma_f64_4x4x4f64(d0, f, d3, 0, 0, 0);
        d0 = __builtin_fma(c0, k1 + k2, d0);
      }
    }
  }
  const unsigned long long t1 = stamp();
  out[blockIdx.x * 64 + lane] = d0 + d1 + d2 + d3 + a;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
int run_time(const char* what, int per_iter, int blocks) {
  double* out;
  unsigned long long* cyc;
  CK(hipMalloc(&out, blocks * 64 * sizeof(double)));
  CK(hipMalloc(&cyc, blocks * sizeof(unsigned long long)));
  const int iters = 2000;
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(time_kernel<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(blocks);
  CK(hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double s = 0;
  for (auto v : h) s += (double)v;
  printf("%-70s blocks %5d: %.2f cycles per unit (%d units per iteration)\n", what, blocks, s / blocks / iters / per_iter, per_iter);
  hipFree(out);
  hipFree(cyc);
  return 0;
}

int main() {
  double* out;
  CK(hipMalloc(&out, 64 * 64 * 64 * sizeof(double)));
  hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, out);
  CK(hipDeviceSynchronize());
  std::vector<double> h(64 * 64 * 64);
  CK(hipMemcpy(h.data(), out, h.size() * sizeof(double), hipMemcpyDeviceToHost));
  // For each D lane: the (la, lb) pairs that reach it.  Expect 4 pairs per D lane (k = 0..3).
  printf("D lane <- (A lane, B lane) pairs\n");
  for (int ld = 0; ld < 64; ++ld) {
    printf("D%2d:", ld);
    for (int la = 0; la < 64; ++la)
      for (int lb = 0; lb < 64; ++lb)
        if (h[(la * 64 + lb) * 64 + ld] != 0.0) printf(" (A%2d,B%2d)", la, lb);
    printf("\n");
  }
  for (int blocks : {1, 1024, 2048}) {
    if (run_time<0>("mfma chained through C", 8, blocks)) return 1;
    if (run_time<1>("mfma, four independent accumulators", 8, blocks)) return 1;
    if (run_time<2>("mfma -> fma -> mfma (A operand from the fma)", 8, blocks)) return 1;
    if (run_time<3>("fma chain alone", 8, blocks)) return 1;
    if (run_time<4>("stage: 3 dependent fma + 2 chained mfma + 1 fma", 4, blocks)) return 1;
    if (run_time<5>("stage: 3 dependent fma + 2 independent mfma + add + fma", 4, blocks)) return 1;
  }
  return 0;
}
