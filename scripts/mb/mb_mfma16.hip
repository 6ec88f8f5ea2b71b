// Micro-benchmark (never shipped): what ONE wavefront per SIMD pays for the 64 x 64 x 9 products of the MLP right-hand side
// (cdkf_wave8_kernels.h) in its different forms -- v_mfma_{f64,f32}_16x16x4 back to back, with the B operand built from LDS reads
// as the kernel does, and the same product on the vector pipe (lane = row of W2, broadcast reads of the other factor).
// Build: hipcc --offload-arch=gfx950 -O3 -o mb_mfma16 mb_mfma16.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f64x4 mfma(double a, double b, f64x4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
template <typename R> struct V4;
template <> struct V4<double> { using T = f64x4; };
template <> struct V4<float> { using T = f32x4; };

__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
__device__ __forceinline__ unsigned long long realtime() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
template <typename R>
__device__ __forceinline__ R pin(R x) {
  asm volatile("" : "+v"(x));
  return x;
}

// MODE 0: 64 MFMA, four accumulators, operands in registers (the pipe alone)
//      1: the kernel's loop: B operand of k-step ks from two LDS reads and a multiply-add, software-pipelined
//      2: as 1 with the A operands forced through "a" (accumulator-file) constraints
//      3: vector pipe: lane = row p of W2 (64 registers), nine accumulators, B[q][0..8] by broadcast LDS reads
//      4: 16 chained MFMA on one accumulator + 64 on four (the fused layer-3 / transposed product), B from LDS
template <typename R, int MODE>
__global__ __launch_bounds__(256, 1) void time_kernel(R* out, unsigned long long* cyc, int iters, const R* src) {
  __shared__ R lds[4][64 * 12];
  __shared__ R w2s[64 * 80];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lm = lane & 15, lg = lane >> 4;
  R* W = lds[wave];
  for (int e = lane; e < 64 * 12; e += 64) W[e] = src[e & 1023];
  for (int e = threadIdx.x; e < 64 * 80; e += 256) w2s[e] = src[(e * 3) & 1023];
  R w2A[4][16], w1B[16], w2row[64];
  if constexpr (MODE == 3) {
#pragma unroll
    for (int q = 0; q < 64; ++q) w2row[q] = pin(src[(lane * 64 + q) & 1023]);
  } else {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) w2A[mt][ks] = pin(src[(mt * 16 + ks + lane) & 1023]);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) w1B[ks] = pin(src[(ks * 7 + lane) & 1023]);
  }
  const R e8 = (lm == 8) ? R(1) : R(0);
  __syncthreads();
  using T4 = typename V4<R>::T;
  T4 acc[4];
  R vacc[9];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) acc[mt] = T4{0, 0, 0, 0};
#pragma unroll
  for (int c = 0; c < 9; ++c) vacc[c] = 0;
  T4 acc3{0, 0, 0, 0};
  const unsigned long long r0 = realtime();
  const unsigned long long t0 = stamp();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) {
      R bv = w1B[0];
#pragma unroll
      for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = mfma(w2A[mt][ks], bv, acc[mt]);
    } else if constexpr (MODE == 1 || MODE == 2) {
      R dq_n = W[lg], aq_n = W[64 + lg];
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        R bv = __builtin_fma(dq_n, w1B[ks], aq_n * e8);
        if (ks < 15) {
          dq_n = W[4 * (ks + 1) + lg];
          aq_n = W[64 + 4 * (ks + 1) + lg];
        }
        asm volatile("" : "+v"(bv) : : "memory");
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          if constexpr (MODE == 2 && sizeof(R) == 8) {
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[mt]) : "a"(w2A[mt][ks]), "v"(bv));
          } else if constexpr (MODE == 2) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[mt]) : "a"(w2A[mt][ks]), "v"(bv));
          } else {
            acc[mt] = mfma(w2A[mt][ks], bv, acc[mt]);
          }
        }
      }
    } else if constexpr (MODE == 3) {
      // B image [64][10] (stride 10: 16-byte aligned pairs), every lane reads the same address
#pragma unroll
      for (int q = 0; q < 64; ++q) {
        const R w = w2row[q];
#pragma unroll
        for (int c = 0; c < 9; ++c) vacc[c] = __builtin_fma(w, W[128 + q * 10 + c], vacc[c]);
      }
    } else if constexpr (MODE == 4) {
      R sc_n = W[lg], x2_n = W[64 + lg], w_n[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) w_n[nt] = w2s[lg * 80 + 16 * nt + lm];
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const R sc = sc_n, x2 = x2_n, w0 = w_n[0], w1 = w_n[1], w2 = w_n[2], w3 = w_n[3];
        const R w3a = w1B[ks];
        R av = x2 * (w3a + e8);
        R b3v = sc * __builtin_fma(acc[ks >> 2][ks & 3], e8, e8);
        if (ks < 15) {
          const int pn = 4 * (ks + 1) + lg;
          sc_n = W[pn];
          x2_n = W[64 + pn];
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) w_n[nt] = w2s[pn * 80 + 16 * nt + lm];
        }
        asm volatile("" : "+v"(av), "+v"(b3v) : : "memory");
        acc3 = mfma(w3a, b3v, acc3);
        acc[0] = mfma(av, w0, acc[0]);
        acc[1] = mfma(av, w1, acc[1]);
        acc[2] = mfma(av, w2, acc[2]);
        acc[3] = mfma(av, w3, acc[3]);
      }
    }
  }
  const unsigned long long t1 = stamp();
  const unsigned long long r1 = realtime();
  R s = 0;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) s += acc[mt][0] + acc[mt][1] + acc[mt][2] + acc[mt][3];
#pragma unroll
  for (int c = 0; c < 9; ++c) s += vacc[c];
  s += acc3[0] + acc3[1] + acc3[2] + acc3[3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    cyc[2 * blockIdx.x] = t1 - t0;
    cyc[2 * blockIdx.x + 1] = r1 - r0;
  }
}

template <typename R, int MODE>
int run_time(const char* what, int blocks) {
  R *out, *src;
  unsigned long long* cyc;
  CK(hipMalloc(&out, blocks * 256 * sizeof(R)));
  CK(hipMalloc(&src, 1024 * sizeof(R)));
  CK(hipMalloc(&cyc, 2 * blocks * sizeof(unsigned long long)));
  std::vector<R> hs(1024);
  for (int i = 0; i < 1024; ++i) hs[i] = (R)(1e-3 * ((i * 37) % 101) - 0.05);
  CK(hipMemcpy(src, hs.data(), 1024 * sizeof(R), hipMemcpyHostToDevice));
  const int iters = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((time_kernel<R, MODE>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, src);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((time_kernel<R, MODE>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, src);
  (void)hipEventRecord(e1, 0);
  CK(hipDeviceSynchronize());
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2 * blocks);
  CK(hipMemcpy(h.data(), cyc, 2 * blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double s = 0, r = 0;
  for (int b = 0; b < blocks; ++b) s += (double)h[2 * b], r += (double)h[2 * b + 1];
  printf("%-88s fp%d blocks %4d: %8.1f s_memtime ticks per product, %7.3f us (s_memrealtime, 100 MHz), kernel %.3f ms\n", what, (int)sizeof(R) * 8, blocks,
         s / blocks / iters, r / blocks / iters / 100.0, ms);
  (void)hipFree(out);
  (void)hipFree(src);
  (void)hipFree(cyc);
  return 0;
}

template <typename R>
int all(int blocks) {
  if (run_time<R, 0>("64 MFMA 16x16x4, four accumulators, operands in registers", blocks)) return 1;
  if (run_time<R, 1>("64 MFMA, B from two LDS reads + fma per k-step (the kernel's tangent product)", blocks)) return 1;
  if (run_time<R, 2>("  the same, A operands through the accumulator file (inline asm)", blocks)) return 1;
  if (run_time<R, 3>("vector pipe: lane = row of W2, 576 fma, B by broadcast LDS reads", blocks)) return 1;
  if (run_time<R, 4>("16 chained + 64 MFMA, B operands from LDS (fused layer 3 / transposed product)", blocks)) return 1;
  return 0;
}

int main() {
  for (int blocks : {1, 256}) {
    if (all<double>(blocks)) return 1;
    if (all<float>(blocks)) return 1;
  }
  return 0;
}
