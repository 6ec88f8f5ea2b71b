// Micro-benchmark (never shipped): do matrix-core and vector instructions of TWO wavefronts on ONE SIMD overlap on gfx950?
// (the question behind "spread one trajectory over several wavefronts so that one trajectory's latency phases
// run under another's products".)  Eight wavefronts per workgroup = two per SIMD; wavefronts 0-3 run role A, 4-7 role B:
//   roles: M = back-to-back v_mfma_*_16x16x4 on four accumulators, V = back-to-back vector FMAs on eight accumulators,
//          m4 = v_mfma_f64_4x4x4 (four blocks) on four accumulators, I = idle.
// Printed: cycles (s_memtime of wavefront 0 / 4) per role pair; "sum-like" = shared pipe, "max-like" = separate pipes.
// Build: hipcc --offload-arch=gfx950 -O3 -o mb_overlap mb_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

enum { ROLE_I = 0, ROLE_M = 1, ROLE_V = 2, ROLE_M4 = 3, ROLE_INT = 4, ROLE_LDS = 5, ROLE_V32 = 6, ROLE_MD = 7, ROLE_MN = 8, ROLE_MV = 9 };

template <typename R>
__device__ __forceinline__ R run_role(int role, int iters, R seed) {
  using V4 = typename std::conditional<sizeof(R) == 8, f64x4, f32x4>::type;
  R out = 0;
  if (role == ROLE_M) {
    V4 c0{0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    R a = seed, b = seed * R(0.5);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if constexpr (sizeof(R) == 8) {
          c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
          c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        } else {
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
          c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
        }
      }
    }
    out = c0[0] + c1[1] + c2[2] + c3[3];
  } else if (role == ROLE_MD) {  // ONE accumulator: every product waits for the one before it (a data hazard, not a busy pipe)
    V4 c0{0, 0, 0, 0};
    R a = seed, b = seed * R(0.5);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if constexpr (sizeof(R) == 8) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        else c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
      }
    }
    out = c0[0] + c0[1] + c0[2] + c0[3];
  } else if (role == ROLE_MN) {  // four accumulators, and after every product the wavefront steps aside (s_nop) for most of the product's passes
    V4 c[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) c[q] = V4{0, 0, 0, 0};
    R a = seed, b = seed * R(0.5);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if constexpr (sizeof(R) == 8) {
          c[u & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[u & 3], 0, 0, 0);
          asm volatile("s_nop 15");
          asm volatile("s_nop 15");
          asm volatile("s_nop 15");
        } else {
          c[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[u & 3], 0, 0, 0);
          asm volatile("s_nop 15");
        }
      }
    }
    out = c[0][0] + c[1][1] + c[2][2] + c[3][3];
  } else if (role == ROLE_MV) {  // ONE wavefront: every product followed by independent vector FMAs of its own (what a GEMM kernel interleaves)
    V4 c[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) c[q] = V4{0, 0, 0, 0};
    R v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = seed + R(q);
    R a = seed, b = seed * R(0.5);
    const R va = seed * R(1e-3), vb = R(0.999);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if constexpr (sizeof(R) == 8) {
          c[u & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[u & 3], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = (R)__builtin_fma((double)v[q], (double)vb, (double)va);
        } else {
          c[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[u & 3], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = (R)__builtin_fmaf((float)v[q], (float)vb, (float)va);
        }
      }
    }
    out = c[0][0] + c[1][1] + c[2][2] + c[3][3];
#pragma unroll
    for (int q = 0; q < 8; ++q) out += v[q];
  } else if (role == ROLE_M4) {
    if constexpr (sizeof(R) == 8) {
      double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
      double a = seed, b = seed * 0.5;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
          c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
        }
      }
      out = c0 + c1 + c2 + c3;
    }
  } else if (role == ROLE_V) {
    R c[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) c[q] = seed + R(q);
    const R a = seed * R(1e-3), b = R(0.999);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int q = 0; q < 8; ++q) c[q] = (sizeof(R) == 8) ? (R)__builtin_fma((double)c[q], (double)b, (double)a) : (R)__builtin_fmaf((float)c[q], (float)b, (float)a);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) out += c[q];
  }
  else if (role == ROLE_V32) {  // 32-bit float FMAs whatever R is
    float c[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) c[q] = (float)seed + q;
    const float a = (float)seed * 1e-3f, b = 0.999f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int q = 0; q < 8; ++q) c[q] = __builtin_fmaf(c[q], b, a);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) out += (R)c[q];
  } else if (role == ROLE_INT) {  // 32-bit integer multiply-adds
    unsigned c[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) c[q] = (unsigned)seed + q;
    const unsigned b = 3u + (unsigned)seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int q = 0; q < 8; ++q) c[q] = (c[q] & 0xffffffu) * (b & 0xffffffu) + q;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) out += (R)(c[q] & 7);
  } else if (role == ROLE_LDS) {  // dependent 64-bit LDS reads (a pointer chase: latency), sixteen per iteration
    extern __shared__ unsigned long long lds[];
    unsigned idx = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) idx = (unsigned)lds[idx & 511];
    }
    out = (R)(idx & 7);
  }
  return out;
}

template <typename R>
__global__ __launch_bounds__(512, 1) void overlap_kernel(R* out, unsigned long long* cyc, int* simd, int roleA, int roleB, int iters, int prio) {
  const int wave = threadIdx.x >> 6;
  const int role = wave < 4 ? roleA : roleB;
  if (prio) {  // role A (the products) at the lowest priority, role B at the highest
    if (wave < 4) __builtin_amdgcn_s_setprio(0);
    else __builtin_amdgcn_s_setprio(3);
  }
  unsigned hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  {
    extern __shared__ unsigned long long lds[];
    lds[threadIdx.x] = (threadIdx.x * 37 + 11) & 511;
  }
  __syncthreads();
  const unsigned long long t0 = stamp();
  const R r = run_role<R>(role, iters, R(threadIdx.x & 7) + R(1));
  const unsigned long long t1 = stamp();
  out[blockIdx.x * 512 + threadIdx.x] = r;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
    cyc[wave] = t1 - t0;
    simd[wave] = (hwid >> 4) & 3;
  }
}

template <typename R>
int bench(const char* name) {
  R* out;
  unsigned long long* cyc;
  int* simd;
  CK(hipMalloc(&out, 256 * 512 * sizeof(R)));
  CK(hipMalloc(&cyc, 8 * sizeof(unsigned long long)));
  CK(hipMalloc(&simd, 8 * sizeof(int)));
  const int iters = 2000;  // 16 instructions per iteration and role
  const char* nm[] = {"idle", "MFMA16x16x4", "VALU-FMA", "MFMA4x4x4", "VALU-int24", "LDS-chase", "VALU-fma32", "MFMA-chain", "MFMA+s_nop", "MFMA+ownFMA"};
  const int pairs[][2] = {{ROLE_M, ROLE_I}, {ROLE_V, ROLE_I}, {ROLE_INT, ROLE_I}, {ROLE_LDS, ROLE_I}, {ROLE_V32, ROLE_I}, {ROLE_M, ROLE_M}, {ROLE_V, ROLE_V},
                          {ROLE_M, ROLE_V}, {ROLE_M, ROLE_INT}, {ROLE_M, ROLE_LDS}, {ROLE_M, ROLE_V32}, {ROLE_M4, ROLE_I}, {ROLE_M4, ROLE_V},
                          {ROLE_MD, ROLE_I}, {ROLE_MD, ROLE_V}, {ROLE_MD, ROLE_LDS}, {ROLE_MD, ROLE_MD},
                          {ROLE_MN, ROLE_I}, {ROLE_MN, ROLE_V}, {ROLE_MN, ROLE_LDS}, {ROLE_MN, ROLE_INT}, {ROLE_MV, ROLE_I}};
  for (int prio = 0; prio < 2; ++prio)
  for (auto& p : pairs) {
    if (sizeof(R) == 4 && (p[0] == ROLE_M4)) continue;
    if (prio && (p[1] == ROLE_I || p[0] == p[1])) continue;
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(overlap_kernel<R>, dim3(256), dim3(512), 4096, 0, out, cyc, simd, p[0], p[1], iters, prio);
      CK(hipDeviceSynchronize());
      if (rep == 0) CK(hipMemset(cyc, 0, 8 * sizeof(unsigned long long)));
    }
    unsigned long long h[8];
    int s[8];
    CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    CK(hipMemcpy(s, simd, sizeof(s), hipMemcpyDeviceToHost));
    printf("%s%s  A=%-12s B=%-12s  cycles/instr: A %.1f  B %.1f   (SIMD of waves 0..7: %d %d %d %d %d %d %d %d)\n", name, prio ? " setprio(A 0, B 3)" : "", nm[p[0]], nm[p[1]],
           (double)h[0] / (16.0 * iters), (double)h[4] / (16.0 * iters), s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7]);
  }
  return 0;
}

int main() {
  if (bench<double>("f64")) return 1;
  if (bench<float>("f32")) return 1;
  return 0;
}
