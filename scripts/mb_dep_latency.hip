// Microbenchmark (dev aid): issue cost of dependent vs independent fp64 FMA chains for ONE wavefront on a SIMD.
// hipcc --offload-arch=gfx950 -O3 -o dep_latency dep_latency.hip && ./dep_latency
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CHAINS>
__global__ void fma_chains(double* out, long* cyc, int iters, double a, double b) {
  double x[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x * 1e-3 + c;
  long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 48 / CHAINS; ++r)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) x[c] = __builtin_fma(x[c], a, b);
  }
  long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}

__global__ void rsq_chain(double* out, long* cyc, int iters) {
  double x = 1.5 + threadIdx.x * 1e-3;
  long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r) x = __builtin_amdgcn_rsq(x) + 1.0;
  }
  long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}

// 12 dwordx2 stores per iteration at a row stride (the [T,comp,N] output pattern) interleaved with `fill` FMAs each.
// mode 0: 64 distinct lanes; 1: lanes repeat 8 addresses (the shipped small-batch grouping); 2: only 8 lanes active.
template <int FILL>
__global__ void store_burst(double* out, long* cyc, int iters, double* buf, long row, int mode) {
  const int lane = threadIdx.x;
  if (mode == 2 && lane >= 8) return;
  const int col = mode == 1 ? (lane & 7) : lane;
  double x = lane * 1e-3, y = 1.0;
  double* p = buf + col;
  long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    double* q = p + (long)(i & 63) * 12 * row;
#pragma unroll
    for (int r = 0; r < 12; ++r) {
      q[r * row] = x;
#pragma unroll
      for (int f = 0; f < FILL; ++f) y = __builtin_fma(y, 1.0000001, 1e-9);
    }
    x += 1.0;
  }
  long t1 = __builtin_readcyclecounter();
  out[lane] = y;
  if (lane == 0) *cyc = t1 - t0;
}

// The same 12 doubles per iteration written component-contiguous ([T,N,w] layout): per lane 3 + 9 adjacent doubles at a lane
// stride of 24 / 72 bytes -- 2 + 5 wide stores with immediate offsets and two address computations.
typedef double double2a __attribute__((ext_vector_type(2), aligned(8)));
template <int FILL>
__global__ void store_tn(double* out, long* cyc, int iters, double* buf, long row, int mode) {
  const int lane = threadIdx.x;
  const int col = mode == 1 ? (lane & 7) : lane;
  double x = lane * 1e-3, y = 1.0;
  long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    double* m = buf + (long)(i & 63) * 12 * row + col * 3;
    double* P = buf + (long)(i & 63) * 12 * row + 3 * row + col * 9;
    *(double2a*)(m) = double2a{x, x};
    m[2] = x;
#pragma unroll
    for (int f = 0; f < 2 * FILL; ++f) y = __builtin_fma(y, 1.0000001, 1e-9);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      *(double2a*)(P + 2 * r) = double2a{x, x};
#pragma unroll
      for (int f = 0; f < FILL; ++f) y = __builtin_fma(y, 1.0000001, 1e-9);
    }
    P[8] = x;
    x += 1.0;
  }
  long t1 = __builtin_readcyclecounter();
  out[lane] = y;
  if (lane == 0) *cyc = t1 - t0;
}

__global__ void add64_chain(double* out, long* cyc, int iters, long stride) {
  long a = threadIdx.x;
  long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      a += stride;
      asm volatile("" : "+v"(a));
    }
  }
  long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = (double)a;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}

template <typename K, typename... A>
static void run(const char* name, int per_iter, K k, A... args) {
  double* out;
  long* cyc;
  hipMalloc(&out, 64 * 8);
  hipMalloc(&cyc, 8);
  const int iters = 20000;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, iters, args...);
    hipDeviceSynchronize();
  }
  long h;
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-28s %.2f counter ticks per instruction\n", name, (double)h / iters / per_iter);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  // s_memtime / readcyclecounter ticks at a fixed 100 MHz on gfx9: calibrate against the 16-chain case (4 cycles / FMA expected)
  run("fma f64, 1 chain", 48, fma_chains<1>, 1.0000001, 1e-9);
  run("fma f64, 2 chains", 48, fma_chains<2>, 1.0000001, 1e-9);
  run("fma f64, 4 chains", 48, fma_chains<4>, 1.0000001, 1e-9);
  run("fma f64, 16 chains", 48, fma_chains<16>, 1.0000001, 1e-9);
  run("rsq f64 + add, 1 chain", 16, rsq_chain);
  run("64-bit integer add chain", 16, add64_chain, 32768L);
  double* buf;
  const long row = 4096;
  hipMalloc(&buf, 64 * 12 * row * 8 + 4096);
  const char* names[3] = {"64 distinct lanes", "8 addresses x 8 lanes", "8 lanes active"};
  for (int mode = 0; mode < 3; ++mode) {
    char nm[96];
    snprintf(nm, sizeof nm, "store burst, %s", names[mode]);
    run(nm, 12, store_burst<0>, buf, row, mode);
    snprintf(nm, sizeof nm, "store + 4 fma, %s", names[mode]);
    run(nm, 12, store_burst<4>, buf, row, mode);
    snprintf(nm, sizeof nm, "store + 16 fma, %s", names[mode]);
    run(nm, 12, store_burst<16>, buf, row, mode);
  }
  for (int mode = 0; mode < 2; ++mode) {
    char nm[96];
    snprintf(nm, sizeof nm, "[T,N,w] 7 wide stores (ticks per 12 doubles / 12), %s", names[mode]);
    run(nm, 12, store_tn<0>, buf, row, mode);
    snprintf(nm, sizeof nm, "[T,N,w] 7 wide stores + 6x4 fma (per 12 doubles / 12), %s", names[mode]);
    run(nm, 12, store_tn<4>, buf, row, mode);
  }
  return 0;
}
