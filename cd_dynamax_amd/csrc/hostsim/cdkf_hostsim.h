// cdkf_hostsim.h -- the HIP device vocabulary of the kernel headers restated for a HOST build (CDKF_HOST_SIM), so that the same
// templates the GPU runs -- filter_reg_body / ekf_grad_reg_body, the workgroup kernels of cdkf_wg2_kernels.h, the dual numbers of
// cdkf_dual.h, the run-time generated drift sources -- compile with clang++ for x86-64 under AddressSanitizer / UndefinedBehavior-
// Sanitizer / MemorySanitizer / ThreadSanitizer (the pool has no GPU sanitizers).  Test infrastructure: nothing under cd_dynamax_amd/
// includes this unless CDKF_HOST_SIM is defined, and the product path never is.
//
// Execution model: one OS thread per GPU thread of ONE workgroup at a time (hostsim::launch runs the grid's blocks in sequence, so a
// `__shared__` variable can be a function-local static).  __syncthreads() is a pthread barrier over the block (ThreadSanitizer sees
// it, so two threads touching an LDS word without a barrier between them are a reported race -- also inside a wavefront, where the
// GPU would have run them in lockstep: wave-synchronous code must say so with wave_barrier, as the device code does).  Cross-lane
// operations (readlane, shuffles, the 16x16x4 MFMA) exchange through a per-wavefront slot array between two wavefront barriers; every
// lane of the wavefront must reach them, as on the device.  Lane maps of the matrix instruction: cdkf_wg2_kernels.h, wg_mm.
#pragma once
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <type_traits>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

namespace hostsim {
struct Dim3 {
  unsigned x = 0, y = 0, z = 0;
};
struct Block {
  // TWO barrier objects used in turn: ThreadSanitizer models a barrier as release-before / acquire-after on the object's address, so
  // with one object a thread that is slow to LEAVE barrier k would acquire what a fast thread released ENTERING barrier k + 1 -- and
  // everything the fast thread did in between would count as ordered before the slow thread's next accesses (measured: a removed
  // __syncthreads() went unreported more often than not).  The next release on the same object now needs everybody through the other one.
  pthread_barrier_t all[2];
  pthread_barrier_t wave[16][2];
  uint64_t slot[16][64][4];  // exchange area of the cross-lane operations
  unsigned nthreads = 0;
};
inline Block*& cur_block() {
  static Block* b = nullptr;
  return b;
}
}  // namespace hostsim

inline thread_local hostsim::Dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace hostsim {
// ThreadSanitizer checks its shadow words without locking: two conflicting accesses that happen at (almost) the same moment -- two
// threads that have just left the same barrier -- can slip past each other.  HOSTSIM_JITTER=<k> therefore lets about one thread in k
// sleep for 0 .. 255 microseconds after every barrier (a hash of thread and barrier count decides), which pulls the two sides of a
// missing-barrier race apart in time; tests/test_hostsim.py checks that a deliberately removed barrier is reported this way.
inline int jitter_period() {
  static const int k = [] {
    const char* e = getenv("HOSTSIM_JITTER");
    return e ? atoi(e) : 0;
  }();
  return k;
}
inline void jitter() {
  const int k = jitter_period();
  if (k <= 0) return;
  static thread_local uint32_t count = 0;
  uint32_t h = (threadIdx.x + 1) * 2654435761u ^ (++count * 40503u);
  h ^= h >> 15;
  h *= 2246822519u;
  h ^= h >> 13;
  if ((int)(h % (uint32_t)k) == 0) {
    struct timespec ts = {0, (long)((h >> 8) & 255) * 1000};
    nanosleep(&ts, nullptr);
  }
}
inline thread_local unsigned block_phase = 0, wave_phase = 0;
// mutation check of the instrument itself: HOSTSIM_SKIP_BARRIER=<k> makes every thread walk through its k-th __syncthreads() without
// waiting -- ThreadSanitizer must then report the accesses that barrier kept apart (tests/test_hostsim.py)
inline unsigned skipped_barrier() {
  static const unsigned k = [] {
    const char* e = getenv("HOSTSIM_SKIP_BARRIER");
    return e ? (unsigned)atoi(e) : 0u;
  }();
  return k;
}
inline void sync_block() {
  Block* b = cur_block();
  if (b) {
    static thread_local unsigned calls = 0;
    if (block_phase == 0) calls = 0;
    if (++calls == skipped_barrier()) return;
    pthread_barrier_wait(&b->all[block_phase++ & 1]);
    jitter();
  }
}
inline void sync_wave() {
  Block* b = cur_block();
  if (b) {
    pthread_barrier_wait(&b->wave[threadIdx.x >> 6][wave_phase++ & 1]);
    jitter();
  }
}
inline uint64_t exchange(uint64_t mine, int src) {
  Block* b = cur_block();
  if (!b) return mine;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  b->slot[w][l][0] = mine;
  sync_wave();
  const uint64_t r = b->slot[w][src & 63][0];
  sync_wave();
  return r;
}
// grid of `blocks` workgroups of `threads` threads, one workgroup at a time (raw pthreads: no uninstrumented C++ runtime between the
// sanitizers and the threads they watch)
template <typename F>
struct ThreadArg {
  F* body;
  unsigned tid, bidx, threads, blocks;
};
template <typename F>
void* thread_main(void* p) {
  ThreadArg<F>* a = (ThreadArg<F>*)p;
  threadIdx.x = a->tid;
  blockIdx.x = a->bidx;
  blockDim.x = a->threads;
  gridDim.x = a->blocks;
  block_phase = wave_phase = 0;
  (*a->body)();
  return nullptr;
}
template <typename F>
void launch(unsigned blocks, unsigned threads, F&& body) {
  for (unsigned bidx = 0; bidx < blocks; ++bidx) {
    Block* b = (Block*)calloc(1, sizeof(Block));
    b->nthreads = threads;
    for (int ph = 0; ph < 2; ++ph) {
      pthread_barrier_init(&b->all[ph], nullptr, threads);
      for (unsigned w = 0; w * 64 < threads; ++w) {
        const unsigned n = threads - w * 64 < 64 ? threads - w * 64 : 64;
        pthread_barrier_init(&b->wave[w][ph], nullptr, n);
      }
    }
    cur_block() = b;
    typedef typename std::remove_reference<F>::type Fn;
    ThreadArg<Fn>* args = (ThreadArg<Fn>*)calloc(threads, sizeof(ThreadArg<Fn>));
    pthread_t* ts = (pthread_t*)calloc(threads, sizeof(pthread_t));
    pthread_attr_t attr;
    pthread_attr_init(&attr);
    pthread_attr_setstacksize(&attr, 16u << 20);  // the device code keeps KBs of private arrays per thread; sanitizers add red zones
    for (unsigned tid = 0; tid < threads; ++tid) {
      args[tid].body = &body;
      args[tid].tid = tid;
      args[tid].bidx = bidx;
      args[tid].threads = threads;
      args[tid].blocks = blocks;
      if (pthread_create(&ts[tid], &attr, thread_main<Fn>, &args[tid]) != 0) {
        fprintf(stderr, "hostsim: pthread_create failed at thread %u\n", tid);
        abort();
      }
    }
    for (unsigned tid = 0; tid < threads; ++tid) pthread_join(ts[tid], nullptr);
    pthread_attr_destroy(&attr);
    cur_block() = nullptr;
    for (int ph = 0; ph < 2; ++ph) {
      pthread_barrier_destroy(&b->all[ph]);
      for (unsigned w = 0; w * 64 < threads; ++w) pthread_barrier_destroy(&b->wave[w][ph]);
    }
    free(ts);
    free(args);
    free(b);
  }
}
// lane-per-unit kernels without any cross-lane traffic (the register-resident sweeps): the lanes of a block one after the other
template <typename F>
void launch_serial(unsigned blocks, unsigned threads, F&& body) {
  for (unsigned bidx = 0; bidx < blocks; ++bidx)
    for (unsigned tid = 0; tid < threads; ++tid) {
      threadIdx.x = tid;
      blockIdx.x = bidx;
      blockDim.x = threads;
      gridDim.x = blocks;
      body();
    }
}
}  // namespace hostsim

// ---- the runtime's device functions ------------------------------------------------------------------------------------------
inline void __syncthreads() { hostsim::sync_block(); }
inline double rsqrt(double x) { return 1.0 / sqrt(x); }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float __frcp_rn(float x) { return 1.0f / x; }
inline float __int_as_float(int v) {
  float f;
  memcpy(&f, &v, 4);
  return f;
}
inline int __float_as_int(float f) {
  int v;
  memcpy(&v, &f, 4);
  return v;
}
inline long long __double_as_longlong(double d) {
  long long v;
  memcpy(&v, &d, 8);
  return v;
}
inline double __longlong_as_double(long long v) {
  double d;
  memcpy(&d, &v, 8);
  return d;
}
template <typename T>
inline T __shfl_xor(T v, int mask, int = 64) {
  uint64_t u = 0;
  memcpy(&u, &v, sizeof(T));
  u = hostsim::exchange(u, (int)(threadIdx.x & 63) ^ mask);
  T r;
  memcpy(&r, &u, sizeof(T));
  return r;
}
template <typename T>
inline T __shfl_down(T v, int off, int = 64) {
  uint64_t u = 0;
  memcpy(&u, &v, sizeof(T));
  const int l = threadIdx.x & 63;
  const uint64_t got = hostsim::exchange(u, l + off < 64 ? l + off : l);
  T r;
  memcpy(&r, &got, sizeof(T));
  return r;
}
inline int atomicOr(int* p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
inline int atomicAnd(int* p, int v) { return __atomic_fetch_and(p, v, __ATOMIC_SEQ_CST); }

// ---- compiler builtins of the target -----------------------------------------------------------------------------------------
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_fence(order, scope) __atomic_thread_fence(order)
#define __builtin_amdgcn_wave_barrier() hostsim::sync_wave()
inline double __builtin_amdgcn_rcp(double x) { return 1.0 / x; }
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
inline double __builtin_amdgcn_rsq(double x) { return 1.0 / sqrt(x); }
inline float __builtin_amdgcn_rsqf(float x) { return 1.0f / sqrtf(x); }
inline int __builtin_amdgcn_readlane(int v, int src) { return (int)(uint32_t)hostsim::exchange((uint32_t)v, src); }

// v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32: lane l feeds A[l & 15][l >> 4] and B[l >> 4][l & 15]; D: column l & 15, rows
// (l >> 4) + 4 r in f64, 4 (l >> 4) + r in f32 (cdkf_wg2_kernels.h WgAcc, cdna_hip_programming.md "Fragment layout")
typedef double hostsim_f64x4 __attribute__((ext_vector_type(4)));
typedef float hostsim_f32x4 __attribute__((ext_vector_type(4)));
inline hostsim_f64x4 __builtin_amdgcn_mfma_f64_16x16x4f64(double a, double b, hostsim_f64x4 c, int, int, int) {
  hostsim::Block* blk = hostsim::cur_block();
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  memcpy(&blk->slot[w][l][0], &a, 8);
  memcpy(&blk->slot[w][l][1], &b, 8);
  hostsim::sync_wave();
  const int col = l & 15;
  hostsim_f64x4 d = c;
  for (int r = 0; r < 4; ++r) {
    const int row = (l >> 4) + 4 * r;
    double acc = d[r];
    for (int k = 0; k < 4; ++k) {
      double av, bv;
      memcpy(&av, &blk->slot[w][k * 16 + row][0], 8);
      memcpy(&bv, &blk->slot[w][k * 16 + col][1], 8);
      acc = __builtin_fma(av, bv, acc);
    }
    d[r] = acc;
  }
  hostsim::sync_wave();
  return d;
}
inline hostsim_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, hostsim_f32x4 c, int, int, int) {
  hostsim::Block* blk = hostsim::cur_block();
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  memcpy(&blk->slot[w][l][0], &a, 4);
  memcpy(&blk->slot[w][l][1], &b, 4);
  hostsim::sync_wave();
  const int col = l & 15;
  hostsim_f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * (l >> 4) + r;
    float acc = d[r];
    for (int k = 0; k < 4; ++k) {
      float av, bv;
      memcpy(&av, &blk->slot[w][k * 16 + row][0], 4);
      memcpy(&bv, &blk->slot[w][k * 16 + col][1], 4);
      acc = __builtin_fmaf(av, bv, acc);
    }
    d[r] = acc;
  }
  hostsim::sync_wave();
  return d;
}
