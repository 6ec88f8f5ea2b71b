// cdkf_api.hip -- the C ABI of include/cdkf.h: argument checks, kernel selection, host-buffer wrappers.
#include "cdkf_host.h"
#include "cdkf_launch.h"

namespace cdkf {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_common(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const void* t, const void* y,
                 const void* ll) {
  if (!mdl || !o) {
    set_error("model and opts must not be NULL");
    return CDKF_EINVAL;
  }
  if (N < 0 || T < 1) {
    set_error("need N >= 0 and T >= 1 (got N=%lld T=%lld)", (long long)N, (long long)T);
    return CDKF_EINVAL;
  }
  if (!t || !ll || (!y && !(o && o->forecast))) {
    set_error("t, y and ll must not be NULL (y may be NULL only in forecast mode)");
    return CDKF_EINVAL;
  }
  if (mdl->state_dim < 1 || mdl->emission_dim < 1) {
    set_error("state_dim and emission_dim must be >= 1");
    return CDKF_EINVAL;
  }
  if (!mdl->theta || !mdl->L || !mdl->Qc || !mdl->H || !mdl->h_bias || !mdl->R || !mdl->m0 || !mdl->P0) {
    set_error("model parameter pointers must not be NULL");
    return CDKF_EINVAL;
  }
  if (o->state_order < 0 || o->state_order > 2) {
    set_error("EKF hyperparams.state_order = %d not implemented yet", o->state_order);
    return CDKF_EINVAL;
  }
  if (o->layout != CDKF_LAYOUT_NT && o->layout != CDKF_LAYOUT_TN && o->layout != CDKF_LAYOUT_TCN) {
    set_error("opts.layout must be CDKF_LAYOUT_NT, CDKF_LAYOUT_TN or CDKF_LAYOUT_TCN");
    return CDKF_EINVAL;
  }
  if (o->num_iter < 1 || !(o->dt0 > 0) || o->max_steps < 1) {
    set_error("need num_iter >= 1, dt0 > 0, max_steps >= 1");
    return CDKF_EINVAL;
  }
  return CDKF_OK;
}

int select_device(const cdkf_opts* o) {
  if (o && o->device >= 0) CDKF_HIP_CHECK(hipSetDevice(o->device));
  return CDKF_OK;
}

// ---- host-buffer wrapper: allocate, upload, run the _dev path, download -------------------------
template <typename R, typename DevFn>
int run_with_host_buffers(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y,
                          R* ll, R* o1m, R* o1P, R* o2m, R* o2P, int32_t* status, DevFn fn) {
  int rc = check_common(mdl, o, N, T, t, y, ll);
  if (rc) return rc;
  if (N == 0) return CDKF_OK;
  rc = select_device(o);
  if (rc) return rc;
  const int d = mdl->state_dim, m = mdl->emission_dim;
  const size_t nt = (size_t)(o->t_shared ? T : N * T);
  const size_t nm = (size_t)N * T * d, nP = nm * d;
  DevBuf dt, dy, dll, d1m, d1P, d2m, d2P, dst;
  if ((rc = dt.alloc(nt * sizeof(R))) || (rc = dy.alloc((size_t)N * T * m * sizeof(R))) ||
      (rc = dll.alloc(N * sizeof(R))) || (rc = dst.alloc(N * sizeof(int32_t))))
    return rc;
  if (o1m && (rc = d1m.alloc(nm * sizeof(R)))) return rc;
  if (o1P && (rc = d1P.alloc(nP * sizeof(R)))) return rc;
  if (o2m && (rc = d2m.alloc(nm * sizeof(R)))) return rc;
  if (o2P && (rc = d2P.alloc(nP * sizeof(R)))) return rc;
  CDKF_HIP_CHECK(hipMemcpy(dt.p, t, nt * sizeof(R), hipMemcpyHostToDevice));
  if (y) CDKF_HIP_CHECK(hipMemcpy(dy.p, y, (size_t)N * T * m * sizeof(R), hipMemcpyHostToDevice));
  rc = fn(mdl, o, N, T, (const R*)dt.p, (const R*)dy.p, (R*)dll.p, (R*)d1m.p, (R*)d1P.p, (R*)d2m.p, (R*)d2P.p,
          (int32_t*)dst.p, (void*)nullptr);
  if (rc) return rc;
  CDKF_HIP_CHECK(hipDeviceSynchronize());
  CDKF_HIP_CHECK(hipMemcpy(ll, dll.p, N * sizeof(R), hipMemcpyDeviceToHost));
  if (status) CDKF_HIP_CHECK(hipMemcpy(status, dst.p, N * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (o1m) CDKF_HIP_CHECK(hipMemcpy(o1m, d1m.p, nm * sizeof(R), hipMemcpyDeviceToHost));
  if (o1P) CDKF_HIP_CHECK(hipMemcpy(o1P, d1P.p, nP * sizeof(R), hipMemcpyDeviceToHost));
  if (o2m) CDKF_HIP_CHECK(hipMemcpy(o2m, d2m.p, nm * sizeof(R), hipMemcpyDeviceToHost));
  if (o2P) CDKF_HIP_CHECK(hipMemcpy(o2P, d2P.p, nP * sizeof(R), hipMemcpyDeviceToHost));
  return CDKF_OK;
}

// ---- sum of per-trajectory log-likelihoods ----------------------------------------------------
template <typename R>
__global__ __launch_bounds__(256) void ll_sum_kernel(const R* __restrict__ ll, long N, double* __restrict__ out) {
  __shared__ double part[4];
  double s = 0.0;
  for (long i = threadIdx.x; i < N; i += 256) s += (double)ll[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (part[0] + part[1]) + (part[2] + part[3]);
}

template <typename R>
int ll_sum_dev(const R* ll, int64_t N, double* out, void* stream) {
  if (!ll || !out || N < 0) {
    set_error("ll_sum: bad arguments");
    return CDKF_EINVAL;
  }
  hipLaunchKernelGGL(ll_sum_kernel<R>, dim3(1), dim3(256), 0, (hipStream_t)stream, ll, (long)N, out);
  CDKF_HIP_CHECK(hipGetLastError());
  return CDKF_OK;
}

}  // namespace cdkf

using namespace cdkf;

extern "C" {

void cdkf_default_opts(cdkf_opts* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->state_order = CDKF_ORDER_SECOND;
  o->num_iter = 1;
  o->t_shared = 0;
  o->device = -1;
  o->layout = CDKF_LAYOUT_NT;
  o->forecast = 0;
  o->max_steps = 100000;
  o->dt0 = 0.01;
  o->dt_final = 1e-10;
  o->cov_rescaling = 1.0;
  o->ukf_alpha = std::sqrt(3.0);
  o->ukf_beta = 2.0;
  o->ukf_kappa = 1.0;
}

int cdkf_version(void) { return CDKF_VERSION; }
const char* cdkf_last_error(void) { return g_err; }

int cdkf_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
    return CDKF_EHIP;
  }
  return n;
}

int cdkf_supported(const cdkf_model* mdl, const cdkf_opts* o, int algo, int bytes_per_real) {
  if (!mdl || !o || (bytes_per_real != 4 && bytes_per_real != 8) || algo < 0 || algo > 2) return 0;
  return kernel_available(mdl, o, algo, bytes_per_real) ? 1 : 0;
}

int cdkf_preferred_layout(const cdkf_model* mdl) {
  return (mdl && reg_shape_available(mdl)) ? CDKF_LAYOUT_TCN : CDKF_LAYOUT_TN;
}

int cdkf_malloc(void** p, int64_t bytes) {
  if (!p || bytes < 0) {
    set_error("cdkf_malloc: bad arguments");
    return CDKF_EINVAL;
  }
  CDKF_HIP_CHECK(hipMalloc(p, bytes ? (size_t)bytes : 1));
  return CDKF_OK;
}
int cdkf_free(void* p) {
  if (p) CDKF_HIP_CHECK(hipFree(p));
  return CDKF_OK;
}
int cdkf_memcpy_h2d(void* d, const void* h, int64_t bytes) {
  if (bytes < 0 || (bytes > 0 && (!d || !h))) {
    set_error("cdkf_memcpy_h2d: bad arguments");
    return CDKF_EINVAL;
  }
  CDKF_HIP_CHECK(hipMemcpy(d, h, (size_t)bytes, hipMemcpyHostToDevice));
  return CDKF_OK;
}
int cdkf_memcpy_d2h(void* h, const void* d, int64_t bytes) {
  if (bytes < 0 || (bytes > 0 && (!d || !h))) {
    set_error("cdkf_memcpy_d2h: bad arguments");
    return CDKF_EINVAL;
  }
  CDKF_HIP_CHECK(hipMemcpy(h, d, (size_t)bytes, hipMemcpyDeviceToHost));
  return CDKF_OK;
}
int cdkf_memset(void* d, int value, int64_t bytes) {
  if (bytes < 0 || (bytes > 0 && !d)) {
    set_error("cdkf_memset: bad arguments");
    return CDKF_EINVAL;
  }
  CDKF_HIP_CHECK(hipMemset(d, value, (size_t)bytes));
  return CDKF_OK;
}
int cdkf_synchronize(void* stream) {
  CDKF_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  return CDKF_OK;
}

#define CDKF_DEFINE_ALGO(NAME, SUFFIX, RTYPE, LAUNCH)                                                              \
  int cdkf_##NAME##_##SUFFIX##_dev(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const RTYPE* t, \
                                   const RTYPE* y, RTYPE* ll, RTYPE* a1, RTYPE* a2, RTYPE* a3, RTYPE* a4,          \
                                   int32_t* status, void* stream) {                                                \
    int rc = check_common(mdl, o, N, T, t, y, ll);                                                                 \
    if (rc) return rc;                                                                                             \
    if (N == 0) return CDKF_OK;                                                                                    \
    if ((rc = select_device(o))) return rc;                                                                        \
    return LAUNCH<RTYPE>(mdl, o, N, T, t, y, ll, a1, a2, a3, a4, status, (hipStream_t)stream);                     \
  }                                                                                                                \
  int cdkf_##NAME##_##SUFFIX(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const RTYPE* t,      \
                             const RTYPE* y, RTYPE* ll, RTYPE* a1, RTYPE* a2, RTYPE* a3, RTYPE* a4,                \
                             int32_t* status) {                                                                    \
    return run_with_host_buffers<RTYPE>(mdl, o, N, T, t, y, ll, a1, a2, a3, a4, status,                            \
                                        cdkf_##NAME##_##SUFFIX##_dev);                                             \
  }

CDKF_DEFINE_ALGO(ekf_filter, f64, double, launch_ekf_filter)
CDKF_DEFINE_ALGO(ekf_filter, f32, float, launch_ekf_filter)
CDKF_DEFINE_ALGO(ukf_filter, f64, double, launch_ukf_filter)
CDKF_DEFINE_ALGO(ukf_filter, f32, float, launch_ukf_filter)
CDKF_DEFINE_ALGO(ekf_smoother, f64, double, launch_ekf_smoother)
CDKF_DEFINE_ALGO(ekf_smoother, f32, float, launch_ekf_smoother)

int cdkf_ll_sum_f64_dev(const double* ll, int64_t N, double* out, void* stream) {
  return ll_sum_dev<double>(ll, N, out, stream);
}
int cdkf_ll_sum_f32_dev(const float* ll, int64_t N, double* out, void* stream) {
  return ll_sum_dev<float>(ll, N, out, stream);
}

}  // extern "C"
