// cdkf_api.hip -- the C ABI of include/cdkf.h: argument checks, kernel selection, host-buffer wrappers.
#include <cstdlib>

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

static thread_local char g_kernel[192] = "";

void note_kernel(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
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
  {  // the drift's parameter block is read with the size the drift implies: a short block from a C caller must not be read past
    const long d = mdl->state_dim, h1 = mdl->hidden1, h2 = mdl->hidden2;
    long want = -1;
    switch (mdl->drift_kind) {
      case CDKF_DRIFT_LINEAR: want = d * d + d; break;
      case CDKF_DRIFT_LORENZ63: want = 3; break;
      case CDKF_DRIFT_LORENZ96: want = 1; break;
      case CDKF_DRIFT_MLP_TANH: want = (h1 >= 1 && h2 >= 1) ? h1 * d + h1 + h2 * h1 + h2 + d * h2 + d : -2; break;
      default: break;  // custom drifts: checked against the registered size where they are looked up
    }
    if (want == -2) {
      set_error("drift MLP_TANH needs hidden1, hidden2 >= 1 (got %d, %d)", mdl->hidden1, mdl->hidden2);
      return CDKF_EINVAL;
    }
    if (want >= 0 && mdl->n_theta != want) {
      set_error("drift_kind %d with state_dim %d takes n_theta = %ld parameters (got %lld)", mdl->drift_kind, mdl->state_dim, want,
                (long long)mdl->n_theta);
      return CDKF_EINVAL;
    }
  }
  if (mdl->emission_kind != 0 && !custom_kind(mdl->drift_kind)) {
    set_error("emission_kind %d: a custom emission runs on the run-time compiled kernels, which need the drift as source too "
              "(cdkf_custom_drift_register; the Python host converts the built-in drifts)", mdl->emission_kind);
    return CDKF_EUNSUPPORTED;
  }
  if (o->state_order < 0 || o->state_order > 2) {
    set_error("EKF hyperparams.state_order = %d not implemented yet", o->state_order);
    return CDKF_EINVAL;
  }
  if (o->layout != CDKF_LAYOUT_NT && o->layout != CDKF_LAYOUT_TN && o->layout != CDKF_LAYOUT_TCN) {
    set_error("opts.layout must be CDKF_LAYOUT_NT, CDKF_LAYOUT_TN or CDKF_LAYOUT_TCN");
    return CDKF_EINVAL;
  }
  if (o->layout_in != CDKF_LAYOUT_SAME && o->layout_in != CDKF_LAYOUT_NT && o->layout_in != CDKF_LAYOUT_TN &&
      o->layout_in != CDKF_LAYOUT_TCN) {
    set_error("opts.layout_in must be CDKF_LAYOUT_SAME or one of the CDKF_LAYOUT_* values");
    return CDKF_EINVAL;
  }
  if (o->solver < CDKF_SOLVER_DOPRI5 || o->solver > CDKF_SOLVER_EULER || (o->adaptive != 0 && o->adaptive != 1)) {
    set_error("opts.solver must be one of CDKF_SOLVER_* (got %d) and opts.adaptive 0 or 1 (got %d)", o->solver, o->adaptive);
    return CDKF_EINVAL;
  }
  if (o->adaptive) {
    const bool has_estimate = o->solver == CDKF_SOLVER_DOPRI5 || o->solver == CDKF_SOLVER_TSIT5 || o->solver == CDKF_SOLVER_BOSH3 ||
                              o->solver == CDKF_SOLVER_HEUN;
    if (!has_estimate || !(o->rtol >= 0) || !(o->atol >= 0) || !(o->rtol + o->atol > 0)) {
      set_error("adaptive stepping needs a method with an embedded error estimate (DOPRI5, TSIT5, BOSH3, HEUN) and rtol, atol >= 0 "
                "not both zero (solver %d, rtol %g, atol %g)", o->solver, o->rtol, o->atol);
      return CDKF_EINVAL;
    }
    if (!(o->dtmin >= 0) || !(o->dtmax >= 0) || (o->dtmax > 0 && !(o->dtmin <= o->dtmax))) {
      set_error("adaptive stepping: need 0 <= dtmin <= dtmax (got dtmin %g, dtmax %g; defaults 0 and infinity, dtmax 0 = no bound)", o->dtmin, o->dtmax);
      return CDKF_EINVAL;
    }
    if (!(o->pid_safety >= 0) || !(o->pid_factormin >= 0) || !(o->pid_factormax >= 0) || (o->pid_factormin > 1) ||
        (o->pid_factormax > 0 && o->pid_factormax < 1)) {
      set_error("adaptive stepping: need safety > 0, 0 < factormin <= 1 <= factormax (got %g, %g, %g; 0 = the defaults 0.9, 0.2, 10)",
                o->pid_safety, o->pid_factormin, o->pid_factormax);
      return CDKF_EINVAL;
    }
  }
  if (o->flags & ~CDKF_FLAG_UKF_SIGMA_POINTS) {
    set_error("opts.flags = 0x%x has bits this library version does not know (CDKF_FLAG_*)", (unsigned)o->flags);
    return CDKF_EINVAL;
  }
  if (mdl->input_dim < 0 || mdl->input_dim > 64) {
    set_error("model.input_dim must be in 0 .. 64 (got %d)", mdl->input_dim);
    return CDKF_EINVAL;
  }
  if (o->num_iter < 1 || !(o->dt0 > 0) || o->max_steps < 1) {
    set_error("need num_iter >= 1, dt0 > 0, max_steps >= 1");
    return CDKF_EINVAL;
  }
  return CDKF_OK;
}


// Distinct trajectories per wavefront for the lane-per-trajectory sweeps (cdkf_filter_reg_body.inc): the largest power of two
// that still yields two wavefronts per CU of the current device, 64 once the batch is that large.  CDKF_LANES_PER_WAVE
// overrides (tuning / tests).
int reg_lanes_per_wave(int64_t N) {
  static const int forced = [] {
    const char* e = std::getenv("CDKF_LANES_PER_WAVE");
    const int v = e ? std::atoi(e) : 0;
    return (v >= 1 && v <= 64 && (v & (v - 1)) == 0) ? v : 0;
  }();
  if (forced) return forced;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    cus = 256;
  const int64_t waves = 2 * (int64_t)cus;
  int lanes = 64;
  while (lanes > 1 && (N + lanes - 1) / lanes < waves) lanes >>= 1;
  return lanes;
}

// ---- host-buffer wrapper: allocate, upload, run the _dev path, download -------------------------
template <typename R, typename DevFn>
int run_with_host_buffers(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y,
                          R* ll, R* o1m, R* o1P, R* o2m, R* o2P, int32_t* status, DevFn fn) {
  int rc = check_common(mdl, o, N, T, t, y, ll);
  if (rc) return rc;
  if (N == 0) return CDKF_OK;
  CDKF_SELECT_DEVICE(o);
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
  // inputs [N,T,d_u]: resident where y is -- uploaded here, the sweep gets the device copy through its own opts
  DevBuf du_;
  cdkf_opts od = *o;
  if (mdl->input_dim > 0 && o->inputs) {
    const size_t nu = (size_t)N * T * mdl->input_dim * sizeof(R);
    if ((rc = du_.alloc(nu))) return rc;
    CDKF_HIP_CHECK(hipMemcpy(du_.p, o->inputs, nu, hipMemcpyHostToDevice));
    od.inputs = du_.p;
  }
  o = &od;
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

// ---- column sums of the per-trajectory gradient [N, P]: one workgroup per parameter ---------------------------
template <typename R>
__global__ __launch_bounds__(256) void grad_sum_kernel(const R* __restrict__ g, long N, int P, double* __restrict__ out) {
  __shared__ double part[4];
  const int p = blockIdx.x;
  double s = 0.0;
  for (long i = threadIdx.x; i < N; i += 256) s += (double)g[i * P + p];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[p] = (part[0] + part[1]) + (part[2] + part[3]);
}

template <typename R>
int grad_sum_dev(const R* g, int64_t N, int64_t P, double* out, void* stream) {
  if (!g || !out || N < 0 || P < 1) {
    set_error("grad_sum: bad arguments");
    return CDKF_EINVAL;
  }
  hipLaunchKernelGGL(grad_sum_kernel<R>, dim3((unsigned)P), dim3(256), 0, (hipStream_t)stream, g, (long)N, (int)P, out);
  CDKF_HIP_CHECK(hipGetLastError());
  return CDKF_OK;
}

template <typename R>
int loglik_grad_dev(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll,
                    R* grad, R* grad_model, int32_t* status, void* stream, bool ukf = false) {
  int rc = check_common(mdl, o, N, T, t, y, ll);
  if (rc) return rc;
  if (!grad) {
    set_error("loglik_grad: grad must not be NULL");
    return CDKF_EINVAL;
  }
  if (N == 0) return CDKF_OK;
  CDKF_SELECT_DEVICE(o);
  if (ukf && grad_model) return launch_ukf_grad_all<R>(mdl, o, N, T, t, y, ll, grad, grad_model, status, (hipStream_t)stream);
  if (ukf) return launch_ukf_grad<R>(mdl, o, N, T, t, y, ll, grad, status, (hipStream_t)stream);
  return launch_ekf_grad<R>(mdl, o, N, T, t, y, ll, grad, grad_model, status, (hipStream_t)stream);
}

template <typename R>
int loglik_grad_host(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll,
                     R* grad, R* grad_model, int32_t* status, bool ukf = false) {
  int rc = check_common(mdl, o, N, T, t, y, ll);
  if (rc) return rc;
  if (!grad) {
    set_error("loglik_grad: grad must not be NULL");
    return CDKF_EINVAL;
  }
  if (N == 0) return CDKF_OK;
  CDKF_SELECT_DEVICE(o);
  const size_t nt = (size_t)(o->t_shared ? T : N * T), ny = (size_t)N * T * mdl->emission_dim;
  const size_t ng = (size_t)N * (size_t)mdl->n_theta;
  const size_t dd = mdl->state_dim, mm = mdl->emission_dim;
  const size_t ngm = (size_t)N * (dd + 2 * dd * dd + mm * dd + mm + mm * mm);
  DevBuf dt, dy, dll, dg, dst, dgm;
  if (grad_model && (rc = dgm.alloc(ngm * sizeof(R)))) return rc;
  if ((rc = dt.alloc(nt * sizeof(R))) || (rc = dy.alloc(ny * sizeof(R))) || (rc = dll.alloc(N * sizeof(R))) ||
      (rc = dg.alloc(ng * sizeof(R))) || (rc = dst.alloc(N * sizeof(int32_t))))
    return rc;
  CDKF_HIP_CHECK(hipMemcpy(dt.p, t, nt * sizeof(R), hipMemcpyHostToDevice));
  CDKF_HIP_CHECK(hipMemcpy(dy.p, y, ny * sizeof(R), hipMemcpyHostToDevice));
  DevBuf du_;  // inputs [N,T,d_u]: as in run_with_host_buffers
  cdkf_opts od = *o;
  if (mdl->input_dim > 0 && o->inputs) {
    const size_t nu = (size_t)N * T * mdl->input_dim * sizeof(R);
    if ((rc = du_.alloc(nu))) return rc;
    CDKF_HIP_CHECK(hipMemcpy(du_.p, o->inputs, nu, hipMemcpyHostToDevice));
    od.inputs = du_.p;
  }
  o = &od;
  rc = (ukf && grad_model) ? launch_ukf_grad_all<R>(mdl, o, N, T, (const R*)dt.p, (const R*)dy.p, (R*)dll.p, (R*)dg.p, (R*)dgm.p,
                                                   (int32_t*)dst.p, nullptr)
       : ukf ? launch_ukf_grad<R>(mdl, o, N, T, (const R*)dt.p, (const R*)dy.p, (R*)dll.p, (R*)dg.p, (int32_t*)dst.p, nullptr)
           : launch_ekf_grad<R>(mdl, o, N, T, (const R*)dt.p, (const R*)dy.p, (R*)dll.p, (R*)dg.p, (R*)dgm.p, (int32_t*)dst.p,
                                nullptr);
  if (rc) return rc;
  CDKF_HIP_CHECK(hipDeviceSynchronize());
  CDKF_HIP_CHECK(hipMemcpy(ll, dll.p, N * sizeof(R), hipMemcpyDeviceToHost));
  CDKF_HIP_CHECK(hipMemcpy(grad, dg.p, ng * sizeof(R), hipMemcpyDeviceToHost));
  if (grad_model) CDKF_HIP_CHECK(hipMemcpy(grad_model, dgm.p, ngm * sizeof(R), hipMemcpyDeviceToHost));
  if (status) CDKF_HIP_CHECK(hipMemcpy(status, dst.p, N * sizeof(int32_t), hipMemcpyDeviceToHost));
  return CDKF_OK;
}

// ---- value + every gradient with per-step jumps of the predicted mean (host buffers; the linear front-end's bias / inputs) -----------
template <typename R>
int loglik_grad_jumps_host(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, const R* jumps,
                           R* ll, R* grad, R* grad_model, R* grad_jumps, R* grad_y, int32_t* status) {
  int rc = check_common(mdl, o, N, T, t, y, ll);
  if (rc) return rc;
  if (!grad || !grad_model || !jumps || !grad_jumps || !grad_y) {
    set_error("loglik_grad_jumps: grad, grad_model, jumps, grad_jumps and grad_y must not be NULL");
    return CDKF_EINVAL;
  }
  if (N == 0) return CDKF_OK;
  CDKF_SELECT_DEVICE(o);
  const size_t dd = mdl->state_dim, mm = mdl->emission_dim;
  const size_t nt = (size_t)(o->t_shared ? T : N * T), ny = (size_t)N * T * mm, nj = (size_t)N * T * dd;
  const size_t ng = (size_t)N * (size_t)(mdl->n_theta > 0 ? mdl->n_theta : 1), ngm = (size_t)N * (dd + 2 * dd * dd + mm * dd + mm + mm * mm);
  DevBuf dt, dy, dj, dll, dg, dgm, dgj, dgy, dst;
  if ((rc = dt.alloc(nt * sizeof(R))) || (rc = dy.alloc(ny * sizeof(R))) || (rc = dj.alloc(nj * sizeof(R))) ||
      (rc = dll.alloc(N * sizeof(R))) || (rc = dg.alloc(ng * sizeof(R))) || (rc = dgm.alloc(ngm * sizeof(R))) ||
      (rc = dgj.alloc(nj * sizeof(R))) || (rc = dgy.alloc(ny * sizeof(R))) || (rc = dst.alloc(N * sizeof(int32_t))))
    return rc;
  CDKF_HIP_CHECK(hipMemcpy(dt.p, t, nt * sizeof(R), hipMemcpyHostToDevice));
  CDKF_HIP_CHECK(hipMemcpy(dy.p, y, ny * sizeof(R), hipMemcpyHostToDevice));
  CDKF_HIP_CHECK(hipMemcpy(dj.p, jumps, nj * sizeof(R), hipMemcpyHostToDevice));
  rc = launch_ekf_grad_adjoint_jumps<R>(mdl, o, N, T, (const R*)dt.p, (const R*)dy.p, (const R*)dj.p, (R*)dll.p, (R*)dg.p, (R*)dgm.p,
                                        (R*)dgj.p, (R*)dgy.p, (int32_t*)dst.p, nullptr);
  if (rc) return rc;
  CDKF_HIP_CHECK(hipDeviceSynchronize());
  CDKF_HIP_CHECK(hipMemcpy(ll, dll.p, N * sizeof(R), hipMemcpyDeviceToHost));
  CDKF_HIP_CHECK(hipMemcpy(grad, dg.p, (size_t)N * (size_t)mdl->n_theta * sizeof(R), hipMemcpyDeviceToHost));
  CDKF_HIP_CHECK(hipMemcpy(grad_model, dgm.p, ngm * sizeof(R), hipMemcpyDeviceToHost));
  CDKF_HIP_CHECK(hipMemcpy(grad_jumps, dgj.p, nj * sizeof(R), hipMemcpyDeviceToHost));
  CDKF_HIP_CHECK(hipMemcpy(grad_y, dgy.p, ny * sizeof(R), hipMemcpyDeviceToHost));
  if (status) CDKF_HIP_CHECK(hipMemcpy(status, dst.p, N * sizeof(int32_t), hipMemcpyDeviceToHost));
  return CDKF_OK;
}

// ---- linear model, smoother type 1 -------------------------------------------------------------------------------------
template <typename R>
int kf_smoother1_dev(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll, R* fm,
                     R* fP, R* sm, R* sP, R* cross, int32_t* status, void* stream) {
  int rc = check_common(mdl, o, N, T, t, y, ll);
  if (rc) return rc;
  if (N == 0) return CDKF_OK;
  CDKF_SELECT_DEVICE(o);
  return launch_kf_smoother1<R>(mdl, o, N, T, t, y, ll, fm, fP, sm, sP, cross, status, (hipStream_t)stream);
}

template <typename R>
int kf_smoother1_host(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll, R* fm,
                      R* fP, R* sm, R* sP, R* cross, int32_t* status) {
  int rc = check_common(mdl, o, N, T, t, y, ll);
  if (rc) return rc;
  if (!fm || !fP || !sm || !sP) {
    set_error("kf_smoother1: filtered and smoothed output arrays must not be NULL");
    return CDKF_EINVAL;
  }
  if (N == 0) return CDKF_OK;
  CDKF_SELECT_DEVICE(o);
  const size_t d = mdl->state_dim, m = mdl->emission_dim;
  const size_t nt = (size_t)(o->t_shared ? T : N * T), ny = (size_t)N * T * m, nm = (size_t)N * T * d, nP = nm * d;
  DevBuf dt, dy, dll, dst, dfm, dfP, dsm, dsP, dcr;
  if ((rc = dt.alloc(nt * sizeof(R))) || (rc = dy.alloc(ny * sizeof(R))) || (rc = dll.alloc(N * sizeof(R))) ||
      (rc = dst.alloc(N * sizeof(int32_t))) || (rc = dfm.alloc(nm * sizeof(R))) || (rc = dfP.alloc(nP * sizeof(R))) ||
      (rc = dsm.alloc(nm * sizeof(R))) || (rc = dsP.alloc(nP * sizeof(R))))
    return rc;
  if (cross && (rc = dcr.alloc(nP * sizeof(R)))) return rc;
  CDKF_HIP_CHECK(hipMemcpy(dt.p, t, nt * sizeof(R), hipMemcpyHostToDevice));
  CDKF_HIP_CHECK(hipMemcpy(dy.p, y, ny * sizeof(R), hipMemcpyHostToDevice));
  if (cross) CDKF_HIP_CHECK(hipMemset(dcr.p, 0, nP * sizeof(R)));
  rc = launch_kf_smoother1<R>(mdl, o, N, T, (const R*)dt.p, (const R*)dy.p, (R*)dll.p, (R*)dfm.p, (R*)dfP.p, (R*)dsm.p,
                              (R*)dsP.p, (R*)dcr.p, (int32_t*)dst.p, nullptr);
  if (rc) return rc;
  CDKF_HIP_CHECK(hipDeviceSynchronize());
  CDKF_HIP_CHECK(hipMemcpy(ll, dll.p, N * sizeof(R), hipMemcpyDeviceToHost));
  if (status) CDKF_HIP_CHECK(hipMemcpy(status, dst.p, N * sizeof(int32_t), hipMemcpyDeviceToHost));
  CDKF_HIP_CHECK(hipMemcpy(fm, dfm.p, nm * sizeof(R), hipMemcpyDeviceToHost));
  CDKF_HIP_CHECK(hipMemcpy(fP, dfP.p, nP * sizeof(R), hipMemcpyDeviceToHost));
  CDKF_HIP_CHECK(hipMemcpy(sm, dsm.p, nm * sizeof(R), hipMemcpyDeviceToHost));
  CDKF_HIP_CHECK(hipMemcpy(sP, dsP.p, nP * sizeof(R), hipMemcpyDeviceToHost));
  if (cross) CDKF_HIP_CHECK(hipMemcpy(cross, dcr.p, nP * sizeof(R), hipMemcpyDeviceToHost));
  return CDKF_OK;
}

// ---- linear model: the pushed-forward (A, Q) of every observation interval ------------------------------------------------
template <typename R>
int kf_pushforward_host(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, R* AQ) {
  R dummy = 0;
  int rc = check_common(mdl, o, N, T, t, &dummy, &dummy);
  if (rc) return rc;
  if (!AQ) {
    set_error("kf_pushforward: the output array must not be NULL");
    return CDKF_EINVAL;
  }
  if (N == 0 || T < 2) return CDKF_OK;
  CDKF_SELECT_DEVICE(o);
  const size_t d = mdl->state_dim, nt = (size_t)(o->t_shared ? T : N * T), nA = (size_t)N * (size_t)(T - 1) * 2 * d * d;
  DevBuf dt, dA, dst;
  if ((rc = dt.alloc(nt * sizeof(R))) || (rc = dA.alloc(nA * sizeof(R))) || (rc = dst.alloc(N * sizeof(int32_t)))) return rc;
  CDKF_HIP_CHECK(hipMemcpy(dt.p, t, nt * sizeof(R), hipMemcpyHostToDevice));
  CDKF_HIP_CHECK(hipMemset(dst.p, 0, N * sizeof(int32_t)));
  rc = launch_kf_pushforward<R>(mdl, o, N, T, (const R*)dt.p, (R*)dA.p, (int32_t*)dst.p, nullptr);
  if (rc) return rc;
  CDKF_HIP_CHECK(hipDeviceSynchronize());
  CDKF_HIP_CHECK(hipMemcpy(AQ, dA.p, nA * sizeof(R), hipMemcpyDeviceToHost));
  // an interval cut off at max_steps leaves a truncated (A, Q) pair: that is an error for this call -- it has no status output, and its
  // callers (the linear front end's dynamics-bias / input offsets) would build on the pair silently (diffrax raises in the reference)
  std::vector<int32_t> st((size_t)N);
  CDKF_HIP_CHECK(hipMemcpy(st.data(), dst.p, N * sizeof(int32_t), hipMemcpyDeviceToHost));
  for (int64_t n = 0; n < N; ++n)
    if (st[(size_t)n] & CDKF_STATUS_MAX_STEPS) {
      set_error("kf_pushforward: an interval of trajectory %lld needs more than max_steps = %lld Runge-Kutta steps of dt0 = %g: the "
                "pushed-forward (A, Q) would be truncated", (long long)n, (long long)o->max_steps, o->dt0);
      return CDKF_EINVAL;
    }
  return CDKF_OK;
}

// ---- emission moments: one workgroup (64 threads) per state marginal, H and the d x d covariance staged in LDS ----
template <typename R>
__global__ __launch_bounds__(64) void emission_moments_kernel(int d, int m, const R* __restrict__ par, long rows,
                                                              const R* __restrict__ means, const R* __restrict__ covs,
                                                              R* __restrict__ out_mean, R* __restrict__ out_cov) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  R* P = reinterpret_cast<R*>(smem_raw);  // [d*d]
  R* HP = P + d * d;                      // [m*d]
  const R* H = par;                       // [m*d]
  const R* hb = par + m * d;              // [m]
  const R* Rm = hb + m;                   // [m*m]
  for (long row = blockIdx.x; row < rows; row += gridDim.x) {
    const R* mu = means + row * d;
    for (int r = threadIdx.x; r < m; r += 64) {
      R s = hb[r];
      for (int k = 0; k < d; ++k) s = rfma(H[r * d + k], mu[k], s);
      out_mean[row * m + r] = s;
    }
    if (covs && out_cov) {
      for (int e = threadIdx.x; e < d * d; e += 64) P[e] = covs[row * d * d + e];
      __syncthreads();
      for (int e = threadIdx.x; e < m * d; e += 64) {
        const int r = e / d, j = e - r * d;
        R s = 0;
        for (int k = 0; k < d; ++k) s = rfma(H[r * d + k], P[k * d + j], s);
        HP[e] = s;
      }
      __syncthreads();
      for (int e = threadIdx.x; e < m * m; e += 64) {
        const int r = e / m, c = e - r * m;
        R s = 0;
        for (int k = 0; k < d; ++k) s = rfma(HP[r * d + k], H[c * d + k], s);
        out_cov[row * m * m + e] = s + Rm[e];
      }
      __syncthreads();
    }
  }
}

template <typename R>
int emission_moments_dev(const cdkf_model* mdl, int64_t rows, const R* means, const R* covs, R* out_mean, R* out_cov,
                         hipStream_t stream) {
  if (!mdl || !mdl->H || !mdl->h_bias || !mdl->R || rows < 0 || (rows > 0 && (!means || !out_mean))) {
    set_error("emission_moments: bad arguments");
    return CDKF_EINVAL;
  }
  if (rows == 0) return CDKF_OK;
  const int d = mdl->state_dim, m = mdl->emission_dim;
  if (d < 1 || m < 1 || (size_t)(d * d + m * d) * sizeof(R) > 150 * 1024) {
    set_error("emission_moments: unsupported dimensions d=%d m=%d", d, m);
    return CDKF_EUNSUPPORTED;
  }
  std::vector<R> h;
  for (int i = 0; i < m * d; ++i) h.push_back(R(mdl->H[i]));
  for (int i = 0; i < m; ++i) h.push_back(R(mdl->h_bias[i]));
  for (int i = 0; i < m * m; ++i) h.push_back(R(mdl->R[i]));
  R* par = nullptr;  // small, synchronous upload: this entry point is not on the sweep path
  CDKF_HIP_CHECK(hipMalloc((void**)&par, h.size() * sizeof(R)));
  hipError_t e = hipMemcpy(par, h.data(), h.size() * sizeof(R), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    const size_t lds = (size_t)(d * d + m * d) * sizeof(R);
    if (lds > 48 * 1024)
      e = hipFuncSetAttribute((const void*)emission_moments_kernel<R>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    if (e == hipSuccess) {
      const unsigned blocks = (unsigned)(rows < 65536 ? rows : 65536);
      hipLaunchKernelGGL(emission_moments_kernel<R>, dim3(blocks), dim3(64), lds, stream, d, m, (const R*)par, (long)rows, means,
                         covs, out_mean, out_cov);
      e = hipGetLastError();
      if (e == hipSuccess) e = hipStreamSynchronize(stream);
    }
  }
  (void)hipFree(par);
  if (e != hipSuccess) {
    set_error("emission_moments failed: %s", hipGetErrorString(e));
    return CDKF_EHIP;
  }
  return CDKF_OK;
}

template <typename R>
int emission_moments_host(const cdkf_model* mdl, int64_t rows, const R* means, const R* covs, R* out_mean, R* out_cov) {
  if (!mdl || rows < 0 || (rows > 0 && (!means || !out_mean))) {
    set_error("emission_moments: bad arguments");
    return CDKF_EINVAL;
  }
  if (mdl->emission_kind != 0) {
    set_error("emission_moments: linear emission only (emission_kind %d)", mdl->emission_kind);
    return CDKF_EUNSUPPORTED;
  }
  if (rows == 0) return CDKF_OK;
  const size_t d = mdl->state_dim, m = mdl->emission_dim;
  DevBuf dm, dP, om, oP;
  int rc;
  if ((rc = dm.alloc(rows * d * sizeof(R))) || (rc = om.alloc(rows * m * sizeof(R)))) return rc;
  CDKF_HIP_CHECK(hipMemcpy(dm.p, means, rows * d * sizeof(R), hipMemcpyHostToDevice));
  const bool with_cov = covs && out_cov;
  if (with_cov) {
    if ((rc = dP.alloc(rows * d * d * sizeof(R))) || (rc = oP.alloc(rows * m * m * sizeof(R)))) return rc;
    CDKF_HIP_CHECK(hipMemcpy(dP.p, covs, rows * d * d * sizeof(R), hipMemcpyHostToDevice));
  }
  rc = emission_moments_dev<R>(mdl, rows, (const R*)dm.p, with_cov ? (const R*)dP.p : nullptr, (R*)om.p,
                               with_cov ? (R*)oP.p : nullptr, nullptr);
  if (rc) return rc;
  CDKF_HIP_CHECK(hipMemcpy(out_mean, om.p, rows * m * sizeof(R), hipMemcpyDeviceToHost));
  if (with_cov) CDKF_HIP_CHECK(hipMemcpy(out_cov, oP.p, rows * m * m * sizeof(R), hipMemcpyDeviceToHost));
  return CDKF_OK;
}

// (launch_custom.hip)
template <typename R>
int launch_custom_emission_moments(const cdkf_model* mdl, const cdkf_opts* o, int ukf, int64_t rows, const R* t, const R* u, const R* means,
                                   const R* covs, R* out_mean, R* out_cov, hipStream_t stream);

// host buffers in, host buffers out (cdkf_custom_emission_moments_*)
template <typename R>
int custom_emission_moments_host(const cdkf_model* mdl, const cdkf_opts* o, int ukf, int64_t rows, const R* t, const R* u, const R* means,
                                 const R* covs, R* out_mean, R* out_cov) {
  if (!mdl || !o || rows < 0 || (rows > 0 && (!means || !out_mean))) {
    set_error("custom_emission_moments: bad arguments");
    return CDKF_EINVAL;
  }
  if (!mdl->emission_kind) {  // (before anything touches the device: the refusal does not need one)
    set_error("custom_emission_moments: the model's emission is linear (cdkf_emission_moments_* serves it)");
    return CDKF_EINVAL;
  }
  const size_t d = mdl->state_dim, m = mdl->emission_dim, du = mdl->input_dim > 0 ? mdl->input_dim : 0;
  if (du > 0 && !u && rows > 0) {
    set_error("custom_emission_moments: the model has input_dim %d but no inputs were given", mdl->input_dim);
    return CDKF_EINVAL;
  }
  const bool with_cov = covs && out_cov;
  DevBuf dm, dP, dt, dn, om, oP;
  int rc;
  const size_t n = rows > 0 ? (size_t)rows : 0;
  if ((rc = dm.alloc(n * d * sizeof(R))) || (rc = om.alloc(n * m * sizeof(R)))) return rc;
  if (n) CDKF_HIP_CHECK(hipMemcpy(dm.p, means, n * d * sizeof(R), hipMemcpyHostToDevice));
  if (with_cov) {
    if ((rc = dP.alloc(n * d * d * sizeof(R))) || (rc = oP.alloc(n * m * m * sizeof(R)))) return rc;
    if (n) CDKF_HIP_CHECK(hipMemcpy(dP.p, covs, n * d * d * sizeof(R), hipMemcpyHostToDevice));
  }
  if (t) {
    if ((rc = dt.alloc(n * sizeof(R)))) return rc;
    if (n) CDKF_HIP_CHECK(hipMemcpy(dt.p, t, n * sizeof(R), hipMemcpyHostToDevice));
  }
  if (du > 0) {
    if ((rc = dn.alloc(n * du * sizeof(R)))) return rc;
    if (n) CDKF_HIP_CHECK(hipMemcpy(dn.p, u, n * du * sizeof(R), hipMemcpyHostToDevice));
  }
  rc = launch_custom_emission_moments<R>(mdl, o, ukf, rows, t ? (const R*)dt.p : nullptr, du > 0 ? (const R*)dn.p : nullptr, (const R*)dm.p,
                                         with_cov ? (const R*)dP.p : nullptr, (R*)om.p, with_cov ? (R*)oP.p : nullptr, nullptr);
  if (rc) return rc;
  CDKF_HIP_CHECK(hipStreamSynchronize(nullptr));
  if (n) CDKF_HIP_CHECK(hipMemcpy(out_mean, om.p, n * m * sizeof(R), hipMemcpyDeviceToHost));
  if (with_cov && n) CDKF_HIP_CHECK(hipMemcpy(out_cov, oP.p, n * m * m * sizeof(R), hipMemcpyDeviceToHost));
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
  o->solver = CDKF_SOLVER_DOPRI5;
  o->adaptive = 0;
  o->rtol = 1e-3;
  o->atol = 1e-6;
  o->pid_p = 0.0;
  o->pid_i = 1.0;
  o->pid_d = 0.0;
  o->layout_in = CDKF_LAYOUT_SAME;
  o->flags = 0;
  o->dtmin = 0.0;
  o->dtmax = HUGE_VAL;
  o->pid_safety = 0.9;
  o->pid_factormin = 0.2;
  o->pid_factormax = 10.0;
  o->inputs = nullptr;
  o->max_steps = 100000;
  o->dt0 = 0.01;
  o->dt_final = 1e-10;
  o->cov_rescaling = 1.0;
  o->ukf_alpha = std::sqrt(3.0);
  o->ukf_beta = 2.0;
  o->ukf_kappa = 1.0;
}

int cdkf_version(void) { return CDKF_VERSION; }
void cdkf_struct_sizes(int64_t* model_bytes, int64_t* opts_bytes) {
  if (model_bytes) *model_bytes = (int64_t)sizeof(cdkf_model);
  if (opts_bytes) *opts_bytes = (int64_t)sizeof(cdkf_opts);
}
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

int cdkf_trajectories_per_wavefront(int64_t N) { return cdkf::reg_lanes_per_wave(N < 1 ? 1 : N); }

int cdkf_preferred_layout(const cdkf_model* mdl) {
  // (drifts given as source: the register-resident kernels up to six dimensions, the workgroup kernels beyond)
  return (mdl && (reg_shape_available(mdl) || (custom_kind(mdl->drift_kind) && mdl->state_dim <= 6 && mdl->emission_dim <= 6))) ? CDKF_LAYOUT_TCN
                                                                                                                              : CDKF_LAYOUT_TN;
}

int cdkf_custom_drift_register(int state_dim, int n_theta, const char* f_src, const char* jac_src, const char* divgrad_src) {
  return custom_register(state_dim, n_theta, f_src, jac_src, divgrad_src);
}
int cdkf_custom_drift_compile(int drift_kind, int bytes_per_real, int emission_dim, int algo, int state_order,
                              int emission_kind) {
  return custom_compile_check(drift_kind, bytes_per_real, emission_dim, algo, state_order, emission_kind);
}
int cdkf_custom_emission_register(int state_dim, int emission_dim, const char* h_src, const char* hjac_src) {
  return custom_emission_register(state_dim, emission_dim, h_src, hjac_src);
}
void cdkf_set_kernel_source_dir(const char* dir) { custom_set_source_dir(dir); }
void cdkf_rtc_cache_stats(int64_t* hits, int64_t* misses) { custom_rtc_cache_stats(hits, misses); }
int cdkf_debug_custom_reg_blob(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, int algo, int bytes_per_real,
                               void* par_out, int64_t par_cap_bytes, int64_t* ip_out) {
  return custom_debug_reg_blob(mdl, opts, N, T, algo, bytes_per_real, par_out, par_cap_bytes, ip_out);
}
int cdkf_ukf_tangent_compile(const cdkf_model* mdl, const cdkf_opts* opts, int bytes_per_real) {
  return ukf_tangent_compile_check(mdl, opts, bytes_per_real, 0);
}
int cdkf_ekf_tangent_compile(const cdkf_model* mdl, const cdkf_opts* opts, int bytes_per_real) {
  return ukf_tangent_compile_check(mdl, opts, bytes_per_real, 1);
}
int cdkf_custom_emission_moments_compile(const cdkf_model* mdl, const cdkf_opts* opts, int bytes_per_real) {
  if (!mdl || !opts) return CDKF_EINVAL;
  cdkf_opts oo = *opts;  // (the integrator's settings do not matter to this kernel)
  oo.solver = CDKF_SOLVER_DOPRI5;
  oo.adaptive = 0;
  oo.forecast = 0;
  return ukf_tangent_compile_check(mdl, &oo, bytes_per_real, 2);
}
int cdkf_custom_emission_moments_f64(const cdkf_model* mdl, const cdkf_opts* opts, int ukf, int64_t rows, const double* t, const double* inputs,
                                     const double* means, const double* covs, double* out_mean, double* out_cov) {
  return custom_emission_moments_host<double>(mdl, opts, ukf, rows, t, inputs, means, covs, out_mean, out_cov);
}
int cdkf_custom_emission_moments_f32(const cdkf_model* mdl, const cdkf_opts* opts, int ukf, int64_t rows, const float* t, const float* inputs,
                                     const float* means, const float* covs, float* out_mean, float* out_cov) {
  return custom_emission_moments_host<float>(mdl, opts, ukf, rows, t, inputs, means, covs, out_mean, out_cov);
}
int cdkf_custom_emission_moments_f64_dev(const cdkf_model* mdl, const cdkf_opts* opts, int ukf, int64_t rows, const double* t, const double* inputs,
                                         const double* means, const double* covs, double* out_mean, double* out_cov, void* stream) {
  return launch_custom_emission_moments<double>(mdl, opts, ukf, rows, t, inputs, means, covs, out_mean, out_cov, (hipStream_t)stream);
}
int cdkf_custom_emission_moments_f32_dev(const cdkf_model* mdl, const cdkf_opts* opts, int ukf, int64_t rows, const float* t, const float* inputs,
                                         const float* means, const float* covs, float* out_mean, float* out_cov, void* stream) {
  return launch_custom_emission_moments<float>(mdl, opts, ukf, rows, t, inputs, means, covs, out_mean, out_cov, (hipStream_t)stream);
}
int cdkf_debug_ukf_tangent_args(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, int bytes_per_real, int all,
                                void* args_out, int64_t args_cap_bytes, void* par_out, int64_t par_cap_bytes) {
  return ukf_tangent_debug_args(mdl, opts, N, T, bytes_per_real, all, args_out, args_cap_bytes, par_out, par_cap_bytes);
}
int cdkf_debug_exec_prologue_check(const void* code, int64_t bytes, const char* arch, char* where, int64_t where_cap) {
  if (!code || bytes <= 0 || !arch) return -1;
  std::vector<char> v((const char*)code, (const char*)code + bytes);
  std::string w;
  const int rc = rtc_exec_prologue_check(v, arch, &w);
  if (where && where_cap > 0) snprintf(where, (size_t)where_cap, "%s", w.c_str());
  return rc;
}
int cdkf_debug_wg_args(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, int bytes_per_real, int ukf, int smoother,
                       void* args_out, int64_t args_cap_bytes, void* blob_out, int64_t blob_cap_bytes, int64_t* geom_out) {
  return debug_wg_args(mdl, opts, N, T, bytes_per_real, ukf, smoother, args_out, args_cap_bytes, blob_out, blob_cap_bytes, geom_out);
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
const char* cdkf_last_kernel(void) { return g_kernel; }
int cdkf_event_create(void** event) {
  if (!event) {
    set_error("cdkf_event_create: NULL argument");
    return CDKF_EINVAL;
  }
  hipEvent_t e = nullptr;
  CDKF_HIP_CHECK(hipEventCreate(&e));
  *event = e;
  return CDKF_OK;
}
int cdkf_event_record(void* event, void* stream) {
  CDKF_HIP_CHECK(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
  return CDKF_OK;
}
int cdkf_event_elapsed_ms(void* start, void* stop, float* ms) {
  if (!ms) {
    set_error("cdkf_event_elapsed_ms: NULL argument");
    return CDKF_EINVAL;
  }
  CDKF_HIP_CHECK(hipEventSynchronize((hipEvent_t)stop));
  CDKF_HIP_CHECK(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return CDKF_OK;
}
int cdkf_event_destroy(void* event) {
  if (event) CDKF_HIP_CHECK(hipEventDestroy((hipEvent_t)event));
  return CDKF_OK;
}
int cdkf_stream_create(void** stream) {
  if (!stream) {
    set_error("cdkf_stream_create: NULL argument");
    return CDKF_EINVAL;
  }
  hipStream_t s = nullptr;
  CDKF_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = s;
  return CDKF_OK;
}
int cdkf_stream_destroy(void* stream) {
  if (stream) CDKF_HIP_CHECK(hipStreamDestroy((hipStream_t)stream));
  return CDKF_OK;
}
int cdkf_set_device(int device) {
  CDKF_HIP_CHECK(hipSetDevice(device));
  return CDKF_OK;
}

#define CDKF_DEFINE_ALGO(NAME, SUFFIX, RTYPE, LAUNCH)                                                              \
  int cdkf_##NAME##_##SUFFIX##_dev(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const RTYPE* t, \
                                   const RTYPE* y, RTYPE* ll, RTYPE* a1, RTYPE* a2, RTYPE* a3, RTYPE* a4,          \
                                   int32_t* status, void* stream) {                                                \
    int rc = check_common(mdl, o, N, T, t, y, ll);                                                                 \
    if (rc) return rc;                                                                                             \
    if (N == 0) return CDKF_OK;                                                                                    \
    CDKF_SELECT_DEVICE(o);                                                                        \
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

int cdkf_emission_moments_f64(const cdkf_model* mdl, int64_t rows, const double* means, const double* covs, double* om,
                              double* oc) {
  return emission_moments_host<double>(mdl, rows, means, covs, om, oc);
}
int cdkf_emission_moments_f32(const cdkf_model* mdl, int64_t rows, const float* means, const float* covs, float* om,
                              float* oc) {
  return emission_moments_host<float>(mdl, rows, means, covs, om, oc);
}
int cdkf_emission_moments_f64_dev(const cdkf_model* mdl, int64_t rows, const double* means, const double* covs,
                                  double* om, double* oc, void* stream) {
  return emission_moments_dev<double>(mdl, rows, means, covs, om, oc, (hipStream_t)stream);
}
int cdkf_emission_moments_f32_dev(const cdkf_model* mdl, int64_t rows, const float* means, const float* covs, float* om,
                                  float* oc, void* stream) {
  return emission_moments_dev<float>(mdl, rows, means, covs, om, oc, (hipStream_t)stream);
}

int cdkf_ekf_loglik_grad_f64(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const double* t,
                             const double* y, double* ll, double* grad, int32_t* status) {
  return loglik_grad_host<double>(mdl, o, N, T, t, y, ll, grad, nullptr, status);
}
int cdkf_ekf_loglik_grad_f32(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const float* t,
                             const float* y, float* ll, float* grad, int32_t* status) {
  return loglik_grad_host<float>(mdl, o, N, T, t, y, ll, grad, nullptr, status);
}
int cdkf_ekf_loglik_grad_f64_dev(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const double* t,
                                 const double* y, double* ll, double* grad, int32_t* status, void* stream) {
  return loglik_grad_dev<double>(mdl, o, N, T, t, y, ll, grad, nullptr, status, stream);
}
int cdkf_ekf_loglik_grad_f32_dev(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const float* t,
                                 const float* y, float* ll, float* grad, int32_t* status, void* stream) {
  return loglik_grad_dev<float>(mdl, o, N, T, t, y, ll, grad, nullptr, status, stream);
}
int cdkf_ukf_loglik_grad_f64(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const double* t,
                             const double* y, double* ll, double* grad, int32_t* status) {
  return loglik_grad_host<double>(mdl, o, N, T, t, y, ll, grad, nullptr, status, true);
}
int cdkf_ukf_loglik_grad_f32(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const float* t,
                             const float* y, float* ll, float* grad, int32_t* status) {
  return loglik_grad_host<float>(mdl, o, N, T, t, y, ll, grad, nullptr, status, true);
}
int cdkf_ukf_loglik_grad_f64_dev(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const double* t,
                                 const double* y, double* ll, double* grad, int32_t* status, void* stream) {
  return loglik_grad_dev<double>(mdl, o, N, T, t, y, ll, grad, nullptr, status, stream, true);
}
int cdkf_ukf_loglik_grad_f32_dev(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const float* t,
                                 const float* y, float* ll, float* grad, int32_t* status, void* stream) {
  return loglik_grad_dev<float>(mdl, o, N, T, t, y, ll, grad, nullptr, status, stream, true);
}
int cdkf_ukf_grad_supported(const cdkf_model* mdl, const cdkf_opts* o) { return (mdl && o && ukf_grad_shape_available(mdl, o)) ? 1 : 0; }
int cdkf_kf_smoother1_f64(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const double* t, const double* y,
                          double* ll, double* fm, double* fP, double* sm, double* sP, double* cross, int32_t* status) {
  return kf_smoother1_host<double>(mdl, o, N, T, t, y, ll, fm, fP, sm, sP, cross, status);
}
int cdkf_kf_smoother1_f32(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const float* t, const float* y,
                          float* ll, float* fm, float* fP, float* sm, float* sP, float* cross, int32_t* status) {
  return kf_smoother1_host<float>(mdl, o, N, T, t, y, ll, fm, fP, sm, sP, cross, status);
}
int cdkf_kf_smoother1_f64_dev(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const double* t,
                              const double* y, double* ll, double* fm, double* fP, double* sm, double* sP, double* cross,
                              int32_t* status, void* stream) {
  return kf_smoother1_dev<double>(mdl, o, N, T, t, y, ll, fm, fP, sm, sP, cross, status, stream);
}
int cdkf_kf_smoother1_f32_dev(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const float* t,
                              const float* y, float* ll, float* fm, float* fP, float* sm, float* sP, float* cross,
                              int32_t* status, void* stream) {
  return kf_smoother1_dev<float>(mdl, o, N, T, t, y, ll, fm, fP, sm, sP, cross, status, stream);
}
int cdkf_kf_smoother1_supported(const cdkf_model* mdl) { return (mdl && smoother1_shape_available(mdl)) ? 1 : 0; }
int cdkf_kf_pushforward_f64(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const double* t, double* AQ) {
  return kf_pushforward_host<double>(mdl, o, N, T, t, AQ);
}
int cdkf_kf_pushforward_f32(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const float* t, float* AQ) {
  return kf_pushforward_host<float>(mdl, o, N, T, t, AQ);
}

static int need_model_grad(const void* gm) {
  if (gm) return CDKF_OK;
  set_error("loglik_grad_all: grad_model must not be NULL");
  return CDKF_EINVAL;
}
int cdkf_ekf_loglik_grad_all_f64(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const double* t,
                                 const double* y, double* ll, double* grad, double* grad_model, int32_t* status) {
  if (int rc = need_model_grad(grad_model)) return rc;
  return loglik_grad_host<double>(mdl, o, N, T, t, y, ll, grad, grad_model, status);
}
int cdkf_ekf_loglik_grad_all_f32(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const float* t,
                                 const float* y, float* ll, float* grad, float* grad_model, int32_t* status) {
  if (int rc = need_model_grad(grad_model)) return rc;
  return loglik_grad_host<float>(mdl, o, N, T, t, y, ll, grad, grad_model, status);
}
int cdkf_ekf_loglik_grad_all_f64_dev(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const double* t,
                                     const double* y, double* ll, double* grad, double* grad_model, int32_t* status,
                                     void* stream) {
  if (int rc = need_model_grad(grad_model)) return rc;
  return loglik_grad_dev<double>(mdl, o, N, T, t, y, ll, grad, grad_model, status, stream);
}
int cdkf_ekf_loglik_grad_all_f32_dev(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const float* t,
                                     const float* y, float* ll, float* grad, float* grad_model, int32_t* status,
                                     void* stream) {
  if (int rc = need_model_grad(grad_model)) return rc;
  return loglik_grad_dev<float>(mdl, o, N, T, t, y, ll, grad, grad_model, status, stream);
}
int cdkf_ukf_loglik_grad_all_f64(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const double* t, const double* y,
                                 double* ll, double* grad, double* grad_model, int32_t* status) {
  if (int rc = need_model_grad(grad_model)) return rc;
  return loglik_grad_host<double>(mdl, o, N, T, t, y, ll, grad, grad_model, status, true);
}
int cdkf_ukf_loglik_grad_all_f32(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const float* t, const float* y,
                                 float* ll, float* grad, float* grad_model, int32_t* status) {
  if (int rc = need_model_grad(grad_model)) return rc;
  return loglik_grad_host<float>(mdl, o, N, T, t, y, ll, grad, grad_model, status, true);
}
int cdkf_ukf_loglik_grad_all_f64_dev(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const double* t, const double* y,
                                     double* ll, double* grad, double* grad_model, int32_t* status, void* stream) {
  if (int rc = need_model_grad(grad_model)) return rc;
  return loglik_grad_dev<double>(mdl, o, N, T, t, y, ll, grad, grad_model, status, stream, true);
}
int cdkf_ukf_loglik_grad_all_f32_dev(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const float* t, const float* y,
                                     float* ll, float* grad, float* grad_model, int32_t* status, void* stream) {
  if (int rc = need_model_grad(grad_model)) return rc;
  return loglik_grad_dev<float>(mdl, o, N, T, t, y, ll, grad, grad_model, status, stream, true);
}
int cdkf_ukf_grad_all_supported(const cdkf_model* mdl, const cdkf_opts* o) {
  return (mdl && o && ukf_grad_all_shape_available(mdl, o)) ? 1 : 0;
}
int cdkf_ekf_loglik_grad_jumps_f64(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const double* t, const double* y,
                                   const double* jumps, double* ll, double* grad, double* grad_model, double* grad_jumps, double* grad_y,
                                   int32_t* status) {
  return loglik_grad_jumps_host<double>(mdl, o, N, T, t, y, jumps, ll, grad, grad_model, grad_jumps, grad_y, status);
}
int cdkf_ekf_loglik_grad_jumps_f32(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const float* t, const float* y,
                                   const float* jumps, float* ll, float* grad, float* grad_model, float* grad_jumps, float* grad_y,
                                   int32_t* status) {
  return loglik_grad_jumps_host<float>(mdl, o, N, T, t, y, jumps, ll, grad, grad_model, grad_jumps, grad_y, status);
}
int cdkf_release_workspace(void) { return release_grad_workspace(); }
int cdkf_grad_all_supported(const cdkf_model* mdl, const cdkf_opts* o) {
  return (mdl && o && (adjoint_shape_available(mdl, o) || ekf_tangent_available(mdl, o))) ? 1 : 0;
}
int cdkf_grad_supported(const cdkf_model* mdl, const cdkf_opts* o) {
  return (mdl && o && (grad_shape_available(mdl, o) || (mdl->n_theta >= 1 && ekf_tangent_available(mdl, o)))) ? 1 : 0;
}
int cdkf_grad_sum_f64_dev(const double* grad, int64_t N, int64_t n_theta, double* out, void* stream) {
  return grad_sum_dev<double>(grad, N, n_theta, out, stream);
}
int cdkf_grad_sum_f32_dev(const float* grad, int64_t N, int64_t n_theta, double* out, void* stream) {
  return grad_sum_dev<float>(grad, N, n_theta, out, stream);
}

int cdkf_ll_sum_f64_dev(const double* ll, int64_t N, double* out, void* stream) {
  return ll_sum_dev<double>(ll, N, out, stream);
}
int cdkf_ll_sum_f32_dev(const float* ll, int64_t N, double* out, void* stream) {
  return ll_sum_dev<float>(ll, N, out, stream);
}

}  // extern "C"
