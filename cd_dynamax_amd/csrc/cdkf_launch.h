// cdkf_launch.h -- kernel selection.  The register-resident ("reg") kernels are instantiated for the
// (drift, state_dim, emission_dim) shapes listed in CDKF_REG_SHAPES; every other shape goes to the
// LDS-resident wave-per-trajectory kernels (cdkf_wave_kernels.h).
#pragma once
#include <cstdlib>
#include <mutex>
#include <vector>
#include "cdkf_host.h"

// X(drift_kind, DriftTemplate, D, M)
#define CDKF_REG_SHAPES(X)                    \
  X(CDKF_DRIFT_LORENZ63, DriftLorenz63, 3, 1) \
  X(CDKF_DRIFT_LORENZ63, DriftLorenz63, 3, 2) \
  X(CDKF_DRIFT_LORENZ63, DriftLorenz63, 3, 3) \
  X(CDKF_DRIFT_LINEAR, DriftLinear, 1, 1)     \
  X(CDKF_DRIFT_LINEAR, DriftLinear, 2, 1)     \
  X(CDKF_DRIFT_LINEAR, DriftLinear, 2, 2)     \
  X(CDKF_DRIFT_LINEAR, DriftLinear, 2, 6)     \
  X(CDKF_DRIFT_LINEAR, DriftLinear, 3, 1)     \
  X(CDKF_DRIFT_LINEAR, DriftLinear, 3, 3)     \
  X(CDKF_DRIFT_LINEAR, DriftLinear, 4, 2)     \
  X(CDKF_DRIFT_LINEAR, DriftLinear, 4, 4)

namespace cdkf {

template <typename R>
int launch_ekf_filter(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll,
                      R* fm, R* fP, R* pm, R* pP, int32_t* status, hipStream_t stream);
template <typename R>
int launch_ukf_filter(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll,
                      R* fm, R* fP, R* pm, R* pP, int32_t* status, hipStream_t stream);
template <typename R>
int launch_ekf_smoother(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y,
                        R* ll, R* fm, R* fP, R* sm, R* sP, int32_t* status, hipStream_t stream);

// log-likelihood + gradient w.r.t. the drift parameters (launch_grad.hip); grad [N, n_theta]
template <typename R>
int launch_ekf_grad(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll,
                    R* grad, R* grad_model, int32_t* status, hipStream_t stream);
bool grad_shape_available(const cdkf_model* mdl, const cdkf_opts* o);
// the unscented filter's log-likelihood + gradient w.r.t. the drift parameters (launch_grad.hip)
template <typename R>
int launch_ukf_grad(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll, R* grad,
                    int32_t* status, hipStream_t stream);
bool ukf_grad_shape_available(const cdkf_model* mdl, const cdkf_opts* o);
// reverse sweep, state_dim <= 8 (launch_wg.hip, cdkf_adjoint_kernels.h); grad_model (optional): [N, d + 2 d^2 + m d + m + m^2]
template <typename R>
int launch_ekf_grad_adjoint(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll,
                            R* grad, R* grad_model, int32_t* status, hipStream_t stream);
// ... with per-step jumps of the predicted mean ([N, T, d]) and the per-step cotangents of the jumps and of the observations
// ([N, T, d], [N, T, m]; cdkf_ekf_loglik_grad_jumps_*): state_dim, emission_dim <= 8, fixed-step Dormand-Prince
template <typename R>
int launch_ekf_grad_adjoint_jumps(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y,
                                  const R* jumps, R* ll, R* grad, R* grad_model, R* grad_jumps, R* grad_y, int32_t* status,
                                  hipStream_t stream);
bool adjoint_shape_available(const cdkf_model* mdl, const cdkf_opts* o);
// the unscented filter's reverse-sweep gradient, every leaf (launch_wg.hip): Lorenz-63 / Lorenz-96 / linear drifts, linear emission
bool ukf_grad_all_shape_available(const cdkf_model* mdl, const cdkf_opts* o);
template <typename R>
int launch_ukf_grad_all(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll, R* grad,
                        R* grad_model, int32_t* status, hipStream_t stream);
int release_grad_workspace();  // frees the per-process reverse-sweep workspace once its last user has finished (launch_wg.hip)
// lease of the per-process reverse-sweep workspace (launch_wg.hip): holds its lock from construction to destruction
struct GradWorkspaceLease {
  GradWorkspaceLease();
  ~GradWorkspaceLease();
  GradWorkspaceLease(const GradWorkspaceLease&) = delete;
  GradWorkspaceLease& operator=(const GradWorkspaceLease&) = delete;
  int reserve(size_t bytes, hipStream_t stream, void** p);  // grow-only; waits for an earlier launch that still uses it
  int done(hipStream_t stream);                             // behind the last kernel that reads the workspace
};

// linear model, smoother type 1 (launch_wg.hip, cdkf_rts1_kernels.h); cross (optional) has the strides of sP, entries 0..T-2
template <typename R>
int launch_kf_smoother1(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll,
                        R* fm, R* fP, R* sm, R* sP, R* cross, int32_t* status, hipStream_t stream);
bool smoother1_shape_available(const cdkf_model* mdl);
// the pushed-forward (A, Q) of every interval, AQ [N, T-1, 2, d, d] (same kernel, same shapes as the type-1 smoother)
template <typename R>
int launch_kf_pushforward(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, R* AQ, int32_t* status,
                          hipStream_t stream);

// ring of persistent parameter buffers (launch_wg.hip): device block + pinned staging + an event behind the readers
struct ParamSlot {
  void* dev = nullptr;
  void* host = nullptr;
  size_t cap = 0;
  int device = -1;
  hipEvent_t done = nullptr;
  bool in_flight = false;
};
int param_pool_acquire(size_t bytes, ParamSlot** out);
int param_pool_release(ParamSlot* s, hipStream_t stream);
// A slot on loan: whatever way the launcher leaves, the slot's event is recorded behind the work enqueued so far, so the ring
// cannot hand the pinned staging buffer to the next call while this call's copy is still pending.
struct ParamLease {
  ParamSlot* slot = nullptr;
  hipStream_t stream = nullptr;
  explicit ParamLease(hipStream_t s) : stream(s) {}
  ParamLease(const ParamLease&) = delete;
  ParamLease& operator=(const ParamLease&) = delete;
  int release() {
    ParamSlot* s = slot;
    slot = nullptr;
    return s ? param_pool_release(s, stream) : CDKF_OK;
  }
  ~ParamLease() { (void)release(); }
};

// user-supplied drifts compiled at run time (launch_custom.hip); algo: 0 EKF filter, 1 UKF filter, 2 EKF smoother
bool custom_kind(int kind);
bool custom_shape_available(const cdkf_model* mdl, const cdkf_opts* o);
bool custom_grad_available(const cdkf_model* mdl, const cdkf_opts* o);  // algo 3: log-likelihood + gradient w.r.t. theta (a1: grad [N, n_theta])
template <typename R>
int launch_custom(int algo, const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll,
                  R* a1, R* a2, R* a3, R* a4, int32_t* status, hipStream_t stream);
int custom_register(int state_dim, int n_theta, const char* f_src, const char* jac_src, const char* divgrad_src);
int custom_compile_check(int kind, int bytes_per_real, int emission_dim, int algo, int state_order, int emission_kind);
// the unscented filter's gradient for any drift / emission: forward mode through the literal sigma-point recursion (launch_custom.hip,
// cdkf_ukf_tangent_kernels.h); grad_model null: the drift parameters only
bool ukf_tangent_available(const cdkf_model* mdl, const cdkf_opts* o);
template <typename R>
int launch_ukf_tangent(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll, R* grad,
                       R* grad_model, int32_t* status, hipStream_t stream);
// ... and the EXTENDED filter's on the same plan (ekf_tangent_body: jacfwd by an outer dual level; any num_iter, emissions given as source)
bool ekf_tangent_available(const cdkf_model* mdl, const cdkf_opts* o);
template <typename R>
int launch_ekf_tangent(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll, R* grad,
                       R* grad_model, int32_t* status, hipStream_t stream);
int ukf_tangent_compile_check(const cdkf_model* mdl, const cdkf_opts* o, int bytes_per_real, int ekf);
// 1: the code object shows the ROCm 7.2 spill-placement defect (vector spill code in front of an execution-mask restore), 0: clean, -1: could not look
int rtc_exec_prologue_check(const std::vector<char>& code, const std::string& arch, std::string* where);
int ukf_tangent_debug_args(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, int bytes_per_real, int all, void* args_out,
                           int64_t args_cap, void* par_out, int64_t par_cap);
void custom_rtc_cache_stats(int64_t* hits, int64_t* misses);
int custom_debug_reg_blob(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, int algo, int bytes_per_real, void* par_out,
                          int64_t par_cap_bytes, int64_t* ip_out);
int custom_emission_register(int state_dim, int emission_dim, const char* h_src, const char* hjac_src);
bool custom_emission_kind(int emission_kind, int d, int m);
void custom_set_source_dir(const char* dir);
// ... beyond six state / emission dimensions: on the workgroup kernels (launch_wg.hip), which are compiled with the drift at run time
long custom_ntheta(int kind, int state_dim);
bool custom_wg_fits(const cdkf_model* mdl);  // launch_wg.hip: the workgroup kernels' LDS plan holds this shape (in fp32)
int custom_wg_geometry(int kind, int d, int m, int bytes_per_real, bool ukf, int* ept, int* threads, size_t* lds_f, size_t* lds_s);
template <typename R>
struct WgArgs;
template <typename R>
int launch_custom_wg(const WgArgs<R>& a, int ept, bool filter, bool smoother, int threads, size_t lds_f, size_t lds_s, hipStream_t stream);
// ... and the reverse sweep (launch_adjwg.hip)
bool custom_adjoint_available(const cdkf_model* mdl, const cdkf_opts* o);
int custom_awg_geometry(int d, int m, int bytes_per_real, int* ne, size_t* lds);  // launch_adjwg.hip; nonzero: does not fit
template <typename R>
int launch_custom_awg(const WgArgs<R>& a, R* grad, R* grad_model, R* scratch, long scratch_stride, int cap, int ne, size_t lds, hipStream_t stream);

bool kernel_available(const cdkf_model* mdl, const cdkf_opts* o, int algo, int bytes_per_real);

// workgroup-per-trajectory kernels (launch_wg.hip): any registry drift, d and m up to what fits 160 KB of LDS
bool wg_shape_available(const cdkf_model* mdl, int bytes_per_real);
int debug_wg_args(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, int bytes_per_real, int ukf, int smoother, void* args_out,
                  int64_t args_cap, void* blob_out, int64_t blob_cap, int64_t* geom);
template <typename R>
int launch_ekf_filter_wg(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll,
                         R* fm, R* fP, R* pm, R* pP, int32_t* status, hipStream_t stream);
template <typename R>
int launch_ukf_filter_wg(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll,
                         R* fm, R* fP, R* pm, R* pP, int32_t* status, hipStream_t stream);
template <typename R>
int launch_ekf_smoother_wg(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y,
                           R* ll, R* fm, R* fP, R* sm, R* sP, int32_t* status, hipStream_t stream);

// true if the emission picks the first M state coordinates: H = I[:M], bias = 0
inline bool emission_is_selection(const cdkf_model* mdl) {
  const int d = mdl->state_dim, m = mdl->emission_dim;
  if (m > d) return false;
  for (int r = 0; r < m; ++r) {
    if (mdl->h_bias[r] != 0.0) return false;
    for (int j = 0; j < d; ++j)
      if (mdl->H[r * d + j] != (r == j ? 1.0 : 0.0)) return false;
  }
  return true;
}

// launch filter_reg_kernel with the OUT specialisation matching the requested output pointers
template <typename R, int D, int M, typename Drift, bool UKF, bool ZEROTH, bool HSEL>
inline void launch_filter_reg(const RegArgs<R, D, M, Drift>& a, hipStream_t stream) {
  const dim3 grid(reg_grouping(a.N, (int)sizeof(R)).blocks), block(64);
  const bool all = a.fm && a.fP && a.pm && a.pP, none = !a.fm && !a.fP && !a.pm && !a.pP;
  note_kernel("filter_reg_kernel<%s, %d, %d, ", real_name<R>(), D, M);
  if (a.rk.stages && (a.solver != CDKF_SOLVER_DOPRI5 || a.rk.adaptive)) {  // non-default method or adaptive steps: run-time tableau
    if (a.forecast)
      hipLaunchKernelGGL((filter_reg_kernel<R, D, M, Drift, UKF, ZEROTH, false, kOutSome, true, true>), grid, block, 0, stream, a);
    else
      hipLaunchKernelGGL((filter_reg_kernel<R, D, M, Drift, UKF, ZEROTH, false, kOutSome, false, true>), grid, block, 0, stream, a);
    return;
  }
  if (a.forecast)  // rare path: one generic instantiation (run-time output checks, general emission)
    hipLaunchKernelGGL((filter_reg_kernel<R, D, M, Drift, UKF, ZEROTH, false, kOutSome, true>), grid, block, 0, stream, a);
  else if (all)
    hipLaunchKernelGGL((filter_reg_kernel<R, D, M, Drift, UKF, ZEROTH, HSEL, kOutAll>), grid, block, 0, stream, a);
  else if (none)
    hipLaunchKernelGGL((filter_reg_kernel<R, D, M, Drift, UKF, ZEROTH, HSEL, kOutNone>), grid, block, 0, stream, a);
  else
    hipLaunchKernelGGL((filter_reg_kernel<R, D, M, Drift, UKF, ZEROTH, HSEL, kOutSome>), grid, block, 0, stream, a);
}

inline bool reg_shape_available(const cdkf_model* mdl) {
#define X(KIND, DRIFT, D_, M_) \
  if (mdl->drift_kind == KIND && mdl->state_dim == D_ && mdl->emission_dim == M_) return true;
  CDKF_REG_SHAPES(X)
#undef X
  return false;
}

// ---- shared by the translation units that launch LDS-heavy kernels (launch_wg.hip, launch_w40.hip) ------------------------
static constexpr size_t kLdsLimit = 160 * 1024;

// Raise the dynamic-LDS cap of a kernel ONCE to the whole CU (minus the kernels' few static bytes).  Re-setting the
// attribute to the exact size before every launch misbehaved on ROCm 7.2: the first launch after the size grew
// ran with the stale, smaller cap and produced garbage.
template <typename K>
static int wg_raise_lds_cap(K kernel) {
  CDKF_HIP_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kLdsLimit - 256)));
  return CDKF_OK;
}
// The attribute belongs to the (kernel, device) pair: raise it once per device the process launches on, not once per process.
// F is a distinct lambda type per call site and template instantiation, so each has its own record.
template <typename F>
static int once_per_device(F&& raise) {
  static std::mutex m;
  static std::vector<char> done;
  int dev = 0;
  CDKF_HIP_CHECK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(m);
  if ((size_t)dev >= done.size()) done.resize(dev + 1, 0);
  if (!done[dev]) {
    if (raise()) return CDKF_EHIP;
    done[dev] = 1;
  }
  return CDKF_OK;
}

// Lorenz-96 with H = I at state_dim 40 (BASELINE config 4): wavefront-per-trajectory sweeps (launch_w40.hip,
// cdkf_wave40_kernels.h).  backward = false: the filter sweep; true: the smoother's backward sweep over the filtered moments.
template <typename R>
struct WgArgs;
bool wave40_shape(const cdkf_model* mdl, const cdkf_opts* o);
template <typename R>
int launch_wave40(const WgArgs<R>& a, hipStream_t stream, bool backward = false);
// filter sweep for state_dim <= 8, one wavefront per trajectory (launch_w8.hip, cdkf_wave8_kernels.h)
template <typename R>
int launch_wave8(const WgArgs<R>& a, hipStream_t stream);
// reverse sweep (gradient) / smoother backward sweep for state_dim <= 8 (launch_adj.hip, cdkf_adjoint_kernels.h)
template <typename R, bool MLP, bool SMOOTH = false>
int launch_adjoint_kernel(const WgArgs<R>& a, R* grad, R* grad_model, hipStream_t stream);
// reverse sweep (gradient) for larger states, one workgroup per trajectory (launch_adjwg.hip, cdkf_adjoint_wg_kernels.h); scratch:
// the Lorenz-96 reverse sweep on one wavefront per trajectory (cdkf_adjoint_w40_kernels.h, launch_adjw40.hip): the shapes wave40_shape
// admits; scratch: N * wave40_adjoint_scratch_reals(d, cap) reals
long wave40_adjoint_scratch_reals(int d, int cap);
template <typename R>
int launch_wave40_adjoint(const WgArgs<R>& a, R* grad, R* grad_model, R* scratch, int cap, hipStream_t stream);
// N * adjoint_wg_scratch_reals(d, cap) reals
bool adjoint_wg_fits(int d, int m, int bytes_per_real);
bool adjoint_wg_fits_mlp(int d, int m, int h1, int h2, int bytes_per_real);  // ... + the MLP drift's LDS region
long adjoint_wg_scratch_reals(int d, int cap);
template <typename R>
int launch_adjoint_wg_kernel(const WgArgs<R>& a, R* grad, R* grad_model, R* scratch, int cap, hipStream_t stream);

}  // namespace cdkf
