// launch_eks.hip -- EKF smoother: forward filter sweep (num_iter = 1) then the backward sweep.
#include "cdkf_launch.h"
#include "cdkf_lpe_kernels.h"

namespace cdkf {

template <typename R, int D, int M, typename Drift>
static int run_eks_reg(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll,
                       R* fm, R* fP, R* sm, R* sP, int32_t* status, hipStream_t stream) {
  cdkf_opts of = *o;
  of.num_iter = 1;  // the smoother's internal filter call uses the default (inference_ekf.py:489-495)
  RegArgs<R, D, M, Drift> a;
  fill_reg_args(a, mdl, &of, N, T, t, y, ll, fm, fP, (R*)nullptr, (R*)nullptr, status);
  const unsigned blocks = reg_grouping(N, (int)sizeof(R)).blocks;
  if (of.solver != CDKF_SOLVER_DOPRI5 || of.adaptive) {
    if (of.state_order == CDKF_ORDER_ZEROTH)
      hipLaunchKernelGGL((filter_reg_kernel<R, D, M, Drift, false, true, false, kOutSome, false, true>), dim3(blocks), dim3(64), 0, stream, a);
    else
      hipLaunchKernelGGL((filter_reg_kernel<R, D, M, Drift, false, false, false, kOutSome, false, true>), dim3(blocks), dim3(64), 0, stream, a);
    CDKF_HIP_CHECK(hipGetLastError());
    note_kernel("ekf_smoother_reg_kernel<%s, %d, %d, ", real_name<R>(), D, M);
    hipLaunchKernelGGL((ekf_smoother_reg_kernel<R, D, M, Drift, true>), dim3(blocks), dim3(64), 0, stream, a, sm, sP);
    CDKF_HIP_CHECK(hipGetLastError());
    return CDKF_OK;
  }
  // the smoother's internal filter runs with the caller's state_order (extended_kalman_smoother forwards hyperparams to the
  // filter, inference_ekf.py:489-495); the backward pass is always smooth_order 'first'
  if (try_lpe(a, mdl, &of, stream)) {  // small Lorenz-63 batch: sixteen lanes per trajectory, filtered moments only
  } else if (of.state_order == CDKF_ORDER_ZEROTH)
    hipLaunchKernelGGL((filter_reg_kernel<R, D, M, Drift, false, true, false, kOutSome>), dim3(blocks), dim3(64), 0, stream, a);
  else if (M <= D && emission_is_selection(mdl))  // H = I[:M], no bias: the products with H disappear (as in launch_ekf.hip)
    hipLaunchKernelGGL((filter_reg_kernel<R, D, M, Drift, false, false, (M <= D), kOutSome>), dim3(blocks), dim3(64), 0, stream, a);
  else
    hipLaunchKernelGGL((filter_reg_kernel<R, D, M, Drift, false, false, false, kOutSome>), dim3(blocks), dim3(64), 0, stream, a);
  CDKF_HIP_CHECK(hipGetLastError());
  if (!try_lpe_smoother(a, &of, sm, sP, stream)) {
    note_kernel("ekf_smoother_reg_kernel<%s, %d, %d, ", real_name<R>(), D, M);
    hipLaunchKernelGGL((ekf_smoother_reg_kernel<R, D, M, Drift>), dim3(blocks), dim3(64), 0, stream, a, sm, sP);
  }
  CDKF_HIP_CHECK(hipGetLastError());
  return CDKF_OK;
}

template <typename R>
int launch_ekf_smoother(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y,
                        R* ll, R* fm, R* fP, R* sm, R* sP, int32_t* status, hipStream_t stream) {
  if (!fm || !fP || !sm || !sP) {
    set_error("EKF smoother: filtered and smoothed output pointers must not be NULL");
    return CDKF_EINVAL;
  }
  if (custom_kind(mdl->drift_kind)) return launch_custom<R>(2, mdl, o, N, T, t, y, ll, fm, fP, sm, sP, status, stream);
#define X(KIND, DRIFT, D_, M_)                                                           \
  if (mdl->drift_kind == KIND && mdl->state_dim == D_ && mdl->emission_dim == M_)        \
    return run_eks_reg<R, D_, M_, DRIFT<R, D_>>(mdl, o, N, T, t, y, ll, fm, fP, sm, sP, status, stream);
  CDKF_REG_SHAPES(X)
#undef X
  return launch_ekf_smoother_wg<R>(mdl, o, N, T, t, y, ll, fm, fP, sm, sP, status, stream);
}

template int launch_ekf_smoother<float>(const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const float*,
                                        const float*, float*, float*, float*, float*, float*, int32_t*, hipStream_t);
template int launch_ekf_smoother<double>(const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const double*,
                                         const double*, double*, double*, double*, double*, double*, int32_t*,
                                         hipStream_t);

bool kernel_available(const cdkf_model* mdl, const cdkf_opts* o, int algo, int bytes_per_real) {
  if (custom_kind(mdl->drift_kind)) return custom_shape_available(mdl, algo == 1 ? nullptr : o);
  if (reg_shape_available(mdl)) return true;
  return wg_shape_available(mdl, bytes_per_real);
}

}  // namespace cdkf
