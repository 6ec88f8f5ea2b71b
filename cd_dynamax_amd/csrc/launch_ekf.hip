// launch_ekf.hip -- EKF filter sweep: kernel selection and launch.
#include "cdkf_launch.h"

namespace cdkf {

// true if the emission picks the first M state coordinates: H = I[:M], bias = 0
static bool emission_is_selection(const cdkf_model* mdl) {
  const int d = mdl->state_dim, m = mdl->emission_dim;
  if (m > d) return false;
  for (int r = 0; r < m; ++r) {
    if (mdl->h_bias[r] != 0.0) return false;
    for (int j = 0; j < d; ++j)
      if (mdl->H[r * d + j] != (r == j ? 1.0 : 0.0)) return false;
  }
  return true;
}

template <typename R, int D, int M, typename Drift>
static int run_ekf_reg(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll,
                       R* fm, R* fP, R* pm, R* pP, int32_t* status, hipStream_t stream) {
  RegArgs<R, D, M, Drift> a;
  fill_reg_args(a, mdl, o, N, T, t, y, ll, fm, fP, pm, pP, status);
  const dim3 grid((unsigned)((N + 63) / 64)), block(64);
  const bool all = fm && fP && pm && pP, none = !fm && !fP && !pm && !pP;
  using AS = RegArgs<R, D, (M <= D ? M : D), Drift>;  // HSEL variants exist only for M <= D
#define CDKF_LAUNCH(ZEROTH, HSEL, MM, ARGS)                                                                       \
  do {                                                                                                            \
    if (all)                                                                                                      \
      hipLaunchKernelGGL((ekf_filter_reg_kernel<R, D, MM, Drift, ZEROTH, HSEL, kOutAll>), grid, block, 0, stream, ARGS);  \
    else if (none)                                                                                                \
      hipLaunchKernelGGL((ekf_filter_reg_kernel<R, D, MM, Drift, ZEROTH, HSEL, kOutNone>), grid, block, 0, stream, ARGS); \
    else                                                                                                          \
      hipLaunchKernelGGL((ekf_filter_reg_kernel<R, D, MM, Drift, ZEROTH, HSEL, kOutSome>), grid, block, 0, stream, ARGS); \
  } while (0)
  if (o->state_order == CDKF_ORDER_ZEROTH) {
    hipLaunchKernelGGL((ekf_filter_reg_kernel<R, D, M, Drift, true, false, kOutSome>), grid, block, 0, stream, a);
  } else if (M <= D && emission_is_selection(mdl)) {
    CDKF_LAUNCH(false, true, (M <= D ? M : D), *reinterpret_cast<AS*>(&a));
  } else {
    CDKF_LAUNCH(false, false, M, a);
  }
#undef CDKF_LAUNCH
  CDKF_HIP_CHECK(hipGetLastError());
  return CDKF_OK;
}

template <typename R>
int launch_ekf_filter(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll,
                      R* fm, R* fP, R* pm, R* pP, int32_t* status, hipStream_t stream) {
#define X(KIND, DRIFT, D_, M_)                                                           \
  if (mdl->drift_kind == KIND && mdl->state_dim == D_ && mdl->emission_dim == M_)        \
    return run_ekf_reg<R, D_, M_, DRIFT<R, D_>>(mdl, o, N, T, t, y, ll, fm, fP, pm, pP, status, stream);
  CDKF_REG_SHAPES(X)
#undef X
  set_error("EKF filter: no kernel for drift_kind=%d state_dim=%d emission_dim=%d", mdl->drift_kind, mdl->state_dim,
            mdl->emission_dim);
  return CDKF_EUNSUPPORTED;
}

template int launch_ekf_filter<float>(const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const float*, const float*,
                                      float*, float*, float*, float*, float*, int32_t*, hipStream_t);
template int launch_ekf_filter<double>(const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const double*,
                                       const double*, double*, double*, double*, double*, double*, int32_t*,
                                       hipStream_t);

}  // namespace cdkf
