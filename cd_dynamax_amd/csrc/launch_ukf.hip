// launch_ukf.hip -- UKF filter sweep: kernel selection and launch.
#include "cdkf_launch.h"
#include "cdkf_lpe_kernels.h"

namespace cdkf {

template <typename R, int D, int M, typename Drift>
static int run_ukf_reg(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll,
                       R* fm, R* fP, R* pm, R* pP, int32_t* status, hipStream_t stream) {
  RegArgs<R, D, M, Drift> a;
  fill_reg_args(a, mdl, o, N, T, t, y, ll, fm, fP, pm, pP, status);
  // small Lorenz-63 batches with H = I: sixteen lanes per trajectory, the unscented moment equations in closed form (LpeRhs<R, true>)
  if (!try_lpe(a, mdl, o, stream, true)) launch_filter_reg<R, D, M, Drift, true, false, false>(a, stream);
  CDKF_HIP_CHECK(hipGetLastError());
  return CDKF_OK;
}

template <typename R>
int launch_ukf_filter(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll,
                      R* fm, R* fP, R* pm, R* pP, int32_t* status, hipStream_t stream) {
  if (custom_kind(mdl->drift_kind)) return launch_custom<R>(1, mdl, o, N, T, t, y, ll, fm, fP, pm, pP, status, stream);
#define X(KIND, DRIFT, D_, M_)                                                           \
  if (mdl->drift_kind == KIND && mdl->state_dim == D_ && mdl->emission_dim == M_)        \
    return run_ukf_reg<R, D_, M_, DRIFT<R, D_>>(mdl, o, N, T, t, y, ll, fm, fP, pm, pP, status, stream);
  CDKF_REG_SHAPES(X)
#undef X
  return launch_ukf_filter_wg<R>(mdl, o, N, T, t, y, ll, fm, fP, pm, pP, status, stream);
}

template int launch_ukf_filter<float>(const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const float*, const float*,
                                      float*, float*, float*, float*, float*, int32_t*, hipStream_t);
template int launch_ukf_filter<double>(const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const double*,
                                       const double*, double*, double*, double*, double*, double*, int32_t*,
                                       hipStream_t);

}  // namespace cdkf
