// launch_w40.hip -- Lorenz-96 at state dimension 40 (BASELINE config 4): the wavefront-per-trajectory sweeps of
// cdkf_wave40_kernels.h, in their own translation unit (they build in seconds; launch_wg.hip's reverse sweep takes minutes).
#include "cdkf_launch.h"
#include "cdkf_wave40_kernels.h"

namespace cdkf {

// Lorenz-96 with H = I at state_dim 40 (BASELINE config 4): wavefront-per-trajectory sweep (cdkf_wave40_kernels.h)
bool wave40_shape(const cdkf_model* mdl, const cdkf_opts* o) {
  if (getenv("CDKF_NO_WAVE40")) return false;  // A/B and tests: keep the workgroup kernels
  const int d = mdl->state_dim;
  if (mdl->drift_kind != CDKF_DRIFT_LORENZ96 || d != 40 || mdl->emission_dim != d || !emission_is_selection(mdl)) return false;
  if (o->num_iter != 1 || o->forecast || o->state_order == CDKF_ORDER_ZEROTH || o->solver != CDKF_SOLVER_DOPRI5 || o->adaptive)
    return false;
  for (int r = 0; r < d; ++r)
    for (int c = 0; c < r; ++c)
      if (mdl->R[r * d + c] != mdl->R[c * d + r]) return false;  // P - X^T S X is formed as P - Y^T Y + 1e-9 X^T X
  return true;
}

template <typename R>
int launch_wave40(const WgArgs<R>& a, hipStream_t stream, bool backward) {
  constexpr int D = 40;
  if (backward) {
    if (once_per_device([] { return wg_raise_lds_cap(ekf_smoother_wave_l96_kernel<R, D>); })) return CDKF_EHIP;
    const size_t lds = sizeof(R) * (size_t)wave40_smoother_lds_reals<D>() + 64;
    const unsigned blocks = (unsigned)((a.N + W40S<D>::kWaves - 1) / W40S<D>::kWaves);
    note_kernel("ekf_smoother_wave_l96_kernel<%s, %d>", real_name<R>(), D);
    WgArgs<R> b = a;
    b.forecast = 0;
#ifdef CDKF_W40_PROFILE  // scripts/w40_prof_build.sh: mask of phases to skip (the shipped library has no such switch)
    if (const char* ab = getenv("CDKF_W40_ABLATE_BWD")) b.forecast = atoi(ab);
#endif
    hipLaunchKernelGGL((ekf_smoother_wave_l96_kernel<R, D>), dim3(blocks), dim3(64 * W40S<D>::kWaves), lds, stream, b);
    CDKF_HIP_CHECK(hipGetLastError());
    return CDKF_OK;
  }
  if (once_per_device([] { return wg_raise_lds_cap(ekf_filter_wave_l96_kernel<R, D>); })) return CDKF_EHIP;
  const size_t lds = sizeof(R) * (size_t)wave40_lds_reals<D>() + 64;
  const unsigned blocks = (unsigned)((a.N + W40<D>::kWaves - 1) / W40<D>::kWaves);
  note_kernel("ekf_filter_wave_l96_kernel<%s, %d>", real_name<R>(), D);
  WgArgs<R> b = a;
  b.forecast = 0;
#ifdef CDKF_W40_PROFILE  // scripts/w40_prof_build.sh: mask of phases to skip (the shipped library has no such switch)
  if (const char* ab = getenv("CDKF_W40_ABLATE")) b.forecast = atoi(ab);
#endif
  hipLaunchKernelGGL((ekf_filter_wave_l96_kernel<R, D>), dim3(blocks), dim3(64 * W40<D>::kWaves), lds, stream, b);
  CDKF_HIP_CHECK(hipGetLastError());
  return CDKF_OK;
}

template int launch_wave40<float>(const WgArgs<float>&, hipStream_t, bool);
template int launch_wave40<double>(const WgArgs<double>&, hipStream_t, bool);

}  // namespace cdkf
