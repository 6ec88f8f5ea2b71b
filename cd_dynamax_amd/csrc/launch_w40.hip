// launch_w40.hip -- Lorenz-96 at state dimension 40 (BASELINE config 4): the wavefront-per-trajectory sweeps of
// cdkf_wave40_kernels.h, in their own translation unit (they build in seconds; launch_wg.hip's reverse sweep takes minutes).
#include "cdkf_launch.h"
#include "cdkf_wave40_kernels.h"

namespace cdkf {

// Lorenz-96 observed through H = I at state_dim 40 (BASELINE config 4) -- and through any selection of its components, at every state
// dimension the kernels are instantiated for (the multiples of four from 12 to 40, where the index table's offset fields end; the
// 16-wide panels of the factorisation / solves take a last panel of 4, 8 or 12 columns):
// wavefront-per-trajectory sweep (cdkf_wave40_kernels.h)
static bool wave40_dim(int d) { return d >= 12 && d <= 40 && d % 4 == 0; }
// the emission picks m <= d state components: every row of H a unit vector, no two rows alike, no bias (H = I, H = I[:m], every
// other component, ...)
static bool emission_selects_components(const cdkf_model* mdl) {
  const int d = mdl->state_dim, m = mdl->emission_dim;
  if (m < 1 || m > d) return false;
  bool taken[64] = {false};
  for (int r = 0; r < m; ++r) {
    if (mdl->h_bias[r] != 0.0) return false;
    int col = -1;
    for (int j = 0; j < d; ++j) {
      const double h = mdl->H[r * d + j];
      if (h == 0.0) continue;
      if (h != 1.0 || col >= 0) return false;
      col = j;
    }
    if (col < 0 || taken[col]) return false;
    taken[col] = true;
  }
  return true;
}
bool wave40_shape(const cdkf_model* mdl, const cdkf_opts* o) {
  if (env_flag("CDKF_NO_WAVE40")) return false;  // A/B and tests: keep the workgroup kernels
  const int d = mdl->state_dim, m = mdl->emission_dim;
  if (mdl->drift_kind != CDKF_DRIFT_LORENZ96 || !wave40_dim(d) || !emission_selects_components(mdl)) return false;
  if (o->num_iter != 1 || o->forecast || o->state_order == CDKF_ORDER_ZEROTH || o->solver != CDKF_SOLVER_DOPRI5 || o->adaptive)
    return false;
  for (int r = 0; r < m; ++r)
    for (int c = 0; c < r; ++c)
      if (mdl->R[r * m + c] != mdl->R[c * m + r]) return false;  // P - X^T S X is formed as P - Y^T Y + 1e-9 X^T X
  return true;
}

template <typename R, int D>
static int launch_wave40_d(const WgArgs<R>& a, hipStream_t stream, bool backward) {
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
  const size_t lds = sizeof(R) * (size_t)wave40_lds_reals<D>() + 64;
  const unsigned blocks = (unsigned)((a.N + W40<D>::kWaves - 1) / W40<D>::kWaves);
  WgArgs<R> b = a;
  b.forecast = 0;
#ifdef CDKF_W40_PROFILE  // scripts/w40_prof_build.sh: mask of phases to skip (the shipped library has no such switch)
  if (const char* ab = getenv("CDKF_W40_ABLATE")) b.forecast = atoi(ab);
#endif
  // a diagonal R: one factorisation per update (the log-likelihood's terms by first-order corrections from the gain's factor: kernel
  // header); CDKF_W40_TWO_FACTORS=1 keeps the two systems in lockstep (A/B, tests)
  if (a.r_diag && !env_flag("CDKF_W40_TWO_FACTORS")) {
    if (once_per_device([] { return wg_raise_lds_cap(ekf_filter_wave_l96_kernel<R, D, true>); })) return CDKF_EHIP;
    note_kernel("ekf_filter_wave_l96_kernel<%s, %d, true>", real_name<R>(), D);
    hipLaunchKernelGGL((ekf_filter_wave_l96_kernel<R, D, true>), dim3(blocks), dim3(64 * W40<D>::kWaves), lds, stream, b);
  } else {
    if (once_per_device([] { return wg_raise_lds_cap(ekf_filter_wave_l96_kernel<R, D, false>); })) return CDKF_EHIP;
    note_kernel("ekf_filter_wave_l96_kernel<%s, %d>", real_name<R>(), D);
    hipLaunchKernelGGL((ekf_filter_wave_l96_kernel<R, D, false>), dim3(blocks), dim3(64 * W40<D>::kWaves), lds, stream, b);
  }
  CDKF_HIP_CHECK(hipGetLastError());
  return CDKF_OK;
}

template <typename R>
int launch_wave40(const WgArgs<R>& a, hipStream_t stream, bool backward) {
  switch (a.d) {
#define CDKF_W40_CASE(D_) \
  case D_: return launch_wave40_d<R, D_>(a, stream, backward);
    CDKF_W40_CASE(12) CDKF_W40_CASE(16) CDKF_W40_CASE(20) CDKF_W40_CASE(24) CDKF_W40_CASE(28) CDKF_W40_CASE(32) CDKF_W40_CASE(36)
    CDKF_W40_CASE(40)
#undef CDKF_W40_CASE
    default: set_error("wavefront-per-trajectory Lorenz-96 sweep: state_dim %d is not instantiated", a.d); return CDKF_EUNSUPPORTED;
  }
}

template int launch_wave40<float>(const WgArgs<float>&, hipStream_t, bool);
template int launch_wave40<double>(const WgArgs<double>&, hipStream_t, bool);

}  // namespace cdkf
