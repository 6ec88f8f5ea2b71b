// launch_adjw40.hip -- the wavefront-per-trajectory reverse sweep of the Lorenz-96 model (cdkf_adjoint_w40_kernels.h) in its own
// translation unit.
#include "cdkf_launch.h"
#include "cdkf_adjoint_w40_kernels.h"
#include <cstdlib>

namespace cdkf {

// per-trajectory global scratch: `cap` step starts of a replay chunk (owned entries lane-major + the mean), then the stage inputs 1 .. 5
// of the step being reversed (owned entries lane-major)
long wave40_adjoint_scratch_reals(int d, int cap) {
  const long np = (long)d * (d + 1) / 2, epl = (np + 63) / 64;
  return (long)cap * (64 * epl + 64) + 7 * 64 * epl;  // (+ the stage inputs of the step in hand, the parked Pbar, d ll / d (L Qc L^T))
}

template <typename R, int D>
static int launch_wave40_adjoint_d(const WgArgs<R>& a, R* grad, R* grad_model, R* scratch, int cap, hipStream_t stream) {
  using A = W40A<R, D>;
  static_assert(sizeof(R) * (size_t)A::lds_reals + 64 <= kLdsLimit - 256, "two trajectories per workgroup fit the CU's LDS");
  if (cap < 1 || cap > 64) {
    set_error("reverse sweep: step starts per replay chunk must be 1 .. 64 (got %d)", cap);
    return CDKF_EINVAL;
  }
  // Two wavefronts per trajectory (ekf_adjoint_wave2_l96_kernel: every other owned slot and product tile each, one trajectory per
  // workgroup, two workgroups per CU -- all four SIMDs at work) or one (two trajectories per workgroup).  CDKF_WAVE40_ADJ_WAVES=1|2.
  int nw = 2;
  if (const char* e = getenv("CDKF_WAVE40_ADJ_WAVES")) nw = atoi(e) == 1 ? 1 : 2;
  if (nw == 2) {
    using A2 = W40A<R, D, 2>;
    static_assert(2 * (sizeof(R) * (size_t)A2::lds_reals + 64) <= kLdsLimit - 256, "two one-trajectory workgroups fit the CU's LDS");
    if (once_per_device([] { return wg_raise_lds_cap(ekf_adjoint_wave2_l96_kernel<R, D>); })) return CDKF_EHIP;
    const size_t lds2 = sizeof(R) * (size_t)A2::lds_reals + 64;
    note_kernel("ekf_adjoint_wave2_l96_kernel<%s, %d>", real_name<R>(), D);
    hipLaunchKernelGGL((ekf_adjoint_wave2_l96_kernel<R, D>), dim3((unsigned)a.N), dim3(128), lds2, stream, a, grad, grad_model, scratch,
                       wave40_adjoint_scratch_reals(D, cap), cap);
    CDKF_HIP_CHECK(hipGetLastError());
    return CDKF_OK;
  }
  if (once_per_device([] { return wg_raise_lds_cap(ekf_adjoint_wave_l96_kernel<R, D>); })) return CDKF_EHIP;
  const size_t lds = sizeof(R) * (size_t)A::lds_reals + 64;
  const unsigned blocks = (unsigned)((a.N + A::kWaves - 1) / A::kWaves);
  note_kernel("ekf_adjoint_wave_l96_kernel<%s, %d>", real_name<R>(), D);
  hipLaunchKernelGGL((ekf_adjoint_wave_l96_kernel<R, D>), dim3(blocks), dim3(64 * A::kWaves), lds, stream, a, grad, grad_model, scratch,
                     wave40_adjoint_scratch_reals(D, cap), cap);
  CDKF_HIP_CHECK(hipGetLastError());
  return CDKF_OK;
}

template <typename R>
int launch_wave40_adjoint(const WgArgs<R>& a, R* grad, R* grad_model, R* scratch, int cap, hipStream_t stream) {
  switch (a.d) {
#define CDKF_W40_CASE(D_) \
  case D_: return launch_wave40_adjoint_d<R, D_>(a, grad, grad_model, scratch, cap, stream);
    CDKF_W40_CASE(12) CDKF_W40_CASE(16) CDKF_W40_CASE(20) CDKF_W40_CASE(24) CDKF_W40_CASE(28) CDKF_W40_CASE(32) CDKF_W40_CASE(36)
    CDKF_W40_CASE(40)
#undef CDKF_W40_CASE
    default: set_error("wavefront-per-trajectory Lorenz-96 reverse sweep: state_dim %d is not instantiated", a.d); return CDKF_EUNSUPPORTED;
  }
}

template int launch_wave40_adjoint<float>(const WgArgs<float>&, float*, float*, float*, int, hipStream_t);
template int launch_wave40_adjoint<double>(const WgArgs<double>&, double*, double*, double*, int, hipStream_t);

}  // namespace cdkf
