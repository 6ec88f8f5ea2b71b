// launch_w8.hip -- the wavefront-per-trajectory filter sweep for state_dim <= 8 (cdkf_wave8_kernels.h; BASELINE config 5's forward
// pass) in its own translation unit: it builds in seconds, launch_wg.hip takes minutes.
#include "cdkf_launch.h"
#include "cdkf_wave8_kernels.h"

namespace cdkf {

template <typename R>
int launch_wave8(const WgArgs<R>& a, hipStream_t stream) {
  if (once_per_device([] { return wg_raise_lds_cap(ekf_filter_wave8_kernel<R>); })) return CDKF_EHIP;
  const size_t lds = sizeof(R) * (size_t)wave8_lds_reals(a.kind) + 64;
  const unsigned blocks = (unsigned)((a.N + kW8Waves - 1) / kW8Waves);
  note_kernel("ekf_filter_wave8_kernel<%s>", real_name<R>());
  hipLaunchKernelGGL(ekf_filter_wave8_kernel<R>, dim3(blocks), dim3(64 * kW8Waves), lds, stream, a);
  CDKF_HIP_CHECK(hipGetLastError());
  return CDKF_OK;
}

template int launch_wave8<float>(const WgArgs<float>&, hipStream_t);
template int launch_wave8<double>(const WgArgs<double>&, hipStream_t);

}  // namespace cdkf
