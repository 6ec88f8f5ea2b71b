// launch_w8.hip -- the wavefront-per-trajectory filter sweep for state_dim <= 8 (cdkf_wave8_kernels.h; BASELINE config 5's forward
// pass) in its own translation unit: it builds in seconds, launch_wg.hip takes minutes.
#include "cdkf_launch.h"
#include "cdkf_wave8_kernels.h"
#include "cdkf_wave8s_kernels.h"

#include <cstdlib>

namespace cdkf {

// MLP drift: one trajectory over two wavefronts, four trajectories per compute unit (cdkf_wave8s_kernels.h).  CDKF_W8_SPLIT = 0 / 2 force
// either sweep (the A/B switch).  Default, measured on config 5's slice (1024 x 1000): fp32 takes the split
// (33.7 -> 28.8 ms 'second', 20.6 -> 18.5 'first'), fp64 does not (50.6 -> 52.5, 31.0 -> 32.5): an MFMA of these shapes holds its SIMD's
// vector issue for all of its 64 (fp64) / 32 (fp32) cycles, so a second wavefront on the SIMD can only fill the first one's WAITS -- in
// fp64 the products are two thirds of the busy time and what is left to fill does not pay for the barriers.  Four wavefronts per
// trajectory were slower in both precisions (fp32 36.6 ms, fp64 76.7).
template <typename R>
int w8_split_default() {
  return sizeof(R) == 4 ? 2 : 0;
}
template <typename R, int NW>
int launch_wave8s(const WgArgs<R>& a, hipStream_t stream) {
  const size_t lds = sizeof(R) * (size_t)wave8s_lds_reals<NW>() + 64;
  note_kernel("ekf_filter_wave8s_kernel<%s, %d, %s>", real_name<R>(), NW, a.order == 2 ? "true" : "false");
  if (a.order == 2)
    hipLaunchKernelGGL((ekf_filter_wave8s_kernel<R, NW, true>), dim3((unsigned)a.N), dim3(64 * NW), lds, stream, a);
  else
    hipLaunchKernelGGL((ekf_filter_wave8s_kernel<R, NW, false>), dim3((unsigned)a.N), dim3(64 * NW), lds, stream, a);
  CDKF_HIP_CHECK(hipGetLastError());
  return CDKF_OK;
}

template <typename R>
int launch_wave8(const WgArgs<R>& a, hipStream_t stream) {
  if (a.kind == kDriftMlp && a.o_w2pad > 0 && !a.cj) {  // (the padded copy of W2: wg_prepare; mean jumps: the one-wavefront sweep)
    static const int env_split = [] {
      const char* e = std::getenv("CDKF_W8_SPLIT");
      return e ? std::atoi(e) : -1;
    }();
    const int split = env_split >= 0 ? env_split : w8_split_default<R>();
    if (split == 2) return launch_wave8s<R, 2>(a, stream);
  }
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
