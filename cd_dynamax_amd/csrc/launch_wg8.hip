// launch_wg8.hip -- the workgroup-per-trajectory kernels at eight and sixteen owned covariance entries per thread (state_dim 46 .. 64 at
// 512 threads).  Their own translation unit because it is built at -O1 (Makefile): at -O2 / -O3 the fp64 instantiation with eight
// entries per thread (48 double-precision slopes per thread, ~1400 spilled registers) returned NaN from the second observation on and
// took seconds per sweep on gfx950 / ROCm 7.2 (scripts/dbg_wg_ept8.py).  Round 5 narrowed the family this belongs to down to the greedy
// register allocator at the register limit (launch_custom.hip: rtc_policy, NOTES.md R5.1): the basic VGPR allocator at -O3 is right here
// too, but slower than -O1, which stays (tests/test_gpu_wg.py::test_workgroup_kernels_eight_entries_per_thread is the guard; the host
// build of the same templates is clean under the four sanitizers: tests/test_hostsim.py).
#include "cdkf_wg_launch.h"

namespace cdkf {

template <typename R>
int launch_wg_pair_wide(const WgArgs<R>& a, int ept, bool filter, bool smoother, int threads, size_t lds_f, size_t lds_s, hipStream_t stream) {
  switch (ept) {
    case 8: return launch_wg_pair<R, 8>(a, filter, smoother, threads, lds_f, lds_s, stream);
    case 16: return launch_wg_pair<R, 16>(a, filter, smoother, threads, lds_f, lds_s, stream);
    default: set_error("launch_wg_pair_wide: %d entries per thread", ept); return CDKF_EUNSUPPORTED;
  }
}

template int launch_wg_pair_wide<float>(const WgArgs<float>&, int, bool, bool, int, size_t, size_t, hipStream_t);
template int launch_wg_pair_wide<double>(const WgArgs<double>&, int, bool, bool, int, size_t, size_t, hipStream_t);

}  // namespace cdkf
