// launch_wg8.hip -- the workgroup-per-trajectory kernels at eight and sixteen owned covariance entries per thread (state_dim 46 .. 64 at
// 512 threads).  Their own translation unit because of how it is built (Makefile): at plain -O2 / -O3 the fp64 instantiation with eight
// entries per thread (48 double-precision slopes per thread, ~1400 spilled registers) returned NaN from the second observation on and
// took seconds per sweep on gfx950 / ROCm 7.2 (scripts/dbg_wg_ept8.py).  Rounds 3 / 4 shipped it at -O1; round 5 found what goes wrong
// -- the greedy register allocator's sub-register liveness tracking of 64-bit VGPR pairs in spill-heavy kernels (launch_custom.hip:
// rtc_policy, NOTES.md R5.1) -- and builds it at -O3 with `-mllvm -enable-subreg-liveness=0`
// (tests/test_gpu_wg.py::test_workgroup_kernels_eight_entries_per_thread is the guard; the host build of the same templates is clean
// under the four sanitizers: tests/test_hostsim.py).
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
