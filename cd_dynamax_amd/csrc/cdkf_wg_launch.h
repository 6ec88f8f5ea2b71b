// cdkf_wg_launch.h -- launch of one (filter, smoother) pair of workgroup-per-trajectory kernels at a given number of owned entries
// per thread; shared by launch_wg.hip (EPT <= 4) and launch_wg8.hip (EPT >= 8, built at -O1: see the Makefile).
#pragma once
#include "cdkf_launch.h"
#include "cdkf_wg2_kernels.h"

namespace cdkf {

template <typename R, int EPT>
inline int launch_wg_pair(const WgArgs<R>& a, bool filter, bool smoother, int threads, size_t lds_f, size_t lds_s,
                          hipStream_t stream) {
  if (once_per_device([] {
        return wg_raise_lds_cap(ekf_filter_wg_kernel<R, EPT, false, kDriftAny>) | wg_raise_lds_cap(ekf_filter_wg_kernel<R, EPT, true, kDriftAny>) |
               wg_raise_lds_cap(ekf_filter_wg_kernel<R, EPT, false, kDriftLorenz96>) | wg_raise_lds_cap(ekf_smoother_wg_kernel<R, EPT>);
      }))
    return CDKF_EHIP;
  if (filter) {
    const dim3 grid((unsigned)a.N), block(threads);
    note_kernel("ekf_filter_wg_kernel<%s, %d, %s, ", real_name<R>(), EPT, a.ukf ? "true" : "false");
    if (a.ukf)
      hipLaunchKernelGGL((ekf_filter_wg_kernel<R, EPT, true, kDriftAny>), grid, block, lds_f, stream, a);
    else if (a.kind == kDriftLorenz96)
      hipLaunchKernelGGL((ekf_filter_wg_kernel<R, EPT, false, kDriftLorenz96>), grid, block, lds_f, stream, a);
    else
      hipLaunchKernelGGL((ekf_filter_wg_kernel<R, EPT, false, kDriftAny>), grid, block, lds_f, stream, a);
    CDKF_HIP_CHECK(hipGetLastError());
  }
  if (smoother) {
    note_kernel("ekf_smoother_wg_kernel<%s, %d>", real_name<R>(), EPT);
    hipLaunchKernelGGL((ekf_smoother_wg_kernel<R, EPT>), dim3((unsigned)a.N), dim3(threads), lds_s, stream, a);
    CDKF_HIP_CHECK(hipGetLastError());
  }
  return CDKF_OK;
}

// EPT = 8, 16 (launch_wg8.hip)
template <typename R>
int launch_wg_pair_wide(const WgArgs<R>& a, int ept, bool filter, bool smoother, int threads, size_t lds_f, size_t lds_s, hipStream_t stream);

}  // namespace cdkf
