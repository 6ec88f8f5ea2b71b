// launch_adjwg.hip -- the shape-generic reverse sweep (workgroup per trajectory, cdkf_adjoint_wg_kernels.h) in its own translation unit.
#include "cdkf_launch.h"
#include "cdkf_adjoint_wg_kernels.h"

namespace cdkf {

bool adjoint_wg_fits(int d, int m, int bytes_per_real) {
  return d >= 1 && m >= 1 && d <= 64 && m <= 64 && (size_t)awg_lds_reals(d, m) * bytes_per_real + 64 <= kLdsLimit - 256;
}
// ... with the MLP drift's weights, work vectors and tangent images behind the nine matrices (hidden sizes <= 64)
bool adjoint_wg_fits_mlp(int d, int m, int h1, int h2, int bytes_per_real) {
  return d >= 1 && m >= 1 && d <= 64 && m <= 64 && h1 >= 1 && h2 >= 1 && h1 <= 64 && h2 <= 64 &&
         (size_t)(awg_lds_reals(d, m) + awg_mlp_lds_reals(d, h1, h2)) * bytes_per_real + 64 <= kLdsLimit - 256;
}
long adjoint_wg_scratch_reals(int d, int cap) { return awg_scratch_reals(d, cap); }
int custom_awg_geometry(int d, int m, int bytes_per_real, int* ne, size_t* lds) {
  if (!adjoint_wg_fits(d, m, bytes_per_real)) return 1;
  *ne = awg_entries_per_thread(d) <= 8 ? 8 : 16;
  *lds = (size_t)awg_lds_reals(d, m) * bytes_per_real + 64;
  return 0;
}

template <typename R>
int launch_adjoint_wg_kernel(const WgArgs<R>& a, R* grad, R* grad_model, R* scratch, int cap, hipStream_t stream) {
  if (a.kind == kDriftMlp) {  // the network's passes: an instantiation of its own, LDS plan + awg_mlp_lds_reals
    if (!adjoint_wg_fits_mlp(a.d, a.m, a.h1, a.h2, (int)sizeof(R)) || awg_entries_per_thread(a.d) > 8) {
      set_error("reverse sweep, MLP drift: state_dim %d / emission_dim %d / hidden %d, %d need %zu bytes of LDS in fp%d (nine q x q matrices + "
                "the network's weights and tangent images; the CU has %zu)", a.d, a.m, a.h1, a.h2,
                (size_t)(awg_lds_reals(a.d, a.m) + awg_mlp_lds_reals(a.d, a.h1, a.h2)) * sizeof(R), (int)sizeof(R) * 8, (size_t)kLdsLimit);
      return CDKF_EUNSUPPORTED;
    }
    const size_t lds_mlp = (size_t)(awg_lds_reals(a.d, a.m) + awg_mlp_lds_reals(a.d, a.h1, a.h2)) * sizeof(R) + 64;
    if (once_per_device([] { return wg_raise_lds_cap(&ekf_adjoint_wg_kernel<R, 8, true>); })) return CDKF_EHIP;
    note_kernel("ekf_adjoint_wg_kernel<%s, 8, true>", real_name<R>());
    hipLaunchKernelGGL((ekf_adjoint_wg_kernel<R, 8, true>), dim3((unsigned)a.N), dim3(kAwgThreads), lds_mlp, stream, a, grad, grad_model,
                       scratch, awg_scratch_reals(a.d, cap), cap);
    CDKF_HIP_CHECK(hipGetLastError());
    return CDKF_OK;
  }
  if (!adjoint_wg_fits(a.d, a.m, (int)sizeof(R))) {
    set_error("reverse sweep: state_dim %d / emission_dim %d need %zu bytes of LDS in fp%d (nine q x q matrices; the limit is q = 43 in "
              "fp64, 62 in fp32)", a.d, a.m, (size_t)awg_lds_reals(a.d, a.m) * sizeof(R), (int)sizeof(R) * 8);
    return CDKF_EUNSUPPORTED;
  }
  const size_t lds = (size_t)awg_lds_reals(a.d, a.m) * sizeof(R) + 64;
  const long scratch_stride = awg_scratch_reals(a.d, cap);
  // the slopes and stage cotangents of a thread's covariance entries stay in registers: 8 entries per thread up to d = 42, else 16
  if (a.kind >= CDKF_DRIFT_CUSTOM_BASE)  // a drift given as source: the kernel is compiled with it at run time (launch_custom.hip)
    return launch_custom_awg<R>(a, grad, grad_model, scratch, scratch_stride, cap, awg_entries_per_thread(a.d) <= 8 ? 8 : 16, lds, stream);
  if (awg_entries_per_thread(a.d) <= 8) {
    if (once_per_device([] { return wg_raise_lds_cap(&ekf_adjoint_wg_kernel<R, 8>); })) return CDKF_EHIP;
    note_kernel("ekf_adjoint_wg_kernel<%s, 8>", real_name<R>());
    hipLaunchKernelGGL((ekf_adjoint_wg_kernel<R, 8>), dim3((unsigned)a.N), dim3(kAwgThreads), lds, stream, a, grad, grad_model, scratch,
                       scratch_stride, cap);
  } else {
    if (once_per_device([] { return wg_raise_lds_cap(&ekf_adjoint_wg_kernel<R, 16>); })) return CDKF_EHIP;
    note_kernel("ekf_adjoint_wg_kernel<%s, 16>", real_name<R>());
    hipLaunchKernelGGL((ekf_adjoint_wg_kernel<R, 16>), dim3((unsigned)a.N), dim3(kAwgThreads), lds, stream, a, grad, grad_model, scratch,
                       scratch_stride, cap);
  }
  CDKF_HIP_CHECK(hipGetLastError());
  return CDKF_OK;
}
template int launch_adjoint_wg_kernel<float>(const WgArgs<float>&, float*, float*, float*, int, hipStream_t);
template int launch_adjoint_wg_kernel<double>(const WgArgs<double>&, double*, double*, double*, int, hipStream_t);

}  // namespace cdkf
