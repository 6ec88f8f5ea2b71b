// launch_adj.hip -- the reverse (adjoint) sweep / state_dim <= 8 smoother kernel of cdkf_adjoint_kernels.h in its own translation
// unit: it is the slowest thing in the library to compile, and nothing else needs to be rebuilt with it.
#include "cdkf_launch.h"
#include "cdkf_adjoint_kernels.h"

namespace cdkf {

template <typename R, bool MLP, bool SMOOTH>
int launch_adjoint_kernel(const WgArgs<R>& a, R* grad, R* grad_model, hipStream_t stream) {
  if (once_per_device([] { return wg_raise_lds_cap(&ekf_adjoint_wave8_kernel<R, MLP, SMOOTH>); })) return CDKF_EHIP;
  constexpr int WAVES = adj_waves<R, MLP>();
  constexpr size_t lds = adj_lds_bytes<R, MLP>();
  const dim3 grid((unsigned)((a.N + WAVES - 1) / WAVES)), block(64 * WAVES);
  auto kernel = ekf_adjoint_wave8_kernel<R, MLP, SMOOTH>;
  note_kernel("ekf_adjoint_wave8_kernel<%s, %s, %s>", real_name<R>(), MLP ? "true" : "false", SMOOTH ? "true" : "false");
  hipLaunchKernelGGL(kernel, grid, block, lds, stream, a, grad, grad_model);
  CDKF_HIP_CHECK(hipGetLastError());
  return CDKF_OK;
}

#define CDKF_ADJ_INST(R)                                                                          \
  template int launch_adjoint_kernel<R, true, false>(const WgArgs<R>&, R*, R*, hipStream_t);    \
  template int launch_adjoint_kernel<R, false, false>(const WgArgs<R>&, R*, R*, hipStream_t);   \
  template int launch_adjoint_kernel<R, true, true>(const WgArgs<R>&, R*, R*, hipStream_t);     \
  template int launch_adjoint_kernel<R, false, true>(const WgArgs<R>&, R*, R*, hipStream_t);
CDKF_ADJ_INST(float)
CDKF_ADJ_INST(double)
#undef CDKF_ADJ_INST

}  // namespace cdkf
