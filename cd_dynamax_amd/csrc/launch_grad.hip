// launch_grad.hip -- log-likelihood + gradient w.r.t. the drift parameters (cdkf_grad_kernels.h): shape registry and launch.
#include "cdkf_grad_kernels.h"
#include "cdkf_launch.h"

// X(drift_kind, DriftTemplate, D, M)
#define CDKF_GRAD_SHAPES(X)                   \
  X(CDKF_DRIFT_LORENZ63, DriftLorenz63, 3, 1) \
  X(CDKF_DRIFT_LORENZ63, DriftLorenz63, 3, 2) \
  X(CDKF_DRIFT_LORENZ63, DriftLorenz63, 3, 3) \
  X(CDKF_DRIFT_LINEAR, DriftLinear, 1, 1)     \
  X(CDKF_DRIFT_LINEAR, DriftLinear, 2, 1)     \
  X(CDKF_DRIFT_LINEAR, DriftLinear, 2, 2)     \
  X(CDKF_DRIFT_LINEAR, DriftLinear, 3, 3)

namespace cdkf {

static bool sens_shape_available(const cdkf_model* mdl, const cdkf_opts* o) {
  if (o->state_order == CDKF_ORDER_ZEROTH || o->num_iter != 1 || o->forecast) return false;
#define X(KIND, DRIFT, D_, M_) \
  if (mdl->drift_kind == KIND && mdl->state_dim == D_ && mdl->emission_dim == M_) return true;
  CDKF_GRAD_SHAPES(X)
#undef X
  return false;
}

bool grad_shape_available(const cdkf_model* mdl, const cdkf_opts* o) {
  return sens_shape_available(mdl, o) || adjoint_shape_available(mdl, o);
}

template <typename R, int D, int M, typename Drift>
static int run_grad(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll,
                    R* grad, int32_t* status, hipStream_t stream) {
  GradArgs<R, D, M, Drift> ga;
  fill_reg_args(ga.a, mdl, o, N, T, t, y, ll, (R*)nullptr, (R*)nullptr, (R*)nullptr, (R*)nullptr, status);
  ga.grad = grad;
  const long lanes = (long)N * DriftGrad<R, D, Drift>::NPAR;
  const RegGrouping g = reg_grouping(lanes, (int)sizeof(R));
  ga.a.lanes = g.lanes;
  ga.a.xcd_shift = g.xcd_shift;
  const dim3 grid(g.blocks), block(64);
  note_kernel("ekf_grad_reg_kernel<%s, %d, %d, ", real_name<R>(), D, M);
  if (o->solver != CDKF_SOLVER_DOPRI5 || o->adaptive)  // run-time tableau / adaptive steps: the tangents ride on the primal's steps
    hipLaunchKernelGGL((ekf_grad_reg_kernel<R, D, M, Drift, true>), grid, block, 0, stream, ga);
  else
    hipLaunchKernelGGL((ekf_grad_reg_kernel<R, D, M, Drift>), grid, block, 0, stream, ga);
  CDKF_HIP_CHECK(hipGetLastError());
  return CDKF_OK;
}

template <typename R>
int launch_ekf_grad(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll,
                    R* grad, R* grad_model, int32_t* status, hipStream_t stream) {
  // drift parameters only: forward sensitivities where a register-resident kernel exists (one sweep, no workspace);
  // otherwise, and whenever the model block is requested, the forward + reverse sweep pair
  const bool sens = !grad_model && sens_shape_available(mdl, o);
  if (!sens && adjoint_shape_available(mdl, o))
    return launch_ekf_grad_adjoint<R>(mdl, o, N, T, t, y, ll, grad, grad_model, status, stream);
  if (!sens) {
    set_error("loglik_grad: no kernel for drift_kind=%d state_dim=%d emission_dim=%d state_order=%d num_iter=%d "
              "(forward sensitivities: register-resident Lorenz-63 / linear shapes; reverse sweep: state_dim, emission_dim "
              "<= 8, MLP hidden <= 64 with state_order first; num_iter 1, state_order first|second)",
              mdl->drift_kind, mdl->state_dim, mdl->emission_dim, o->state_order, o->num_iter);
    return CDKF_EUNSUPPORTED;
  }
#define X(KIND, DRIFT, D_, M_)                                                    \
  if (mdl->drift_kind == KIND && mdl->state_dim == D_ && mdl->emission_dim == M_) \
    return run_grad<R, D_, M_, DRIFT<R, D_>>(mdl, o, N, T, t, y, ll, grad, status, stream);
  CDKF_GRAD_SHAPES(X)
#undef X
  return CDKF_EUNSUPPORTED;
}

template int launch_ekf_grad<float>(const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const float*, const float*,
                                    float*, float*, float*, int32_t*, hipStream_t);
template int launch_ekf_grad<double>(const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const double*, const double*,
                                     double*, double*, double*, int32_t*, hipStream_t);

}  // namespace cdkf
