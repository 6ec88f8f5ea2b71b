// launch_grad.hip -- log-likelihood + gradient w.r.t. the drift parameters (cdkf_grad_kernels.h): shape registry and launch.
#include "cdkf_grad_kernels.h"
#include "cdkf_launch.h"
#include "cdkf_lpe_grad_kernels.h"

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
  if (custom_kind(mdl->drift_kind)) return custom_grad_available(mdl, o);  // a run-time compiled drift: its derivatives by dual numbers
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
                    R* grad, int32_t* status, hipStream_t stream, bool ukf = false) {
  GradArgs<R, D, M, Drift> ga;
  fill_reg_args(ga.a, mdl, o, N, T, t, y, ll, (R*)nullptr, (R*)nullptr, (R*)nullptr, (R*)nullptr, status);
  ga.grad = grad;
  const long lanes = (long)N * DriftGrad<R, D, Drift>::NPAR;
  const RegGrouping g = reg_grouping(lanes, (int)sizeof(R));
  ga.a.lanes = g.lanes;
  ga.a.xcd_shift = g.xcd_shift;
  const dim3 grid(g.blocks), block(64);
  const bool generic = o->solver != CDKF_SOLVER_DOPRI5 || o->adaptive;  // run-time tableau / adaptive steps: the tangents ride on the primal's steps
  // (the unscented filter of a drift without curvature has the extended filter's moment equations: the same kernel)
  const bool curved = ukf && DriftGrad<R, D, Drift>::kCurved;
  note_kernel("ekf_grad_reg_kernel<%s, %d, %d, %s, %s>", real_name<R>(), D, M, generic ? "true" : "false", curved ? "true" : "false");
  if constexpr (DriftGrad<R, D, Drift>::kCurved) {
    if (curved) {
      if (generic)
        hipLaunchKernelGGL((ekf_grad_reg_kernel<R, D, M, Drift, true, true>), grid, block, 0, stream, ga);
      else
        hipLaunchKernelGGL((ekf_grad_reg_kernel<R, D, M, Drift, false, true>), grid, block, 0, stream, ga);
      CDKF_HIP_CHECK(hipGetLastError());
      return CDKF_OK;
    }
  }
  if (generic)
    hipLaunchKernelGGL((ekf_grad_reg_kernel<R, D, M, Drift, true>), grid, block, 0, stream, ga);
  else
    hipLaunchKernelGGL((ekf_grad_reg_kernel<R, D, M, Drift>), grid, block, 0, stream, ga);
  CDKF_HIP_CHECK(hipGetLastError());
  return CDKF_OK;
}

// Small Lorenz-63 batches, H = I[:m]: forward sweep + reverse sweep on the sixteen-lanes-per-trajectory grid (cdkf_lpe_grad_kernels.h).
// CDKF_NO_LPE_GRAD=1 keeps the other kernels (A/B timing, tests).  handled = false: not this kernel's case.
template <typename R, int M>
static int run_lpe_grad(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll, R* grad,
                        R* grad_model, int32_t* status, hipStream_t stream, bool* handled) {
  bool sym = true;
  for (int r = 0; r < M; ++r)
    for (int c = 0; c < r; ++c) sym = sym && R(mdl->R[r * M + c]) == R(mdl->R[c * M + r]);
  if (!sym) return CDKF_OK;  // (the reverse update is written for a symmetric S = H P H^T + R)
  // the forward sweep's four moment arrays, [T, N, comp] (a wavefront's stores and loads are contiguous pieces)
  cdkf_opts of = *o;
  of.layout_in = (o->layout_in == CDKF_LAYOUT_SAME) ? o->layout : o->layout_in;
  of.layout = CDKF_LAYOUT_TN;
  const size_t nm = (size_t)N * T * 3, nP = nm * 3;
  GradWorkspaceLease ws;
  void* wp = nullptr;
  if (ws.reserve(2 * (nm + nP) * sizeof(R), stream, &wp)) {  // no room for the moments: the caller's other kernel (the forward
    (void)hipGetLastError();                                  // sensitivities need no workspace)
    return CDKF_OK;
  }
  R* w = (R*)wp;
  RegArgs<R, 3, M, DriftLorenz63<R, 3>> a;
  fill_reg_args(a, mdl, &of, N, T, t, y, ll, w, w + nm, w + nm + nP, w + 2 * nm + nP, status);
  if (!try_lpe(a, mdl, &of, stream, false, grad_model != nullptr)) return CDKF_OK;
  CDKF_HIP_CHECK(hipGetLastError());
  *handled = true;
  constexpr bool kGrid = M == 3;  // the reverse update inside the grid (the forward sweep made the same choice: lpe_update)
  note_kernel("grad_lpe_l63_kernel<%s, %d, %s, %s>", real_name<R>(), M, kGrid ? "true" : "false", grad_model ? "true" : "false");
  const dim3 g(lpe_blocks<R>(N)), b(64);
  if (grad_model)
    hipLaunchKernelGGL((grad_lpe_l63_kernel<R, M, kGrid, true>), g, b, 0, stream, a, grad, grad_model);
  else
    hipLaunchKernelGGL((grad_lpe_l63_kernel<R, M, kGrid, false>), g, b, 0, stream, a, grad, grad_model);
  CDKF_HIP_CHECK(hipGetLastError());
  return ws.done(stream);
}

template <typename R>
static int try_lpe_grad(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll, R* grad,
                        R* grad_model, int32_t* status, hipStream_t stream, bool* handled) {
  *handled = false;
  static const bool off = [] { const char* e = std::getenv("CDKF_NO_LPE_GRAD"); return e && e[0] == '1'; }();
  if (off || mdl->drift_kind != CDKF_DRIFT_LORENZ63 || mdl->state_dim != 3 || mdl->emission_dim > 3 || !emission_is_selection(mdl) ||
      o->num_iter != 1 || o->forecast || o->solver != CDKF_SOLVER_DOPRI5 || o->adaptive || o->state_order == CDKF_ORDER_ZEROTH ||
      N < 1 || T < 1 || !y)
    return CDKF_OK;
  // drift block only: the forward-sensitivity kernel catches up once the grid has more than a wavefront per SIMD; with the model block
  // the alternative is the wavefront-per-trajectory reverse sweep (40x slower at this state dimension), whatever the batch size
  if (!grad_model && !lpe_batch_is_small(N, kLpeGrad)) return CDKF_OK;
  switch (mdl->emission_dim) {
    case 1: return run_lpe_grad<R, 1>(mdl, o, N, T, t, y, ll, grad, grad_model, status, stream, handled);
    case 2: return run_lpe_grad<R, 2>(mdl, o, N, T, t, y, ll, grad, grad_model, status, stream, handled);
    default: return run_lpe_grad<R, 3>(mdl, o, N, T, t, y, ll, grad, grad_model, status, stream, handled);
  }
}

template <typename R>
int launch_ekf_grad(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll,
                    R* grad, R* grad_model, int32_t* status, hipStream_t stream) {
  // drift parameters only: forward sensitivities where a register-resident kernel exists (one sweep, no workspace);
  // otherwise, and whenever the model block is requested, the forward + reverse sweep pair
  const bool sens = !grad_model && sens_shape_available(mdl, o);
  if (sens && custom_kind(mdl->drift_kind)) return launch_custom<R>(3, mdl, o, N, T, t, y, ll, grad, nullptr, nullptr, nullptr, status, stream);
  if (sens || grad_model) {  // (the model block as well: m0, P0, L Qc L^T, H, bias, R)
    bool handled = false;
    const int rc = try_lpe_grad<R>(mdl, o, N, T, t, y, ll, grad, grad_model, status, stream, &handled);
    if (rc || handled) return rc;
  }
  if (!sens && adjoint_shape_available(mdl, o))
    return launch_ekf_grad_adjoint<R>(mdl, o, N, T, t, y, ll, grad, grad_model, status, stream);
  // what neither covers -- update iterations above eight dimensions, emissions given as source, the MLP beyond its LDS plan (round 5) --
  // up to sixteen dimensions: forward mode through the literal recursion on dual numbers, a lane per (trajectory, leaf entry)
  if (!sens && ekf_tangent_available(mdl, o)) return launch_ekf_tangent<R>(mdl, o, N, T, t, y, ll, grad, grad_model, status, stream);
  if (!sens) {
    set_error("loglik_grad: no kernel for drift_kind=%d state_dim=%d emission_dim=%d state_order=%d num_iter=%d "
              "(forward sensitivities: register-resident Lorenz-63 / linear shapes and run-time compiled drifts; reverse sweep: "
              "state_dim, emission_dim <= 8, MLP hidden <= 64 -- Lorenz-96 / linear drifts up to 43 (fp64) / 62 (fp32); num_iter 1, "
              "state_order first|second)",
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

// the unscented filter's log-likelihood and its gradient w.r.t. the drift parameters (cdkf_ukf_loglik_grad_*): forward sensitivities
// through the closed form of the sigma-point sums -- the register-resident Lorenz-63 / linear shapes
static bool ukf_grad_closed_form(const cdkf_model* mdl, const cdkf_opts* o) {
  cdkf_opts e = *o;
  e.state_order = CDKF_ORDER_FIRST;  // (the unscented filter has no state_order; the field is ignored)
  return !custom_kind(mdl->drift_kind) && sens_shape_available(mdl, &e) && mdl->emission_kind == 0;
}
// every other model (MLP, source drifts / emissions, the built-in drifts at other shapes): the tangent sweep of the literal recursion
bool ukf_grad_shape_available(const cdkf_model* mdl, const cdkf_opts* o) {
  return ukf_grad_closed_form(mdl, o) || (mdl->n_theta >= 1 && ukf_tangent_available(mdl, o));
}
template <typename R>
int launch_ukf_grad(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll, R* grad,
                    int32_t* status, hipStream_t stream) {
  if (!ukf_grad_closed_form(mdl, o) || env_flag("CDKF_UKF_GRAD_TANGENT"))
    return launch_ukf_tangent<R>(mdl, o, N, T, t, y, ll, grad, nullptr, status, stream);  // (names what it needs when it refuses)
#define X(KIND, DRIFT, D_, M_)                                                    \
  if (mdl->drift_kind == KIND && mdl->state_dim == D_ && mdl->emission_dim == M_) \
    return run_grad<R, D_, M_, DRIFT<R, D_>>(mdl, o, N, T, t, y, ll, grad, status, stream, true);
  CDKF_GRAD_SHAPES(X)
#undef X
  set_error("ukf_loglik_grad: drift_kind=%d state_dim=%d emission_dim=%d passed the shape gate but has no instantiation", mdl->drift_kind,
            mdl->state_dim, mdl->emission_dim);
  return CDKF_EUNSUPPORTED;
}
template int launch_ukf_grad<float>(const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const float*, const float*, float*, float*,
                                    int32_t*, hipStream_t);
template int launch_ukf_grad<double>(const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const double*, const double*, double*,
                                     double*, int32_t*, hipStream_t);

template int launch_ekf_grad<float>(const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const float*, const float*,
                                    float*, float*, float*, int32_t*, hipStream_t);
template int launch_ekf_grad<double>(const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const double*, const double*,
                                     double*, double*, double*, int32_t*, hipStream_t);

}  // namespace cdkf
