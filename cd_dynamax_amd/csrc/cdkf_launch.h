// cdkf_launch.h -- kernel selection.  The register-resident ("reg") kernels are instantiated for the
// (drift, state_dim, emission_dim) shapes listed in CDKF_REG_SHAPES; every other shape goes to the
// LDS-resident wave-per-trajectory kernels (cdkf_wave_kernels.h).
#pragma once
#include "cdkf_host.h"

// X(drift_kind, DriftTemplate, D, M)
#define CDKF_REG_SHAPES(X)                    \
  X(CDKF_DRIFT_LORENZ63, DriftLorenz63, 3, 1) \
  X(CDKF_DRIFT_LORENZ63, DriftLorenz63, 3, 2) \
  X(CDKF_DRIFT_LORENZ63, DriftLorenz63, 3, 3) \
  X(CDKF_DRIFT_LINEAR, DriftLinear, 1, 1)     \
  X(CDKF_DRIFT_LINEAR, DriftLinear, 2, 1)     \
  X(CDKF_DRIFT_LINEAR, DriftLinear, 2, 2)     \
  X(CDKF_DRIFT_LINEAR, DriftLinear, 2, 6)     \
  X(CDKF_DRIFT_LINEAR, DriftLinear, 3, 1)     \
  X(CDKF_DRIFT_LINEAR, DriftLinear, 3, 3)     \
  X(CDKF_DRIFT_LINEAR, DriftLinear, 4, 2)     \
  X(CDKF_DRIFT_LINEAR, DriftLinear, 4, 4)

namespace cdkf {

template <typename R>
int launch_ekf_filter(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll,
                      R* fm, R* fP, R* pm, R* pP, int32_t* status, hipStream_t stream);
template <typename R>
int launch_ukf_filter(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll,
                      R* fm, R* fP, R* pm, R* pP, int32_t* status, hipStream_t stream);
template <typename R>
int launch_ekf_smoother(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y,
                        R* ll, R* fm, R* fP, R* sm, R* sP, int32_t* status, hipStream_t stream);

bool kernel_available(const cdkf_model* mdl, const cdkf_opts* o, int algo);

inline bool reg_shape_available(const cdkf_model* mdl) {
#define X(KIND, DRIFT, D_, M_) \
  if (mdl->drift_kind == KIND && mdl->state_dim == D_ && mdl->emission_dim == M_) return true;
  CDKF_REG_SHAPES(X)
#undef X
  return false;
}

}  // namespace cdkf
