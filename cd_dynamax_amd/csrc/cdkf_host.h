// cdkf_host.h -- host-side plumbing shared by the C-ABI translation units: error reporting, argument
// checks, conversion of the double-precision model block into the compute type, launch helpers.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/cdkf.h"
#include "cdkf_reg_kernels.h"

namespace cdkf {

void set_error(const char* fmt, ...);
int check_common(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, const void* t, const void* y,
                 const void* ll);

#define CDKF_HIP_CHECK(expr)                                                              \
  do {                                                                                    \
    hipError_t err__ = (expr);                                                            \
    if (err__ != hipSuccess) {                                                            \
      ::cdkf::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(err__), __FILE__, __LINE__); \
      return CDKF_EHIP;                                                                   \
    }                                                                                     \
  } while (0)

// RAII device buffer for the host-pointer entry points
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  int alloc(size_t bytes) {
    CDKF_HIP_CHECK(hipMalloc(&p, bytes ? bytes : 1));
    return CDKF_OK;
  }
};

// (L Qc L^T) evaluated in the compute type, left-associated like `L_t @ Qc_t @ L_t.T`
template <typename R>
void lql_packed(const double* L, const double* Qc, int d, double scale, R* out_packed) {
  std::vector<R> Lr(d * d), Qr(d * d), LQ(d * d), full(d * d);
  for (int i = 0; i < d * d; ++i) {
    Lr[i] = R(L[i]) * R(scale);
    Qr[i] = R(Qc[i]);
  }
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < d; ++j) {
      R s = 0;
      for (int k = 0; k < d; ++k) s += Lr[i * d + k] * Qr[k * d + j];
      LQ[i * d + j] = s;
    }
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < d; ++j) {
      R s = 0;
      for (int k = 0; k < d; ++k) s += LQ[i * d + k] * Lr[j * d + k];
      full[i * d + j] = s;
    }
  int e = 0;
  for (int i = 0; i < d; ++i)
    for (int j = i; j < d; ++j) out_packed[e++] = R(0.5) * (full[i * d + j] + full[j * d + i]);
}

template <typename R, int D, int M, typename Drift>
void fill_reg_args(RegArgs<R, D, M, Drift>& a, const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T,
                   const R* t, const R* y, R* ll, R* fm, R* fP, R* pm, R* pP, int32_t* status) {
  a.drift.load(mdl->theta);
  lql_packed<R>(mdl->L, mdl->Qc, D, 1.0, a.LQL);
  lql_packed<R>(mdl->L, mdl->Qc, D, o->cov_rescaling, a.LQLz);
  for (int r = 0; r < M; ++r) {
    for (int j = 0; j < D; ++j) a.H[r][j] = R(mdl->H[r * D + j]);
    a.hb[r] = R(mdl->h_bias[r]);
    for (int c = 0; c < M; ++c) a.Rm[r][c] = R(mdl->R[r * M + c]);
  }
  int e = 0;
  for (int i = 0; i < D; ++i) {
    a.m0[i] = R(mdl->m0[i]);
    for (int j = i; j < D; ++j) a.P0[e++] = R(0.5) * (R(mdl->P0[i * D + j]) + R(mdl->P0[j * D + i]));
  }
  a.dt0 = R(o->dt0);
  a.dt_final = R(o->dt_final);
  // UKF weights in the compute type (inference_ukf.py:42, 63-89)
  {
    R alpha = R(o->ukf_alpha), n = R(D);
    R lamb = alpha * alpha * (n + R(o->ukf_kappa)) - n;
    a.ukf_c = std::sqrt(n + lamb);
    a.ukf_wm0 = lamb / (n + lamb);
    a.ukf_wc0 = lamb / (n + lamb) + (R(1) - alpha * alpha + R(o->ukf_beta));
    a.ukf_wi = R(1) / (R(2) * (n + lamb));
  }
  a.max_steps = (long)o->max_steps;
  a.order = o->state_order;
  a.num_iter = o->num_iter;
  a.forecast = o->forecast;
  a.N = N;
  a.T = T;
  a.y_si = a.m_si = a.P_si = 1;
  const bool no_y = (y == nullptr);
  if (o->layout == CDKF_LAYOUT_TCN) {  // [T,w,N]: component-major inside a time step
    a.t_sn = o->t_shared ? 0 : 1;
    a.t_sk = o->t_shared ? 1 : N;
    a.y_sn = a.m_sn = a.P_sn = 1;
    a.y_sk = N * M;
    a.m_sk = N * D;
    a.P_sk = N * D * D;
    a.y_si = a.m_si = a.P_si = N;
  } else if (o->layout == CDKF_LAYOUT_TN) {  // time-major [T,N,w]
    a.t_sn = o->t_shared ? 0 : 1;
    a.t_sk = o->t_shared ? 1 : N;
    a.y_sn = M;
    a.y_sk = N * M;
    a.m_sn = D;
    a.m_sk = N * D;
    a.P_sn = D * D;
    a.P_sk = N * D * D;
  } else {  // reference layout [N,T,w]
    a.t_sn = o->t_shared ? 0 : T;
    a.t_sk = 1;
    a.y_sn = T * M;
    a.y_sk = M;
    a.m_sn = T * D;
    a.m_sk = D;
    a.P_sn = T * D * D;
    a.P_sk = D * D;
  }
  if (no_y) a.y_sn = a.y_sk = a.y_si = 0;  // every 'observation' load hits t[0]
  a.t = t;
  a.y = y ? y : t;  // forecast mode ignores the observations; keep the prefetch loads on valid memory
  a.ll = ll;
  a.fm = fm;
  a.fP = fP;
  a.pm = pm;
  a.pP = pP;
  a.status = status;
}

int select_device(const cdkf_opts* o);

}  // namespace cdkf
