// cdkf_host.h -- host-side plumbing shared by the C-ABI translation units: error reporting, argument
// checks, conversion of the double-precision model block into the compute type, launch helpers.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/cdkf.h"
#include "cdkf_reg_kernels.h"

namespace cdkf {

void set_error(const char* fmt, ...);
// records the sweep kernel a call is about to launch (cdkf_last_kernel): the name as a profile prints it, up to where it is known
void note_kernel(const char* fmt, ...);
template <typename R>
inline const char* real_name() { return sizeof(R) == 8 ? "double" : "float"; }
// a debugging switch of the environment: set AND a non-zero number ("CDKF_X=0" is off, as an unset variable is)
inline bool env_flag(const char* name) {
  const char* e = getenv(name);
  return e && atoi(e) != 0;
}
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

// opts.device >= 0 runs the call on that device; the caller's current device is put back on every way out (a call on a cuda:1
// tensor must not move the process's -- PyTorch's -- current device under the caller's feet)
struct DeviceGuard {
  int prev = -1;
  bool switched = false, good = true;
  explicit DeviceGuard(int device) {
    if (device < 0) return;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev == device) return;
    const hipError_t err = hipSetDevice(device);
    if (err != hipSuccess) {
      set_error("hipSetDevice(%d) failed: %s", device, hipGetErrorString(err));
      good = false;
      return;
    }
    switched = true;
  }
  ~DeviceGuard() {
    if (switched && prev >= 0) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
  bool ok() const { return good; }
};
#define CDKF_SELECT_DEVICE(o)                                      \
  ::cdkf::DeviceGuard device_guard__((o) ? (o)->device : -1);      \
  if (!device_guard__.ok()) return CDKF_EHIP

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

// Runge-Kutta tableaus selectable through opts.solver (CDKF_SOLVER_*); published coefficients (Euler; explicit trapezoid =
// diffrax.Heun; explicit midpoint; Ralston; Bogacki-Shampine 3(2) = diffrax.Bosh3; Tsitouras 5(4) = diffrax.Tsit5)
template <typename R>
inline bool fill_rk_tab(const cdkf_opts* o, RkTab<R>& tb) {
  const int solver = o->solver;
  static const double dp5[6][5] = {{0}, {1.0 / 5}, {3.0 / 40, 9.0 / 40}, {44.0 / 45, -56.0 / 15, 32.0 / 9},
                                   {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729},
                                   {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656}};
  static const double dp5b[6] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84};
  static const double ts5[6][5] = {{0}, {0.161}, {-0.008480655492356989, 0.335480655492357},
                                   {2.8971530571054935, -6.359448489975075, 4.3622954328695815},
                                   {5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525},
                                   {5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401,
                                    -0.028269050394068383}};
  static const double ts5b[6] = {0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081,
                                 2.324710524099774};
  double a[6][5] = {{0}}, b[6] = {0}, be[7] = {0};
  int stages = 0, order = 0, fsal = 0;
  switch (solver) {
    case CDKF_SOLVER_DOPRI5: {
      stages = 6; order = 5; fsal = 1;
      std::memcpy(a, dp5, sizeof(a)); std::memcpy(b, dp5b, sizeof(b));
      const double e[7] = {35.0 / 384 - 5179.0 / 57600, 0, 500.0 / 1113 - 7571.0 / 16695, 125.0 / 192 - 393.0 / 640,
                           -2187.0 / 6784 + 92097.0 / 339200, 11.0 / 84 - 187.0 / 2100, -1.0 / 40};
      std::memcpy(be, e, sizeof(be));
      break;
    }
    case CDKF_SOLVER_TSIT5: {
      stages = 6; order = 5; fsal = 1;
      std::memcpy(a, ts5, sizeof(a)); std::memcpy(b, ts5b, sizeof(b));
      const double e[7] = {-0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995, -0.1447110071732629,
                           0.5823571654525552, -0.45808210592918697, 0.015151515151515152};
      std::memcpy(be, e, sizeof(be));
      break;
    }
    case CDKF_SOLVER_EULER: stages = 1; b[0] = 1; break;
    case CDKF_SOLVER_HEUN: stages = 2; order = 2; a[1][0] = 1; b[0] = b[1] = 0.5; be[0] = 0.5; be[1] = -0.5; break;
    case CDKF_SOLVER_MIDPOINT: stages = 2; a[1][0] = 0.5; b[1] = 1; break;
    case CDKF_SOLVER_RALSTON: stages = 2; a[1][0] = 2.0 / 3; b[0] = 0.25; b[1] = 0.75; break;
    case CDKF_SOLVER_BOSH3:
      stages = 3; order = 3; fsal = 1;
      a[1][0] = 0.5; a[2][1] = 0.75; b[0] = 2.0 / 9; b[1] = 1.0 / 3; b[2] = 4.0 / 9;
      be[0] = 2.0 / 9 - 7.0 / 24; be[1] = 1.0 / 3 - 1.0 / 4; be[2] = 4.0 / 9 - 1.0 / 3; be[6] = -1.0 / 8;  // FSAL weight in slot 6
      break;
    default: return false;
  }
  tb.stages = stages;
  for (int i = 0; i < 6; ++i) {
    tb.b[i] = R(b[i]);
    for (int j = 0; j < 5; ++j) tb.a[i][j] = R(a[i][j]);
  }
  for (int i = 0; i < 7; ++i) tb.berr[i] = R(be[i]);
  tb.fsal = fsal;
  tb.adaptive = o->adaptive ? 1 : 0;
  if (tb.adaptive && order == 0) return false;  // no embedded error estimate
  const double ord = order ? order : 1;
  tb.rtol = R(o->rtol);
  tb.atol = R(o->atol);
  tb.c1 = R((o->pid_i + o->pid_p + o->pid_d) / ord);
  tb.c2 = R(-(o->pid_p + 2 * o->pid_d) / ord);
  tb.c3 = R(o->pid_d / ord);
  tb.dtmin = R(o->dtmin);
  tb.dtmax = (o->dtmax == 0.0) ? R(HUGE_VAL) : R(o->dtmax);  // (0: no bound -- a zero-initialised cdkf_opts keeps working)
  tb.safety = R(o->pid_safety > 0 ? o->pid_safety : 0.9);  // (0: diffrax's defaults, as for dtmax)
  tb.fmin = R(o->pid_factormin > 0 ? o->pid_factormin : 0.2);
  tb.fmax = R(o->pid_factormax > 0 ? o->pid_factormax : 10.0);
  return true;
}

// Element (n, k, i) of an array lives at n * sn + k * sk + i * si.  opts.layout places the OUTPUT arrays (means,
// covariances); the inputs t and y follow opts.layout_in (CDKF_LAYOUT_SAME = like the outputs).  w = components per row.
struct ArrayStrides {
  long sn, sk, si;
};
inline ArrayStrides layout_strides(int layout, long N, long T, long w) {
  if (layout == CDKF_LAYOUT_TCN) return {1, N * w, N};
  if (layout == CDKF_LAYOUT_TN) return {w, N * w, 1};
  return {T * w, w, 1};
}
struct SweepStrides {
  long t_sn, t_sk, y_sn, y_sk, y_si, m_sn, m_sk, m_si, P_sn, P_sk, P_si;
};
inline SweepStrides sweep_strides(const cdkf_opts* o, long N, long T, long D, long M, bool no_y) {
  const int lin = (o->layout_in == CDKF_LAYOUT_SAME) ? o->layout : o->layout_in;
  const ArrayStrides ts = layout_strides(lin, N, T, 1), ys = layout_strides(lin, N, T, M);
  const ArrayStrides ms = layout_strides(o->layout, N, T, D), Ps = layout_strides(o->layout, N, T, D * D);
  SweepStrides s;
  s.t_sn = o->t_shared ? 0 : ts.sn;
  s.t_sk = o->t_shared ? 1 : ts.sk;
  s.y_sn = ys.sn; s.y_sk = ys.sk; s.y_si = ys.si;
  s.m_sn = ms.sn; s.m_sk = ms.sk; s.m_si = ms.si;
  s.P_sn = Ps.sn; s.P_sk = Ps.sk; s.P_si = Ps.si;
  if (no_y) s.y_sn = s.y_sk = s.y_si = 0;  // forecast mode: every 'observation' load hits one valid address
  return s;
}

int reg_lanes_per_wave(int64_t N);
// wavefront grouping of `units` lanes-worth of work (reg_unit_index): lanes per wavefront, log2 of the groups per 128-byte
// line of `bytes_per_real`-sized components, number of blocks (a multiple of 8 << xcd_shift; surplus blocks idle)
struct RegGrouping {
  int lanes, xcd_shift;
  unsigned blocks;
};
inline RegGrouping reg_grouping(int64_t units, int bytes_per_real) {
  RegGrouping g;
  g.lanes = reg_lanes_per_wave(units);
  g.xcd_shift = 0;
  while ((g.lanes << g.xcd_shift) * bytes_per_real < 128) ++g.xcd_shift;
  const int64_t groups = (units + g.lanes - 1) / g.lanes, round = (int64_t)8 << g.xcd_shift;
  g.blocks = (unsigned)(g.xcd_shift ? (groups + round - 1) / round * round : groups);
  return g;
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
  a.solver = o->solver;
  {
    const RegGrouping g = reg_grouping(N, (int)sizeof(R));
    a.lanes = g.lanes;
    a.xcd_shift = g.xcd_shift;
  }
  fill_rk_tab<R>(o, a.rk);  // opts.solver / adaptive were validated by check_common
  a.N = N;
  a.T = T;
  const bool no_y = (y == nullptr);
  {
    const SweepStrides st = sweep_strides(o, N, T, D, M, no_y);
    a.t_sn = st.t_sn; a.t_sk = st.t_sk; a.y_sn = st.y_sn; a.y_sk = st.y_sk; a.y_si = st.y_si;
    a.m_sn = st.m_sn; a.m_sk = st.m_sk; a.m_si = st.m_si; a.P_sn = st.P_sn; a.P_sk = st.P_sk; a.P_si = st.P_si;
  }
  a.t = t;
  a.y = y ? y : t;  // forecast mode ignores the observations; keep the prefetch loads on valid memory
  a.ll = ll;
  a.fm = fm;
  a.fP = fP;
  a.pm = pm;
  a.pP = pP;
  a.status = status;
  a.u = nullptr;  // (the registry drifts ignore inputs and time, as the reference's own do: cdnlgssm_utils.py:50-83)
  a.u_sn = a.u_sk = a.u_si = 0;
}


}  // namespace cdkf
