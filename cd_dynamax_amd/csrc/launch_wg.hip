// launch_wg.hip -- workgroup-per-trajectory kernels: parameter block upload, LDS sizing, launch.
#include "cdkf_launch.h"
#include "cdkf_wg2_kernels.h"
#include "cdkf_wg_launch.h"
#include "cdkf_wave8_kernels.h"  // (W8Off / kCkStep for the workspace sizes; the kernel itself is built in launch_w8.hip)
#include "cdkf_rts1_kernels.h"

namespace cdkf {

static int wg_hsel(const cdkf_model* mdl) { return emission_is_selection(mdl) ? 1 : 0; }

template <typename R>
static size_t wg_lds_bytes(const cdkf_model* mdl, bool smoother, bool ukf = false) {
  const int q = mdl->state_dim > mdl->emission_dim ? mdl->state_dim : mdl->emission_dim;
  const int lq = ((q + 3) & ~3) + 1;
  const WgPlan plan = wg_plan(mdl->drift_kind, mdl->state_dim, mdl->hidden1, mdl->hidden2, wg_hsel(mdl), smoother, ukf);
  return sizeof(R) * (size_t)wg_lds_reals(plan, q, lq) + 64;
}

static long expected_theta(const cdkf_model* mdl) {
  const long d = mdl->state_dim, h1 = mdl->hidden1, h2 = mdl->hidden2;
  switch (mdl->drift_kind) {
    case CDKF_DRIFT_LINEAR: return d * d + d;
    case CDKF_DRIFT_LORENZ63: return 3;
    case CDKF_DRIFT_LORENZ96: return 1;
    case CDKF_DRIFT_MLP_TANH: return h1 * d + h1 + h2 * h1 + h2 + d * h2 + d;
    default: return custom_ntheta(mdl->drift_kind, mdl->state_dim);  // a drift given as source (launch_custom.hip), or -1
  }
}

bool wg_shape_available(const cdkf_model* mdl, int bytes_per_real) {
  const int d = mdl->state_dim, m = mdl->emission_dim;
  if (d < 1 || m < 1 || d > 64 || m > 64) return false;
  if (expected_theta(mdl) < 0 || expected_theta(mdl) != mdl->n_theta) return false;
  if (mdl->drift_kind == CDKF_DRIFT_LORENZ63 && d != 3) return false;
  if (mdl->drift_kind == CDKF_DRIFT_LORENZ96 && d < 4) return false;
  if (mdl->drift_kind >= CDKF_DRIFT_CUSTOM_BASE && mdl->emission_kind != 0) return false;
  const size_t lds = bytes_per_real == 8 ? wg_lds_bytes<double>(mdl, true) : wg_lds_bytes<float>(mdl, true);
  return lds <= kLdsLimit - 256;
}
bool custom_wg_fits(const cdkf_model* mdl) { return wg_shape_available(mdl, 4); }

// Parameter blocks (up to ~70 KB for d = 40) go through a small ring of persistent device buffers, each paired
// with a pinned host staging buffer and an event recorded behind the kernels that read it.  The asynchronous
// _dev entry points therefore neither allocate nor synchronise in the steady state.  (hipMallocAsync +
// hipMemcpyAsync from pageable memory was tried first and delivered stale parameter blocks on ROCm 7.2.)
static constexpr int kParamSlots = 8;
static ParamSlot g_slots[kParamSlots];
static int g_next_slot = 0;
static std::mutex g_slot_mutex;

int param_pool_acquire(size_t bytes, ParamSlot** out) {
  std::lock_guard<std::mutex> lock(g_slot_mutex);
  int dev = 0;
  CDKF_HIP_CHECK(hipGetDevice(&dev));
  ParamSlot& s = g_slots[g_next_slot];
  g_next_slot = (g_next_slot + 1) % kParamSlots;
  if (s.in_flight) {
    CDKF_HIP_CHECK(hipEventSynchronize(s.done));
    s.in_flight = false;
  }
  if (s.cap < bytes || s.device != dev) {
    if (s.dev) (void)hipFree(s.dev);
    if (s.host) (void)hipHostFree(s.host);
    if (s.done) (void)hipEventDestroy(s.done);
    s = ParamSlot();
    const size_t cap = bytes < 4096 ? 4096 : bytes;
    CDKF_HIP_CHECK(hipMalloc(&s.dev, cap));
    CDKF_HIP_CHECK(hipHostMalloc(&s.host, cap, hipHostMallocDefault));
    CDKF_HIP_CHECK(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    s.cap = cap;
    s.device = dev;
  }
  *out = &s;
  return CDKF_OK;
}

int param_pool_release(ParamSlot* s, hipStream_t stream) {
  std::lock_guard<std::mutex> lock(g_slot_mutex);
  CDKF_HIP_CHECK(hipEventRecord(s->done, stream));
  s->in_flight = true;
  return CDKF_OK;
}

// Builds the parameter block in the compute type (h) and every field of the kernel argument but the pointers: host arithmetic only
// (cdkf_debug_wg_args hands the result to the CPU-sanitizer build of the same kernels, tests/test_hostsim.py).
template <typename R>
static int wg_fill(WgArgs<R>& a, std::vector<R>& h, const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T) {
  const int d = mdl->state_dim, m = mdl->emission_dim;
  if (!wg_shape_available(mdl, sizeof(R))) {
    set_error("no kernel for drift_kind=%d state_dim=%d emission_dim=%d n_theta=%lld (fp%d)", mdl->drift_kind, d, m,
              (long long)mdl->n_theta, (int)sizeof(R) * 8);
    return CDKF_EUNSUPPORTED;
  }
  h.clear();
  auto push = [&](const double* src, long n) {
    const long off = (long)h.size();
    for (long i = 0; i < n; ++i) h.push_back(R(src[i]));
    return off;
  };
  a.o_theta = push(mdl->theta, mdl->n_theta);
  std::vector<R> full(d * d), packed(d * (d + 1) / 2);
  auto push_lql = [&](double scale) {
    lql_packed<R>(mdl->L, mdl->Qc, d, scale, packed.data());
    int e = 0;
    for (int i = 0; i < d; ++i)
      for (int j = i; j < d; ++j) full[i * d + j] = full[j * d + i] = packed[e++];
    const long off = (long)h.size();
    h.insert(h.end(), full.begin(), full.end());
    return off;
  };
  a.o_LQL = push_lql(1.0);
  a.o_LQLz = push_lql(o->cov_rescaling);
  a.o_H = push(mdl->H, (long)m * d);
  a.o_hb = push(mdl->h_bias, m);
  a.o_R = push(mdl->R, (long)m * m);
  a.o_m0 = push(mdl->m0, d);
  a.o_P0 = push(mdl->P0, (long)d * d);
  a.o_w2pad = -1;
  a.u = nullptr;  // (set by the launchers that have inputs: drifts given as source; the registry drifts ignore them)
  a.u_sn = a.u_sk = a.u_si = 0;
  a.du = 0;
  a.ctx_uoff = 0;
  a.ctx_t = a.ctx_tend = R(0);
  a.ctx_rev = 0;
  if (mdl->drift_kind >= CDKF_DRIFT_CUSTOM_BASE && mdl->input_dim > 0) {
    const int lin = (o->layout_in == CDKF_LAYOUT_SAME) ? o->layout : o->layout_in;
    const ArrayStrides us = layout_strides(lin, N, T, mdl->input_dim);
    a.u = (const R*)o->inputs;  // (device memory at this level)
    a.u_sn = us.sn; a.u_sk = us.sk; a.u_si = us.si;
    a.du = mdl->input_dim;
  }
  a.r_diag = 1;
  for (int r = 0; r < m; ++r)
    for (int c = 0; c < m; ++c)
      if (r == c ? !(mdl->R[r * m + c] >= 1e-2) : mdl->R[r * m + c] != 0.0) a.r_diag = 0;
  if (mdl->drift_kind == CDKF_DRIFT_MLP_TANH && mdl->hidden1 <= 64 && mdl->hidden2 <= 64 && d <= 8) {
    const int h1 = mdl->hidden1, h2 = mdl->hidden2;
    const double* W2 = mdl->theta + (long)h1 * d + h1;
    a.o_w2pad = (long)h.size();
    h.resize(h.size() + 64 * 64, R(0));
    for (int p = 0; p < h2; ++p)
      for (int q = 0; q < h1; ++q) h[a.o_w2pad + p * 64 + q] = R(W2[p * h1 + q]);
  }
  a.kind = mdl->drift_kind;
  a.d = d;
  a.m = m;
  a.h1 = mdl->hidden1;
  a.h2 = mdl->hidden2;
  a.hsel = wg_hsel(mdl);
  a.forecast = o->forecast;
  a.ukf = 0;
  {  // UKF weights in the compute type (inference_ukf.py:42, 63-89)
    const R alpha = R(o->ukf_alpha), nn = R(d);
    const R lamb = alpha * alpha * (nn + R(o->ukf_kappa)) - nn;
    a.ukf_c = std::sqrt(nn + lamb);
    a.ukf_wm0 = lamb / (nn + lamb);
    a.ukf_wi = R(1) / (R(2) * (nn + lamb));
  }
  a.q = d > m ? d : m;
  a.lq = ((a.q + 3) & ~3) + 1;  // multiple of 4 (1x4 strips stay inside a row) plus 1 (odd: column walks hit 32 different LDS banks)
  a.order = o->state_order;
  a.num_iter = o->num_iter;
  a.max_steps = (long)o->max_steps;
  fill_rk_tab<R>(o, a.rk);
  a.dt0 = R(o->dt0);
  a.dt_final = R(o->dt_final);
  a.N = N;
  a.T = T;
  {
    const SweepStrides st = sweep_strides(o, N, T, d, m, false);
    a.t_sn = st.t_sn; a.t_sk = st.t_sk; a.y_sn = st.y_sn; a.y_sk = st.y_sk; a.y_si = st.y_si;
    a.m_sn = st.m_sn; a.m_sk = st.m_sk; a.m_si = st.m_si; a.P_sn = st.P_sn; a.P_sk = st.P_sk; a.P_si = st.P_si;
  }
  return CDKF_OK;
}

// ... and uploads the block through the ring above.
template <typename R>
static int wg_prepare(WgArgs<R>& a, R** dev_block, ParamSlot** slot_out, const cdkf_model* mdl, const cdkf_opts* o,
                      int64_t N, int64_t T, hipStream_t stream) {
  std::vector<R> h;
  const int frc = wg_fill(a, h, mdl, o, N, T);
  if (frc) return frc;
  ParamSlot* slot = nullptr;
  int prc = param_pool_acquire(h.size() * sizeof(R), &slot);
  if (prc) return prc;
  *slot_out = slot;  // the caller's lease covers every exit from here on
  std::memcpy(slot->host, h.data(), h.size() * sizeof(R));
  CDKF_HIP_CHECK(hipMemcpyAsync(slot->dev, slot->host, h.size() * sizeof(R), hipMemcpyHostToDevice, stream));
  *dev_block = (R*)slot->dev;
  a.par = *dev_block;
  return CDKF_OK;
}

static int wg_threads(int d) {
  if (const char* e = getenv("CDKF_WG_THREADS")) return atoi(e);  // debugging aid
  // d >= 32: 512 threads (two wavefronts per SIMD hide each other's LDS latency, and 4 owned entries per thread
  // keep the six RK slopes inside the 256 architectural VGPRs); d = 23 .. 31: 256 threads for the same reason -- with 128 threads a
  // thread owned up to 8 entries, and that instantiation (EPT = 8: 48 slopes per thread in fp64) both crawled and returned NaN
  // (scripts/dbg_wg28.py: Lorenz-96 d = 24, 28)
  return d * d >= 1024 ? 512 : (d * d >= 512 ? 256 : (d * d >= 256 ? 128 : 64));
}
// the MLP's hidden layers give every phase of the right-hand side >= h1*d independent entries
static int wg_threads(const cdkf_model* mdl) {
  const int t = wg_threads(mdl->state_dim);
  return (mdl->drift_kind == CDKF_DRIFT_MLP_TANH && t < 256) ? 256 : t;
}

// entries of the d x d covariance owned by one thread
static int wg_ept(int d, int threads) {
  const int need = (d * d + threads - 1) / threads;
  for (int e : {1, 2, 4, 8, 16})
    if (need <= e) return e;
  return -1;
}

// A drift given as source: at most 256 threads.  Under hipRTC (ROCm 7.2) the device FUNCTIONS these kernels call (s_swappc: wg_ekf_update,
// wg_cholesky2, ...) do not inherit the kernel's __launch_bounds__(512) register budget as they do under hipcc -- the kernel comes out with
// 256 + 128 registers, a workgroup of eight such wavefronts does not fit a CU and the launch dies with HSA_STATUS_ERROR_INVALID_ISA
// (-amdgpu-internalize-symbols, --gpu-max-threads-per-block: no difference; -amdgpu-function-calls=false: 256 registers, but the
// float32 d = 12 instantiation then faulted on memory).  Four wavefronts, one per SIMD, may have them.
static int wg_threads_custom(int d) {
  const int t = wg_threads(d);
  return t > 256 ? 256 : t;
}

// what launch_custom.hip needs to compile a drift into these kernels ahead of a launch (cdkf_custom_drift_compile): a dense emission
// matrix is assumed (the larger plan); nonzero: does not fit
int custom_wg_geometry(int kind, int d, int m, int bytes_per_real, bool ukf, int* ept, int* threads, size_t* lds_f, size_t* lds_s) {
  cdkf_model mdl{};
  mdl.drift_kind = kind;
  mdl.state_dim = d;
  mdl.emission_dim = m;
  if (d < 1 || m < 1 || d > 64 || m > 64) return 1;
  *threads = wg_threads_custom(d);
  *ept = wg_ept(d, *threads);
  const int q = d > m ? d : m, lq = ((q + 3) & ~3) + 1;
  const size_t s = (size_t)bytes_per_real;
  *lds_f = s * (size_t)wg_lds_reals(wg_plan(kind, d, 0, 0, 0, false, ukf), q, lq) + 64;
  *lds_s = s * (size_t)wg_lds_reals(wg_plan(kind, d, 0, 0, 0, true, false), q, lq) + 64;
  return (*ept < 0 || *lds_f > kLdsLimit - 256 || *lds_s > kLdsLimit - 256) ? 1 : 0;
}

template <typename R>
static int launch_wg_dispatch(const WgArgs<R>& a, const cdkf_model* mdl, bool smoother, hipStream_t stream, bool filter = true) {
  const int threads = wg_threads(mdl);
  const size_t lds_f = wg_lds_bytes<R>(mdl, false, a.ukf != 0), lds_s = wg_lds_bytes<R>(mdl, true);
  if (mdl->drift_kind >= CDKF_DRIFT_CUSTOM_BASE) {  // compiled at run time with the drift's source
    const int tc = wg_threads_custom(a.d), ept = wg_ept(a.d, tc);
    if (ept < 0) {
      set_error("state_dim %d too large for the workgroup kernels", a.d);
      return CDKF_EUNSUPPORTED;
    }
    return launch_custom_wg<R>(a, ept, filter, smoother, tc, lds_f, lds_s, stream);
  }
  switch (wg_ept(a.d, threads)) {
    case 1: return launch_wg_pair<R, 1>(a, filter, smoother, threads, lds_f, lds_s, stream);
    case 2: return launch_wg_pair<R, 2>(a, filter, smoother, threads, lds_f, lds_s, stream);
    case 4: return launch_wg_pair<R, 4>(a, filter, smoother, threads, lds_f, lds_s, stream);
    case 8:
    case 16: return launch_wg_pair_wide<R>(a, wg_ept(a.d, threads), filter, smoother, threads, lds_f, lds_s, stream);
    default: set_error("state_dim %d too large for the workgroup kernels", a.d); return CDKF_EUNSUPPORTED;
  }
}

// What launch_wg_dispatch would launch for this model -- the argument struct (pointers null), the parameter block, and the geometry
// {entries per thread, threads, LDS bytes, sizeof(WgArgs<R>)} -- without touching the GPU (cdkf_debug_wg_args).
template <typename R>
static int debug_wg_args_t(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, int ukf, int smoother, void* args_out,
                           int64_t args_cap, void* blob_out, int64_t blob_cap, int64_t* geom) {
  WgArgs<R> a{};
  std::vector<R> h;
  cdkf_opts of = *o;
  if (smoother == 1) of.num_iter = 1;
  const int rc = wg_fill(a, h, mdl, &of, N, T);
  if (rc) return rc;
  if (ukf) {
    a.ukf = 1;
    a.num_iter = 1;
  }
  const bool custom = mdl->drift_kind >= CDKF_DRIFT_CUSTOM_BASE;
  const int threads = custom ? wg_threads_custom(a.d) : wg_threads(mdl);
  geom[0] = wg_ept(a.d, threads);
  geom[1] = threads;
  geom[2] = (int64_t)((wg_lds_bytes<R>(mdl, smoother == 1, ukf != 0) + 15) & ~size_t(15));
  geom[3] = (int64_t)sizeof(WgArgs<R>);
  if (smoother == 2) {  // the reverse sweep (ekf_adjoint_wg_kernel): entries per thread, 256 threads, its LDS plan, scratch reals per trajectory, cap
    int ne = 0;
    size_t lds = 0;
    if (custom_awg_geometry(a.d, a.m, (int)sizeof(R), &ne, &lds)) {
      set_error("cdkf_debug_wg_args: state_dim %d / emission_dim %d do not fit the reverse sweep's LDS plan", a.d, a.m);
      return CDKF_EUNSUPPORTED;
    }
    geom[0] = ne;
    geom[1] = 256;
    geom[2] = (int64_t)((lds + 15) & ~size_t(15));
    geom[4] = adjoint_wg_scratch_reals(a.d, 8);
    geom[5] = 8;
  }
  if ((int64_t)sizeof(WgArgs<R>) > args_cap || (int64_t)(h.size() * sizeof(R)) > blob_cap) {
    set_error("cdkf_debug_wg_args: buffers too small (%lld, %lld bytes needed)", (long long)sizeof(WgArgs<R>), (long long)(h.size() * sizeof(R)));
    return CDKF_EINVAL;
  }
  std::memcpy(args_out, &a, sizeof(WgArgs<R>));
  std::memcpy(blob_out, h.data(), h.size() * sizeof(R));
  return (int)h.size();
}
int debug_wg_args(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, int bytes_per_real, int ukf, int smoother, void* args_out,
                  int64_t args_cap, void* blob_out, int64_t blob_cap, int64_t* geom) {
  if (!mdl || !o || !args_out || !blob_out || !geom || (bytes_per_real != 4 && bytes_per_real != 8)) {
    set_error("cdkf_debug_wg_args: bad arguments");
    return CDKF_EINVAL;
  }
  return bytes_per_real == 8 ? debug_wg_args_t<double>(mdl, o, N, T, ukf, smoother, args_out, args_cap, blob_out, blob_cap, geom)
                             : debug_wg_args_t<float>(mdl, o, N, T, ukf, smoother, args_out, args_cap, blob_out, blob_cap, geom);
}

// wavefront-per-trajectory kernel: state and emission dimensions up to 8, MLP hidden layers up to 64
static bool wave8_shape(const cdkf_model* mdl) {
  if (env_flag("CDKF_NO_WAVE8")) return false;  // debugging aid: force the workgroup kernels
  if (mdl->drift_kind >= CDKF_DRIFT_CUSTOM_BASE) return false;  // (a drift given as source is compiled into the workgroup kernels)
  if (mdl->state_dim > 8 || mdl->emission_dim > 8) return false;
  if (mdl->drift_kind == CDKF_DRIFT_MLP_TANH && (mdl->hidden1 > 64 || mdl->hidden2 > 64)) return false;
  return true;
}

template <typename R>
int launch_ekf_filter_wg(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll,
                         R* fm, R* fP, R* pm, R* pP, int32_t* status, hipStream_t stream) {
  WgArgs<R> a{};
  R* blk = nullptr;
  ParamLease lease(stream);
  int rc = wg_prepare(a, &blk, &lease.slot, mdl, o, N, T, stream);
  if (rc) return rc;
  a.t = t; a.y = y; a.ll = ll; a.fm = fm; a.fP = fP; a.pm = pm; a.pP = pP; a.status = status;
  if (!y) { a.y = t; a.y_sn = a.y_sk = a.y_si = 0; }  // forecast mode: observations are ignored
  if (wave40_shape(mdl, o) && y)
    rc = launch_wave40<R>(a, stream);
  else
    rc = (wave8_shape(mdl) && o->solver == CDKF_SOLVER_DOPRI5 && !o->adaptive) ? launch_wave8<R>(a, stream)  // (wave8: fixed-step Dopri5)
                                                               : launch_wg_dispatch<R>(a, mdl, false, stream);
  const int rc2 = lease.release();
  return rc ? rc : rc2;
}

template <typename R>
int launch_ekf_smoother_wg(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y,
                           R* ll, R* fm, R* fP, R* sm, R* sP, int32_t* status, hipStream_t stream) {
  cdkf_opts of = *o;
  of.num_iter = 1;
  WgArgs<R> a{};
  R* blk = nullptr;
  ParamLease lease(stream);
  int rc = wg_prepare(a, &blk, &lease.slot, mdl, &of, N, T, stream);
  if (rc) return rc;
  a.t = t; a.y = y; a.ll = ll; a.fm = fm; a.fP = fP; a.pm = nullptr; a.pP = nullptr; a.sm = sm; a.sP = sP; a.status = status;
  if (wave8_shape(mdl) && o->solver == CDKF_SOLVER_DOPRI5 && !o->adaptive) {  // state_dim <= 8: both passes on the wavefront-per-trajectory kernels
    rc = launch_wave8<R>(a, stream);
    if (!rc)
      rc = (mdl->drift_kind == CDKF_DRIFT_MLP_TANH) ? launch_adjoint_kernel<R, true, true>(a, nullptr, nullptr, stream)
                                                     : launch_adjoint_kernel<R, false, true>(a, nullptr, nullptr, stream);
  } else if (wave40_shape(mdl, &of)) {  // d = 40: both passes on the wavefront-per-trajectory sweeps
    rc = launch_wave40<R>(a, stream);
    if (!rc)
      rc = env_flag("CDKF_WG_BACKWARD") ? launch_wg_dispatch<R>(a, mdl, true, stream, false)  // (A/B and tests: the workgroup kernel)
                                      : launch_wave40<R>(a, stream, true);
  } else {
    rc = launch_wg_dispatch<R>(a, mdl, true, stream);
  }
  const int rc2 = lease.release();
  return rc ? rc : rc2;
}

// The backward sweep ALONE, over filtered moments already in fm / fP (launch_custom.hip: the forward pass of a model whose EMISSION is
// given as source above six dimensions runs on the tangent kernels' value mode; the backward sweep never evaluates the emission --
// inference_ekf.py:363-448 -- so `mdl` arrives with emission_kind 0).  Declared where it is called, not in cdkf_launch.h.
template <typename R>
int launch_ekf_smoother_backward_wg(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll, R* fm,
                                    R* fP, R* sm, R* sP, int32_t* status, hipStream_t stream) {
  cdkf_opts of = *o;
  of.num_iter = 1;
  WgArgs<R> a{};
  R* blk = nullptr;
  ParamLease lease(stream);
  int rc = wg_prepare(a, &blk, &lease.slot, mdl, &of, N, T, stream);
  if (rc) return rc;
  a.t = t; a.y = y; a.ll = ll; a.fm = fm; a.fP = fP; a.pm = nullptr; a.pP = nullptr; a.sm = sm; a.sP = sP; a.status = status;
  rc = launch_wg_dispatch<R>(a, mdl, true, stream, false);
  const int rc2 = lease.release();
  return rc ? rc : rc2;
}

template <typename R>
int launch_ukf_filter_wg(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll,
                         R* fm, R* fP, R* pm, R* pP, int32_t* status, hipStream_t stream) {
  if (wg_lds_bytes<R>(mdl, false, true) > kLdsLimit - 256) {
    set_error("UKF: state_dim %d does not fit the workgroup kernel's LDS plan", mdl->state_dim);
    return CDKF_EUNSUPPORTED;
  }
  WgArgs<R> a{};
  R* blk = nullptr;
  ParamLease lease(stream);
  int rc = wg_prepare(a, &blk, &lease.slot, mdl, o, N, T, stream);
  if (rc) return rc;
  a.ukf = 1;
  a.num_iter = 1;
  a.t = t; a.y = y; a.ll = ll; a.fm = fm; a.fP = fP; a.pm = pm; a.pP = pP; a.status = status;
  if (!y) { a.y = t; a.y_sn = a.y_sk = a.y_si = 0; }
  rc = launch_wg_dispatch<R>(a, mdl, false, stream);
  const int rc2 = lease.release();
  return rc ? rc : rc2;
}

// ---- reverse-sweep gradient (cdkf_adjoint_kernels.h) ------------------------------------------------------------------
// The forward sweep's moments (the adjoint's checkpoints) live in a grow-only per-process workspace; an event recorded
// behind the backward kernel orders its reuse across streams.
struct AdjWorkspace {
  void* p = nullptr;
  size_t cap = 0;
  int device = -1;
  hipEvent_t done = nullptr;
  bool in_flight = false;
};
static AdjWorkspace g_adj_ws;
static std::mutex g_adj_mutex;

// grow (never shrink) the workspace to `bytes` on the current device; a buffer still in use by an earlier launch is waited for
// on the device (same size) or on the host (before it is replaced).  Call with g_adj_mutex held.
static int workspace_reserve(AdjWorkspace& ws, size_t bytes, hipStream_t stream) {
  int dev = 0;
  CDKF_HIP_CHECK(hipGetDevice(&dev));
  if (ws.cap < bytes || ws.device != dev) {
    if (ws.in_flight) CDKF_HIP_CHECK(hipEventSynchronize(ws.done));
    if (ws.p) (void)hipFree(ws.p);
    if (ws.done) (void)hipEventDestroy(ws.done);
    ws = AdjWorkspace();
    CDKF_HIP_CHECK(hipMalloc(&ws.p, bytes));
    CDKF_HIP_CHECK(hipEventCreateWithFlags(&ws.done, hipEventDisableTiming));
    ws.cap = bytes;
    ws.device = dev;
  } else if (ws.in_flight) {
    CDKF_HIP_CHECK(hipStreamWaitEvent(stream, ws.done, 0));
  }
  return CDKF_OK;
}

GradWorkspaceLease::GradWorkspaceLease() { g_adj_mutex.lock(); }
GradWorkspaceLease::~GradWorkspaceLease() { g_adj_mutex.unlock(); }
int GradWorkspaceLease::reserve(size_t bytes, hipStream_t stream, void** p) {
  if (int rc = workspace_reserve(g_adj_ws, bytes, stream)) return rc;
  *p = g_adj_ws.p;
  return CDKF_OK;
}
int GradWorkspaceLease::done(hipStream_t stream) {
  CDKF_HIP_CHECK(hipEventRecord(g_adj_ws.done, stream));
  g_adj_ws.in_flight = true;
  return CDKF_OK;
}

// cdkf_release_workspace: wait for the last launch that uses the reverse sweeps' workspace and give the memory back
int release_grad_workspace() {
  std::lock_guard<std::mutex> lock(g_adj_mutex);
  AdjWorkspace& ws = g_adj_ws;
  if (ws.in_flight && ws.done) CDKF_HIP_CHECK(hipEventSynchronize(ws.done));
  if (ws.p) {
    int dev = 0;
    CDKF_HIP_CHECK(hipGetDevice(&dev));
    if (dev != ws.device) CDKF_HIP_CHECK(hipSetDevice(ws.device));
    (void)hipFree(ws.p);
    if (ws.done) (void)hipEventDestroy(ws.done);
    if (dev != ws.device) CDKF_HIP_CHECK(hipSetDevice(dev));
  }
  ws = AdjWorkspace();
  return CDKF_OK;
}

// beyond the wavefront kernel's shapes: the workgroup-per-trajectory reverse sweep -- Lorenz-96 and linear drifts (for both the mean's
// second-order term vanishes), fixed or adaptive steps, as far as its LDS plan goes in fp64 (launch_adjwg.hip)
static bool adjoint_wg_shape(const cdkf_model* mdl, const cdkf_opts* o) {
  if (mdl->drift_kind >= CDKF_DRIFT_CUSTOM_BASE) {  // a drift given as source: its derivatives by dual numbers inside the sweep
    if (!custom_adjoint_available(mdl, o)) return false;
  } else if (mdl->drift_kind == CDKF_DRIFT_MLP_TANH) {  // (round 4) the network's reverse pass on the workgroup's threads, both state orders
    return mdl->emission_kind == 0 && wg_shape_available(mdl, 4) && mdl->state_dim <= 42 &&
           adjoint_wg_fits_mlp(mdl->state_dim, mdl->emission_dim, mdl->hidden1, mdl->hidden2, 4);
  } else if (mdl->drift_kind != CDKF_DRIFT_LORENZ96 && mdl->drift_kind != CDKF_DRIFT_LINEAR) {
    return false;
  }
  if (mdl->emission_kind != 0) return false;
  // (the gate is precision-agnostic: what the float32 kernels take; the launch refuses -- CDKF_EUNSUPPORTED, with the numbers -- a
  //  float64 call whose nine matrices do not fit)
  return wg_shape_available(mdl, 4) && adjoint_wg_fits(mdl->state_dim, mdl->emission_dim, 4);
}
bool adjoint_shape_available(const cdkf_model* mdl, const cdkf_opts* o) {
  if (o->num_iter < 1 || o->num_iter > 8 || o->forecast || o->state_order == CDKF_ORDER_ZEROTH) return false;
  // (num_iter > 1 -- the iterated update of inference_ekf.py:153-199 -- is reversed iteration by iteration in both reverse sweeps,
  //  the inputs of iterations 1 .. n-1 recomputed from the predicted moments and parked in LDS -- on the wavefront sweep, d, m <= 8)
  if (wave8_shape(mdl)) return wg_shape_available(mdl, 8);
  return o->num_iter == 1 && adjoint_wg_shape(mdl, o);  // (the workgroup sweep reverses one update iteration)
}

// state_dim > 8: forward sweep on the wavefront- (Lorenz-96, launch_w40.hip) or workgroup-per-trajectory filter with all four moment
// arrays into the workspace, reverse sweep on ekf_adjoint_wg_kernel
template <typename R>
static int launch_ekf_grad_adjoint_wg(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll,
                                      R* grad, R* grad_model, int32_t* status, hipStream_t stream, bool ukf = false) {
  const bool fits = mdl->drift_kind == CDKF_DRIFT_MLP_TANH
                        ? adjoint_wg_fits_mlp(mdl->state_dim, mdl->emission_dim, mdl->hidden1, mdl->hidden2, (int)sizeof(R))
                        : adjoint_wg_fits(mdl->state_dim, mdl->emission_dim, (int)sizeof(R));
  if (!fits) {
    set_error("reverse sweep: state_dim %d / emission_dim %d do not fit its LDS plan in fp%d (nine q x q matrices, q = max of the two: "
              "q <= 43 in fp64, 62 in fp32; with an MLP drift its weights and tangent images as well)", mdl->state_dim, mdl->emission_dim,
              (int)sizeof(R) * 8);
    return CDKF_EUNSUPPORTED;
  }
  WgArgs<R> a{};
  R* blk = nullptr;
  std::lock_guard<std::mutex> lock(g_adj_mutex);
  ParamLease lease(stream);
  int rc = wg_prepare(a, &blk, &lease.slot, mdl, o, N, T, stream);
  if (rc) return rc;
  const size_t nm = (size_t)N * T * mdl->state_dim, nP = nm * mdl->state_dim;
  int cap = 8;  // step starts kept per replay chunk of an interval (CDKF_ADJ_WG_STARTS)
  if (const char* e = getenv("CDKF_ADJ_WG_STARTS")) cap = atoi(e) > 0 ? atoi(e) : 8;
  // Lorenz-96 through a selection of its components: both sweeps on one wavefront per trajectory (126 us per observation step and
  // trajectory at d = 40 against the workgroup kernel's 165, and two trajectories per CU).  CDKF_WAVE40_ADJ=0 keeps the workgroup
  // reverse sweep (A/B, tests).
  bool w40adj = wave40_shape(mdl, o) && !ukf;
  if (const char* e = getenv("CDKF_WAVE40_ADJ")) w40adj = w40adj && atoi(e) != 0;
  if (w40adj && cap > 64) cap = 64;
  const size_t nscr = (size_t)N * (size_t)(w40adj ? wave40_adjoint_scratch_reals(mdl->state_dim, cap) : adjoint_wg_scratch_reals(mdl->state_dim, cap));
  // an adaptive solve: the forward (workgroup) sweep logs the accepted step sizes of every interval (up to CDKF_ADJ_DT_CAP, default
  // 64; a longer interval raises MAX_STEPS on that trajectory) and the reverse sweep replays them
  int dtcap = 64;
  if (const char* e = getenv("CDKF_ADJ_DT_CAP")) dtcap = atoi(e) > 0 ? atoi(e) : 64;
  const size_t ndt = (o->adaptive && T > 1) ? (size_t)N * (size_t)(T - 1) * (size_t)(1 + dtcap) : 0;
  AdjWorkspace& ws = g_adj_ws;
  if (int wrc = workspace_reserve(ws, (2 * (nm + nP) + nscr + ndt) * sizeof(R), stream)) return wrc;
  R* w = (R*)ws.p;
  a.t = t; a.y = y; a.ll = ll; a.status = status;
  a.fm = w; a.fP = w + nm; a.pm = w + nm + nP; a.pP = w + 2 * nm + nP;
  a.dtlog = ndt ? w + 2 * (nm + nP) + nscr : nullptr;
  a.dtlog_cap = ndt ? dtcap : 0;
  // (unscented: the forward sweep forms the sigma points on the workgroup kernel; the reverse sweep replays the same moment equations in
  //  closed form -- equal to rounding for the quadratic drift this path admits)
  a.ukf = ukf ? 1 : 0;
  if (w40adj) {  // the moments between the two sweeps: a trajectory's matrices contiguous, whatever layout the caller's arrays have
    const long d = mdl->state_dim;
    a.m_sn = T * d; a.m_sk = d; a.m_si = 1;
    a.P_sn = T * d * d; a.P_sk = d * d; a.P_si = 1;
  }
  rc = (wave40_shape(mdl, o) && !ukf) ? launch_wave40<R>(a, stream) : launch_wg_dispatch<R>(a, mdl, false, stream);
  if (!rc)
    rc = w40adj ? launch_wave40_adjoint<R>(a, grad, grad_model, w + 2 * (nm + nP), cap, stream)
                : launch_adjoint_wg_kernel<R>(a, grad, grad_model, w + 2 * (nm + nP), cap, stream);
  CDKF_HIP_CHECK(hipEventRecord(ws.done, stream));
  ws.in_flight = true;
  const int rc2 = lease.release();
  return rc ? rc : rc2;
}

template <typename R>
static int adjoint_wave8_impl(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, const R* jumps,
                              R* ll, R* grad, R* grad_model, R* grad_jumps, R* grad_y, int32_t* status, hipStream_t stream,
                              bool ukf = false) {
  WgArgs<R> a{};
  R* blk = nullptr;
  // the workspace lock first, the parameter slot second: a caller waiting for the workspace holds no slot of the ring
  std::lock_guard<std::mutex> lock(g_adj_mutex);
  ParamLease lease(stream);
  int rc = wg_prepare(a, &blk, &lease.slot, mdl, o, N, T, stream);
  if (rc) return rc;
  const size_t nm = (size_t)N * T * mdl->state_dim, nP = nm * mdl->state_dim;
  // stage-slope checkpoints of the first smax steps of every interval (0 = re-integrate everything): two steps cover every
  // interval up to 2 dt0; CDKF_ADJ_CKPT_STEPS overrides.  For the MLP drift the forward sweep also keeps what each right-hand side
  // of an interval's FIRST step evaluates inside the network (WgArgs::ckm; 11 / 23 fields of 64 reals per stage: 34 / 71 KB per
  // interval in fp64), so that the reverse sweep's right-hand-side adjoints start from them instead of repeating the forward pass;
  // CDKF_ADJ_MLP_CKPT=0 turns that off (A/B, tests).  Checkpoints beyond CDKF_ADJ_CKPT_GB (default 128 of the 288 GB) are dropped --
  // the MLP block first, then the slopes: the sweep then recomputes instead.  cdkf_release_workspace() returns the memory.
  int smax = 2;
  if (const char* e = getenv("CDKF_ADJ_CKPT_STEPS")) smax = atoi(e);
  if (smax < 0) smax = 0;
  if (smax > kAdjCk) smax = kAdjCk;
  double cap_gb = 128.0;
  if (const char* e = getenv("CDKF_ADJ_CKPT_GB")) cap_gb = atof(e);
  size_t nck = (T > 1) ? (size_t)N * (size_t)(T - 1) * (size_t)smax * kCkStep : 0;
  if ((double)nck * sizeof(R) > cap_gb * 1e9) nck = 0;
  const int mlp_nf = (o->state_order == CDKF_ORDER_SECOND) ? kMlpCkSecond : kMlpCkFirst;
  bool mlp_ck = mdl->drift_kind == CDKF_DRIFT_MLP_TANH && nck > 0;
  if (const char* e = getenv("CDKF_ADJ_MLP_CKPT")) mlp_ck = mlp_ck && atoi(e) != 0;
  size_t nckm = mlp_ck ? (size_t)N * (size_t)(T - 1) * 6 * (size_t)mlp_nf * 64 : 0;
  if ((double)(nck + nckm) * sizeof(R) > cap_gb * 1e9) nckm = 0;
  // an adaptive solve: the forward (workgroup) sweep logs the accepted step sizes of every interval (up to CDKF_ADJ_DT_CAP, default
  // 64; a longer interval raises MAX_STEPS on that trajectory) and the reverse sweep replays them
  int dtcap = 64;
  if (const char* e = getenv("CDKF_ADJ_DT_CAP")) dtcap = atoi(e) > 0 ? atoi(e) : 64;
  const size_t ndt = (o->adaptive && T > 1) ? (size_t)N * (size_t)(T - 1) * (size_t)(1 + dtcap) : 0;
  if (o->adaptive) nck = nckm = 0;
  AdjWorkspace& ws = g_adj_ws;
  if (int wrc = workspace_reserve(ws, (2 * (nm + nP) + nck + nckm + ndt) * sizeof(R), stream)) {
    if (!nck) return wrc;
    (void)hipGetLastError();  // no room for every checkpoint: first without the network's intermediates, then without the slopes
    int wrc2 = nckm ? workspace_reserve(ws, (2 * (nm + nP) + nck + ndt) * sizeof(R), stream) : wrc;
    nckm = 0;
    if (wrc2) {
      (void)hipGetLastError();
      nck = 0;
      if (int wrc3 = workspace_reserve(ws, (2 * (nm + nP) + ndt) * sizeof(R), stream)) return wrc3;
    }
  }
  R* w = (R*)ws.p;
  a.t = t; a.y = y; a.ll = ll; a.status = status;
  a.cj = jumps; a.gcj = grad_jumps; a.gy = grad_y;
  a.ukf = ukf ? 1 : 0;  // (the unscented filter's closed-form moment equations in both wavefront sweeps: cdkf_ukf_loglik_grad_all_*)
  a.fm = w; a.fP = w + nm; a.pm = w + nm + nP; a.pP = w + 2 * nm + nP;
  // other methods / adaptive steps: forward pass on the workgroup kernel (run-time tableau), no slopes kept
  const bool dp5 = o->solver == CDKF_SOLVER_DOPRI5 && !o->adaptive;
  a.dtlog = ndt ? w + 2 * (nm + nP) + nck : nullptr;
  a.dtlog_cap = ndt ? dtcap : 0;
  a.ck = (nck && dp5) ? w + 2 * (nm + nP) : nullptr;
  a.ck_smax = (nck && dp5) ? smax : 0;
  a.ckm = (nckm && dp5 && smax >= 1) ? w + 2 * (nm + nP) + nck + ndt : nullptr;
  a.ckm_nf = a.ckm ? mlp_nf : 0;
  rc = dp5 ? launch_wave8<R>(a, stream) : launch_wg_dispatch<R>(a, mdl, false, stream);
  if (!rc)
    rc = (mdl->drift_kind == CDKF_DRIFT_MLP_TANH) ? launch_adjoint_kernel<R, true>(a, grad, grad_model, stream)
                                                   : launch_adjoint_kernel<R, false>(a, grad, grad_model, stream);
  CDKF_HIP_CHECK(hipEventRecord(ws.done, stream));
  ws.in_flight = true;
  const int rc2 = lease.release();
  return rc ? rc : rc2;
}
template <typename R>
int launch_ekf_grad_adjoint(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll,
                            R* grad, R* grad_model, int32_t* status, hipStream_t stream) {
  if (!wave8_shape(mdl)) return launch_ekf_grad_adjoint_wg<R>(mdl, o, N, T, t, y, ll, grad, grad_model, status, stream);
  return adjoint_wave8_impl<R>(mdl, o, N, T, t, y, nullptr, ll, grad, grad_model, nullptr, nullptr, status, stream);
}
template <typename R>
int launch_ekf_grad_adjoint_jumps(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y,
                                  const R* jumps, R* ll, R* grad, R* grad_model, R* grad_jumps, R* grad_y, int32_t* status,
                                  hipStream_t stream) {
  // the jumps live in the wavefront-per-trajectory sweeps only (the forward kernel adds them, the reverse kernel emits their
  // cotangents): state_dim, emission_dim <= 8, fixed-step Dormand-Prince, what the reverse sweep itself requires
  if (!wave8_shape(mdl) || !adjoint_shape_available(mdl, o) || o->solver != CDKF_SOLVER_DOPRI5 || o->adaptive) {
    set_error("loglik_grad_jumps: needs state_dim, emission_dim <= 8, the default solver (fixed-step Dopri5), num_iter 1, state_order "
              "first or second (drift_kind=%d state_dim=%d emission_dim=%d solver=%d adaptive=%d)", mdl->drift_kind, mdl->state_dim,
              mdl->emission_dim, o->solver, o->adaptive);
    return CDKF_EUNSUPPORTED;
  }
  return adjoint_wave8_impl<R>(mdl, o, N, T, t, y, jumps, ll, grad, grad_model, grad_jumps, grad_y, status, stream);
}
// The unscented filter's log-likelihood and its gradient w.r.t. EVERY leaf: for the drifts whose sigma-point sums
// collapse exactly -- Lorenz-63 and Lorenz-96 (quadratic), linear -- with a linear emission, the moment equations are the extended
// filter's plus a curvature term in the mean (oracle: ukf_curvature), and the reverse sweeps differentiate exactly those.
static bool ukf_grad_all_closed_form(const cdkf_model* mdl, const cdkf_opts* o) {
  if (mdl->drift_kind != CDKF_DRIFT_LORENZ63 && mdl->drift_kind != CDKF_DRIFT_LORENZ96 && mdl->drift_kind != CDKF_DRIFT_LINEAR) return false;
  if (mdl->emission_kind != 0 || o->solver != CDKF_SOLVER_DOPRI5 || o->adaptive) return false;
  cdkf_opts e = *o;
  e.state_order = CDKF_ORDER_FIRST;  // (the unscented filter has no state_order and no update iterations)
  e.num_iter = 1;
  return adjoint_shape_available(mdl, &e);
}
// ... and for every other model the unscented filter runs (an MLP drift, a drift or an emission given as source; round 5, VERDICT r4
// item 7): forward mode through the literal sigma-point recursion, a lane per (trajectory, leaf entry) -- launch_custom.hip,
// cdkf_ukf_tangent_kernels.h.  CDKF_UKF_GRAD_TANGENT=1 sends the closed-form models there too (A/B of the two derivations).
bool ukf_grad_all_shape_available(const cdkf_model* mdl, const cdkf_opts* o) {
  return ukf_grad_all_closed_form(mdl, o) || ukf_tangent_available(mdl, o);
}
template <typename R>
int launch_ukf_grad_all(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll, R* grad,
                        R* grad_model, int32_t* status, hipStream_t stream) {
  if (!ukf_grad_all_closed_form(mdl, o) || env_flag("CDKF_UKF_GRAD_TANGENT")) {
    if (!grad_model) {
      set_error("ukf_loglik_grad_all: grad_model must not be NULL");
      return CDKF_EINVAL;
    }
    return launch_ukf_tangent<R>(mdl, o, N, T, t, y, ll, grad, grad_model, status, stream);  // (names what it needs when it refuses)
  }
  cdkf_opts e = *o;
  e.state_order = CDKF_ORDER_FIRST;
  e.num_iter = 1;
  if (wave8_shape(mdl)) return adjoint_wave8_impl<R>(mdl, &e, N, T, t, y, nullptr, ll, grad, grad_model, nullptr, nullptr, status, stream, true);
  return launch_ekf_grad_adjoint_wg<R>(mdl, &e, N, T, t, y, ll, grad, grad_model, status, stream, true);
}
template int launch_ukf_grad_all<float>(const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const float*, const float*, float*, float*,
                                        float*, int32_t*, hipStream_t);
template int launch_ukf_grad_all<double>(const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const double*, const double*, double*,
                                         double*, double*, int32_t*, hipStream_t);
template int launch_ekf_grad_adjoint_jumps<float>(const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const float*, const float*, const float*,
                                                  float*, float*, float*, float*, float*, int32_t*, hipStream_t);
template int launch_ekf_grad_adjoint_jumps<double>(const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const double*, const double*,
                                                   const double*, double*, double*, double*, double*, double*, int32_t*, hipStream_t);
template int launch_ekf_grad_adjoint<float>(const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const float*, const float*,
                                            float*, float*, float*, int32_t*, hipStream_t);
template int launch_ekf_grad_adjoint<double>(const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const double*,
                                             const double*, double*, double*, double*, int32_t*, hipStream_t);

// ---- linear model, smoother type 1 (cdkf_rts1_kernels.h): filter sweep, pushed-forward (A, Q) per interval, discrete RTS ----
bool smoother1_shape_available(const cdkf_model* mdl) {
  if (mdl->drift_kind != CDKF_DRIFT_LINEAR || mdl->state_dim > 8 || !wg_shape_available(mdl, 8)) return false;
  const int d = mdl->state_dim;
  for (int i = 0; i < d; ++i)
    if (mdl->theta[d * d + i] != 0.0) return false;  // the reference adds the bias un-integrated; only b = 0 is the same model
  return true;
}

template <typename R>
int launch_kf_smoother1(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, const R* y, R* ll,
                        R* fm, R* fP, R* sm, R* sP, R* cross, int32_t* status, hipStream_t stream) {
  if (!smoother1_shape_available(mdl)) {
    set_error("kf_smoother1: needs a linear drift with zero bias and state_dim <= 8 (got drift_kind=%d state_dim=%d)",
              mdl->drift_kind, mdl->state_dim);
    return CDKF_EUNSUPPORTED;
  }
  if (!fm || !fP || !sm || !sP) {
    set_error("kf_smoother1: filtered and smoothed output arrays must not be NULL");
    return CDKF_EINVAL;
  }
  cdkf_opts of = *o;
  of.state_order = CDKF_ORDER_FIRST;
  of.num_iter = 1;
  of.forecast = 0;
  int rc = launch_ekf_filter<R>(mdl, &of, N, T, t, y, ll, fm, fP, nullptr, nullptr, status, stream);
  if (rc) return rc;
  if (once_per_device([] { return wg_raise_lds_cap(pushforward_wave8_kernel<R>) | wg_raise_lds_cap(rts1_wave8_kernel<R>); }))
    return CDKF_EHIP;
  WgArgs<R> a{};
  R* blk = nullptr;
  std::lock_guard<std::mutex> lock(g_adj_mutex);
  ParamLease lease(stream);
  rc = wg_prepare(a, &blk, &lease.slot, mdl, &of, N, T, stream);
  if (rc) return rc;
  a.t = t; a.y = y; a.ll = ll; a.fm = fm; a.fP = fP; a.sm = sm; a.sP = sP; a.status = status;
  const size_t d = mdl->state_dim, items = (size_t)N * (size_t)(T - 1), bytes = (items ? items : 1) * 2 * d * d * sizeof(R);
  AdjWorkspace& ws = g_adj_ws;
  if (int wrc = workspace_reserve(ws, bytes, stream)) return wrc;
  const size_t lds = sizeof(R) * (size_t)kRts1Waves * Rts1Off::end + 64;
  if (items) {
    hipLaunchKernelGGL(pushforward_wave8_kernel<R>, dim3((unsigned)((items + kRts1Waves - 1) / kRts1Waves)),
                       dim3(64 * kRts1Waves), lds, stream, a, (R*)ws.p);
    CDKF_HIP_CHECK(hipGetLastError());
  }
  hipLaunchKernelGGL(rts1_wave8_kernel<R>, dim3((unsigned)((N + kRts1Waves - 1) / kRts1Waves)), dim3(64 * kRts1Waves), lds,
                     stream, a, (const R*)ws.p, cross);
  CDKF_HIP_CHECK(hipGetLastError());
  CDKF_HIP_CHECK(hipEventRecord(ws.done, stream));
  ws.in_flight = true;
  return lease.release();
}
template int launch_kf_smoother1<float>(const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const float*, const float*,
                                        float*, float*, float*, float*, float*, float*, int32_t*, hipStream_t);
template int launch_kf_smoother1<double>(const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const double*, const double*,
                                         double*, double*, double*, double*, double*, double*, int32_t*, hipStream_t);

// (A, Q) of every observation interval by themselves (cdkf_kf_pushforward_*): AQ [N, T-1, 2, d, d] on the device
template <typename R>
int launch_kf_pushforward(const cdkf_model* mdl, const cdkf_opts* o, int64_t N, int64_t T, const R* t, R* AQ, int32_t* status,
                          hipStream_t stream) {
  if (!smoother1_shape_available(mdl)) {
    set_error("kf_pushforward: needs a linear drift with zero bias and state_dim <= 8 (got drift_kind=%d state_dim=%d)",
              mdl->drift_kind, mdl->state_dim);
    return CDKF_EUNSUPPORTED;
  }
  if (T < 2 || N < 1) return CDKF_OK;
  if (once_per_device([] { return wg_raise_lds_cap(pushforward_wave8_kernel<R>) | wg_raise_lds_cap(rts1_wave8_kernel<R>); }))
    return CDKF_EHIP;
  WgArgs<R> a{};
  R* blk = nullptr;
  ParamLease lease(stream);
  int rc = wg_prepare(a, &blk, &lease.slot, mdl, o, N, T, stream);
  if (rc) return rc;
  a.t = t;
  a.status = status;
  const size_t items = (size_t)N * (size_t)(T - 1);
  const size_t lds = sizeof(R) * (size_t)kRts1Waves * Rts1Off::end + 64;
  note_kernel("pushforward_wave8_kernel<%s>", real_name<R>());
  hipLaunchKernelGGL(pushforward_wave8_kernel<R>, dim3((unsigned)((items + kRts1Waves - 1) / kRts1Waves)), dim3(64 * kRts1Waves), lds,
                     stream, a, AQ);
  CDKF_HIP_CHECK(hipGetLastError());
  return lease.release();
}
template int launch_kf_pushforward<float>(const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const float*, float*, int32_t*,
                                          hipStream_t);
template int launch_kf_pushforward<double>(const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const double*, double*, int32_t*,
                                           hipStream_t);

#define INST(R)                                                                                                        \
  template int launch_ekf_filter_wg<R>(const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const R*, const R*, R*,  \
                                       R*, R*, R*, R*, int32_t*, hipStream_t);                                         \
  template int launch_ukf_filter_wg<R>(const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const R*, const R*, R*,  \
                                       R*, R*, R*, R*, int32_t*, hipStream_t);                                         \
  template int launch_ekf_smoother_wg<R>(const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const R*, const R*,    \
                                         R*, R*, R*, R*, R*, int32_t*, hipStream_t);                                   \
  template int launch_ekf_smoother_backward_wg<R>(const cdkf_model*, const cdkf_opts*, int64_t, int64_t, const R*,     \
                                                  const R*, R*, R*, R*, R*, R*, int32_t*, hipStream_t);
INST(float)
INST(double)

}  // namespace cdkf
