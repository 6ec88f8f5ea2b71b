// cdkf_math.h -- register-resident small dense linear algebra + Dormand-Prince stepper for the
// lane-per-trajectory ("reg") kernels.  Everything is compile-time sized and fully unrolled so that
// the mean, covariance and the six RK slopes of one trajectory live in the VGPRs of one lane.
//
// Reference arithmetic being restated (paths relative to /root/reference):
//   psd_solve / symmetrize            dynamax/utils/utils.py:202-211
//   MVN(...).log_prob                 inference_ekf.py:286 (TFP MultivariateNormalFullCovariance)
//   diffeqsolve -> Dopri5, const dt0  src/utils/diffrax_utils.py:40-165 (diffrax 0.4.0)
#pragma once
#if defined(CDKF_HOST_SIM)  // host build of the device templates for the CPU sanitizers (hostsim/cdkf_hostsim.h, tests/test_sanitize.py)
#include "hostsim/cdkf_hostsim.h"
#elif !defined(__HIPCC_RTC__)  // hipRTC (launch_custom.hip) provides the runtime declarations itself
#include <hip/hip_runtime.h>
#endif

#define CDKF_DEV __device__ __forceinline__
// an empty asm statement that makes the named vector registers opaque to the optimiser (scheduling / rematerialisation fences)
#if defined(CDKF_HOST_SIM)
#define CDKF_OPAQUE(...) ((void)0)
#else
#define CDKF_OPAQUE(...) asm volatile("" : __VA_ARGS__)
#endif

namespace cdkf {

// ---- packed upper-triangular index of a symmetric D x D matrix -----------------------------
template <int D>
CDKF_DEV constexpr int sidx(int i, int j) {
  return (i <= j) ? (i * D - (i * (i - 1)) / 2 + (j - i)) : (j * D - (j * (j - 1)) / 2 + (i - j));
}
template <int D>
struct Dims {
  static constexpr int NP = D * (D + 1) / 2;  // packed covariance entries
  static constexpr int NS = D + NP;           // ODE state: [mean, packed covariance]
};

// ---- scalar helpers ------------------------------------------------------------------------
CDKF_DEV float rfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
CDKF_DEV double rfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
CDKF_DEV float rrsqrt(float x) { return rsqrtf(x); }
CDKF_DEV double rrsqrt(double x) { return rsqrt(x); }
CDKF_DEV float rsqrt_(float x) { return sqrtf(x); }
CDKF_DEV double rsqrt_(double x) { return sqrt(x); }
CDKF_DEV float rmin(float a, float b) { return fminf(a, b); }
CDKF_DEV double rmin(double a, double b) { return fmin(a, b); }

template <typename R>
struct Tol;  // diffrax _clip_to_end tolerance
template <>
struct Tol<double> {
  static constexpr double v = 1e-10;
};
template <>
struct Tol<float> {
  static constexpr float v = 1e-6f;
};

// ---- right-hand sides that depend on TIME (a drift f(x, u, t) given as source: the reference evaluates f and jacfwd(f) at the
// solver's stage time, inference_ekf.py:95, 101-114).  A functor with `static constexpr bool kTime = true` gets `set_time(t_stage)`
// before every evaluation, t_stage = (start of the step) + c_i dt with c_i = sum_j a_ij; everything else (every built-in drift: the
// reference's own ignore t) compiles to exactly what it did before.  No <type_traits> here: hipRTC compiles this header too.
template <typename T, typename = void>
struct RhsTime {
  static constexpr bool value = false;
};
template <typename T>
struct RhsTime<T, decltype((void)T::kTime)> {
  static constexpr bool value = T::kTime;
};
template <typename R, typename Rhs>
CDKF_DEV void rhs_at(const Rhs& rhs, R t) {
  if constexpr (RhsTime<Rhs>::value) rhs.set_time(t);
}
// ... and drifts / emissions that say so: `static constexpr bool TIME` (the source mentions t), `static constexpr int DU` (> 0: it reads
// the inputs u[0 .. DU-1], u = inputs[t0_idx] held over the interval, inference_ekf.py:277-286).  Such a type has set_time(t) const and
// load_inputs(args, n, k) const writing mutable members (the argument block is handed around by const reference).
template <typename T, typename = void>
struct DriftTime {
  static constexpr bool value = false;
};
template <typename T>
struct DriftTime<T, decltype((void)T::TIME)> {
  static constexpr bool value = T::TIME;
};
template <typename T, typename = void>
struct DriftInputs {
  static constexpr bool value = false;
};
template <typename T>
struct DriftInputs<T, decltype((void)T::DU)> {
  static constexpr bool value = (T::DU > 0);
};

// ---- Cholesky of an M x M matrix given by its lower triangle (full storage S[a][b], a >= b read) --
// Returns the factor in Lc (lower) and the reciprocal pivots in inv.  A non-positive pivot gives
// NaN downstream (rsqrt of a negative), as jnp.linalg.cholesky does; `bad` is set for status.
template <typename R, int M>
CDKF_DEV void chol_lower(const R (&S)[M][M], R (&Lc)[M][M], R (&inv)[M], bool& bad) {
#pragma unroll
  for (int j = 0; j < M; ++j) {
    R s = S[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) s = rfma(-Lc[j][k], Lc[j][k], s);
    bad = bad || !(s > R(0));
    R r = rrsqrt(s);
    inv[j] = r;
    Lc[j][j] = s * r;
#pragma unroll
    for (int i = j + 1; i < M; ++i) {
      R v = S[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) v = rfma(-Lc[i][k], Lc[j][k], v);
      Lc[i][j] = v * r;
    }
  }
}

// Solve (Lc Lc^T) X = B for X, B: [M][K] (cho_solve).  In place on B.
template <typename R, int M, int K>
CDKF_DEV void chol_solve(const R (&Lc)[M][M], const R (&inv)[M], R (&B)[M][K]) {
#pragma unroll
  for (int c = 0; c < K; ++c) {
#pragma unroll
    for (int i = 0; i < M; ++i) {  // forward
      R v = B[i][c];
#pragma unroll
      for (int k = 0; k < i; ++k) v = rfma(-Lc[i][k], B[k][c], v);
      B[i][c] = v * inv[i];
    }
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {  // backward with Lc^T
      R v = B[i][c];
#pragma unroll
      for (int k = i + 1; k < M; ++k) v = rfma(-Lc[k][i], B[k][c], v);
      B[i][c] = v * inv[i];
    }
  }
}

// ---- Dormand-Prince 5(4), fixed step (diffrax Dopri5 + ConstantStepSize) ---------------------
// Stage values are formed as y0 + dt * (sum_j a_ij f_j): the increment is summed first and added to
// y0 once, the same association as diffrax's `y0 + a_lower[i] @ ks` (this is what keeps float32
// within 1-2 ulp of the reference's known-answer constants).  The ODEs of this path are autonomous,
// so FSAL's 7th evaluation equals the next step's first and is simply not computed.
template <typename R>
struct Dp5 {
  static constexpr R a21 = R(1.0 / 5.0);
  static constexpr R a31 = R(3.0 / 40.0), a32 = R(9.0 / 40.0);
  static constexpr R a41 = R(44.0 / 45.0), a42 = R(-56.0 / 15.0), a43 = R(32.0 / 9.0);
  static constexpr R a51 = R(19372.0 / 6561.0), a52 = R(-25360.0 / 2187.0), a53 = R(64448.0 / 6561.0),
                     a54 = R(-212.0 / 729.0);
  static constexpr R a61 = R(9017.0 / 3168.0), a62 = R(-355.0 / 33.0), a63 = R(46732.0 / 5247.0),
                     a64 = R(49.0 / 176.0), a65 = R(-5103.0 / 18656.0);
  static constexpr R b1 = R(35.0 / 384.0), b3 = R(500.0 / 1113.0), b4 = R(125.0 / 192.0),
                     b5 = R(-2187.0 / 6784.0), b6 = R(11.0 / 84.0);
};

// Dormand-Prince tableau as arrays (static indices after unrolling)
template <typename R>
struct Dp5T {
  static constexpr R a[6][5] = {{0, 0, 0, 0, 0},
                                {Dp5<R>::a21, 0, 0, 0, 0},
                                {Dp5<R>::a31, Dp5<R>::a32, 0, 0, 0},
                                {Dp5<R>::a41, Dp5<R>::a42, Dp5<R>::a43, 0, 0},
                                {Dp5<R>::a51, Dp5<R>::a52, Dp5<R>::a53, Dp5<R>::a54, 0},
                                {Dp5<R>::a61, Dp5<R>::a62, Dp5<R>::a63, Dp5<R>::a64, Dp5<R>::a65}};
  static constexpr R b[6] = {Dp5<R>::b1, 0, Dp5<R>::b3, Dp5<R>::b4, Dp5<R>::b5, Dp5<R>::b6};
};

// The 21 non-zero tableau entries held in VGPRs.  `pin()` hides the value from the optimiser, which
// otherwise re-materialises every 64-bit constant with two s_mov_b32 at each entry of the RK loop (42+
// scalar moves per observation step) and spills SGPRs around them; a lone wave issues one instruction
// per ~4 cycles whatever its type, so those moves cost as much as the fp64 FMAs they feed.
template <typename R>
CDKF_DEV R pin(R x) {
  CDKF_OPAQUE("+v"(x));
  return x;
}
template <typename R>
struct Dp5V {
  R a21, a31, a32, a41, a42, a43, a51, a52, a53, a54, a61, a62, a63, a64, a65, b1, b3, b4, b5, b6;
  CDKF_DEV void init() {
    using C = Dp5<R>;
    a21 = pin(C::a21);
    a31 = pin(C::a31); a32 = pin(C::a32);
    a41 = pin(C::a41); a42 = pin(C::a42); a43 = pin(C::a43);
    a51 = pin(C::a51); a52 = pin(C::a52); a53 = pin(C::a53); a54 = pin(C::a54);
    a61 = pin(C::a61); a62 = pin(C::a62); a63 = pin(C::a63); a64 = pin(C::a64); a65 = pin(C::a65);
    b1 = pin(C::b1); b3 = pin(C::b3); b4 = pin(C::b4); b5 = pin(C::b5); b6 = pin(C::b6);
  }
};

// One Dormand-Prince step.  Two associations of the stage combination:
//  * float:  y0 + dt * (sum_j a_ij f_j)  -- the increment is summed first and added to y0 once, like
//    diffrax's `y0 + a_lower[i] @ ks`; this keeps float32 within 1-2 ulp of the reference's constants.
//  * double: ((y0 + c_i1 f_1) + c_i2 f_2) + ...  with c_ij = dt a_ij formed once per step: one FMA per
//    term (20 per state entry instead of 26).  The extra roundings are at 1e-16 relative.
template <typename R, int NS, typename Rhs>
CDKF_DEV void dopri5_step(R (&y)[NS], R dt, const Rhs& rhs, const Dp5V<R>& C, R t = R(0)) {
  R k1[NS], k2[NS], k3[NS], k4[NS], k5[NS], k6[NS], ys[NS];
  if constexpr (RhsTime<Rhs>::value) {  // (time-dependent right-hand sides: the generic stage loop with the stage times; custom drifts only)
    R k[6][NS];
    constexpr R cs[6] = {R(0), R(1.0 / 5.0), R(3.0 / 10.0), R(4.0 / 5.0), R(8.0 / 9.0), R(1)};
#pragma unroll
    for (int s = 0; s < 6; ++s) {
#pragma unroll
      for (int e = 0; e < NS; ++e) {
        R acc = R(0);
#pragma unroll
        for (int j = 0; j < 5; ++j)
          if (j < s) acc = rfma(Dp5T<R>::a[s][j], k[j][e], acc);
        ys[e] = rfma(dt, acc, y[e]);
      }
      rhs.set_time(rfma(cs[s], dt, t));
      rhs(ys, k[s]);
    }
#pragma unroll
    for (int e = 0; e < NS; ++e) {
      R acc = R(0);
#pragma unroll
      for (int s = 0; s < 6; ++s) acc = rfma(Dp5T<R>::b[s], k[s][e], acc);
      y[e] = rfma(dt, acc, y[e]);
    }
    return;
  }
  if constexpr (sizeof(R) == 8) {
    const R c21 = dt * C.a21, c31 = dt * C.a31, c32 = dt * C.a32, c41 = dt * C.a41, c42 = dt * C.a42,
            c43 = dt * C.a43, c51 = dt * C.a51, c52 = dt * C.a52, c53 = dt * C.a53, c54 = dt * C.a54,
            c61 = dt * C.a61, c62 = dt * C.a62, c63 = dt * C.a63, c64 = dt * C.a64, c65 = dt * C.a65,
            d1 = dt * C.b1, d3 = dt * C.b3, d4 = dt * C.b4, d5 = dt * C.b5, d6 = dt * C.b6;
    rhs(y, k1);
#pragma unroll
    for (int e = 0; e < NS; ++e) ys[e] = rfma(c21, k1[e], y[e]);
    rhs(ys, k2);
#pragma unroll
    for (int e = 0; e < NS; ++e) ys[e] = rfma(c32, k2[e], rfma(c31, k1[e], y[e]));
    rhs(ys, k3);
#pragma unroll
    for (int e = 0; e < NS; ++e) ys[e] = rfma(c43, k3[e], rfma(c42, k2[e], rfma(c41, k1[e], y[e])));
    rhs(ys, k4);
#pragma unroll
    for (int e = 0; e < NS; ++e)
      ys[e] = rfma(c54, k4[e], rfma(c53, k3[e], rfma(c52, k2[e], rfma(c51, k1[e], y[e]))));
    rhs(ys, k5);
#pragma unroll
    for (int e = 0; e < NS; ++e)
      ys[e] = rfma(c65, k5[e], rfma(c64, k4[e], rfma(c63, k3[e], rfma(c62, k2[e], rfma(c61, k1[e], y[e])))));
    rhs(ys, k6);
#pragma unroll
    for (int e = 0; e < NS; ++e)
      y[e] = rfma(d6, k6[e], rfma(d5, k5[e], rfma(d4, k4[e], rfma(d3, k3[e], rfma(d1, k1[e], y[e])))));
  } else {
    rhs(y, k1);
#pragma unroll
    for (int e = 0; e < NS; ++e) ys[e] = rfma(dt, C.a21 * k1[e], y[e]);
    rhs(ys, k2);
#pragma unroll
    for (int e = 0; e < NS; ++e) ys[e] = rfma(dt, rfma(C.a32, k2[e], C.a31 * k1[e]), y[e]);
    rhs(ys, k3);
#pragma unroll
    for (int e = 0; e < NS; ++e) ys[e] = rfma(dt, rfma(C.a43, k3[e], rfma(C.a42, k2[e], C.a41 * k1[e])), y[e]);
    rhs(ys, k4);
#pragma unroll
    for (int e = 0; e < NS; ++e)
      ys[e] = rfma(dt, rfma(C.a54, k4[e], rfma(C.a53, k3[e], rfma(C.a52, k2[e], C.a51 * k1[e]))), y[e]);
    rhs(ys, k5);
#pragma unroll
    for (int e = 0; e < NS; ++e)
      ys[e] = rfma(dt, rfma(C.a65, k5[e], rfma(C.a64, k4[e], rfma(C.a63, k3[e], rfma(C.a62, k2[e], C.a61 * k1[e])))),
                   y[e]);
    rhs(ys, k6);
#pragma unroll
    for (int e = 0; e < NS; ++e)
      y[e] = rfma(dt, rfma(C.b6, k6[e], rfma(C.b5, k5[e], rfma(C.b4, k4[e], rfma(C.b3, k3[e], C.b1 * k1[e])))), y[e]);
  }
}

// Integrate y from t0 to t1 with the diffrax 0.4.0 loop: tprev = t0, tnext = min(t0 + dt0, t1);
// while tprev < t1: step; tprev = min(tnext, t1); tnext = clip_to_end(tnext + dt0).
// Returns true if max_steps was hit.
// ---- any explicit Runge-Kutta method with up to six stages, coefficients at run time (opts.solver; the reference forwards
// a diffrax solver object through diffeqsolve_settings, src/utils/diffrax_utils.py:40-57).  Fixed steps: only the solution
// weights b are used.  Stage loop fully unrolled with uniform guards so that the slopes keep static register indices.
template <typename R>
struct RkTab {
  int stages;
  R a[6][5];
  R b[6];
  // adaptive stepping (diffrax.PIDController around the embedded pair): error weights b_sol - b_hat, the 7th entry
  // belonging to the first-same-as-last stage f(y_new) (evaluated when fsal != 0); c1, c2, c3 = the controller's
  // exponents (i + p + d) / order, -(p + 2 d) / order, d / order
  int adaptive, fsal;
  R berr[7];
  R rtol, atol, c1, c2, c3;
  R dtmin, dtmax;  // bounds on every proposed step size (0 / +inf: none); a step taken at dtmin is kept (PIDController force_dtmin)
  R safety, fmin, fmax;  // PIDController(safety, factormin, factormax): 0.9, 0.2, 10 unless the caller says otherwise (cdkf_opts, version 110)
};

// stage time offset c_s = sum_j a_sj of a run-time tableau
template <typename R>
CDKF_DEV R rk_stage_c(const RkTab<R>& tb, int s) {
  R c = R(0);
#pragma unroll
  for (int j = 0; j < 5; ++j)
    if (j < s) c += tb.a[s][j];
  return c;
}

template <typename R, int NS, typename Rhs>
CDKF_DEV void rk_step(R (&y)[NS], R dt, const Rhs& rhs, const RkTab<R>& tb, R t = R(0)) {
  R k[6][NS], ys[NS];
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    if (s < tb.stages) {
#pragma unroll
      for (int e = 0; e < NS; ++e) {
        R acc = R(0);
#pragma unroll
        for (int j = 0; j < 5; ++j)
          if (j < s) acc = rfma(tb.a[s][j], k[j][e], acc);
        ys[e] = rfma(dt, acc, y[e]);
      }
      if constexpr (RhsTime<Rhs>::value) rhs.set_time(rfma(rk_stage_c(tb, s), dt, t));
      rhs(ys, k[s]);
    } else {
#pragma unroll
      for (int e = 0; e < NS; ++e) k[s][e] = R(0);
    }
  }
#pragma unroll
  for (int e = 0; e < NS; ++e) {
    R acc = R(0);
#pragma unroll
    for (int s = 0; s < 6; ++s) acc = rfma(tb.b[s], k[s][e], acc);
    y[e] = rfma(dt, acc, y[e]);
  }
}

CDKF_DEV float rpow(float a, float b) { return powf(a, b); }
CDKF_DEV double rpow(double a, double b) { return pow(a, b); }
CDKF_DEV float rabs(float a) { return fabsf(a); }
CDKF_DEV double rabs(double a) { return fabs(a); }
CDKF_DEV float rmax(float a, float b) { return fmaxf(a, b); }
CDKF_DEV double rmax(double a, double b) { return fmax(a, b); }

// Adaptive solve: the step-size controller of diffrax.PIDController (restated from its published algorithm; oracle:
// _diffeqsolve_adaptive).  Per step: the embedded error estimate, its RMS over ALL entries of the state pytree -- the mean
// and the FULL d x d covariance, so the packed off-diagonal entries count twice -- scaled by atol + max(|y|, |y_new|) rtol;
// accept iff < 1; next size = attempted size * clip(safety e^-c1 e1^-c2 e2^-c3, [1 if accepted else factormin, factormax]) (0.9, 0.2, 10 by default); a rejected step
// that would cross the end goes half-way.  max_steps counts attempts.  Each lane adapts on its own.
// NERR: the leading entries of y that form the reference's state pytree and enter the error norm (all of them for the
// filters; the primal half for the forward-sensitivity gradient kernels, whose tangents ride along on the primal's steps
// exactly as JAX differentiates the solve: the controller's factor is under stop_gradient).
template <typename R, int NS, int MEAN_ONLY, int NERR, typename Rhs>
CDKF_DEV bool integrate_adaptive(R (&y)[NS], R t0, R t1, R dt0, long max_steps, const Rhs& rhs, const RkTab<R>& tb) {
  // state dimension behind the packed layout NERR = D + D (D + 1) / 2
  constexpr int D =
      MEAN_ONLY ? NERR : (NERR == 2 ? 1 : NERR == 5 ? 2 : NERR == 9 ? 3 : NERR == 14 ? 4 : NERR == 20 ? 5 : NERR == 27 ? 6 : -1);
  static_assert(D > 0, "integrate_adaptive: unexpected state size");
  constexpr int COUNT = MEAN_ONLY ? NERR : D + D * D;
  R tprev = t0;
  const R dt_first = rmin(dt0, tb.dtmax);
  bool at_min = dt_first <= tb.dtmin;
  R tnext = rmin(t0 + rmax(dt_first, tb.dtmin), t1);
  R inv1 = R(1), inv2 = R(1);
  long steps = 0;
  bool capped = false;
  while (tprev < t1) {
    if (steps >= max_steps) {
      capped = true;
      break;
    }
    const R dt = tnext - tprev;
    R k[7][NS], ys[NS], yn[NS];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      if (s < tb.stages) {
#pragma unroll
        for (int e = 0; e < NS; ++e) {
          R acc = R(0);
#pragma unroll
          for (int j = 0; j < 5; ++j)
            if (j < s) acc = rfma(tb.a[s][j], k[j][e], acc);
          ys[e] = rfma(dt, acc, y[e]);
        }
        if constexpr (RhsTime<Rhs>::value) rhs.set_time(rfma(rk_stage_c(tb, s), dt, tprev));
        rhs(ys, k[s]);
      } else {
#pragma unroll
        for (int e = 0; e < NS; ++e) k[s][e] = R(0);
      }
    }
#pragma unroll
    for (int e = 0; e < NS; ++e) {
      R acc = R(0);
#pragma unroll
      for (int s = 0; s < 6; ++s) acc = rfma(tb.b[s], k[s][e], acc);
      yn[e] = rfma(dt, acc, y[e]);
    }
    if (tb.fsal) {
      if constexpr (RhsTime<Rhs>::value) rhs.set_time(tprev + dt);
      rhs(yn, k[6]);
    } else {
#pragma unroll
      for (int e = 0; e < NS; ++e) k[6][e] = R(0);
    }
    R sq = R(0);
#pragma unroll
    for (int e = 0; e < NERR; ++e) {
      R err = R(0);
#pragma unroll
      for (int s = 0; s < 7; ++s) err = rfma(tb.berr[s], k[s][e], err);
      const R sc = (dt * err) / rfma(rmax(rabs(y[e]), rabs(yn[e])), tb.rtol, tb.atol);
      // weight of the packed entry in the full pytree: mean and diagonal once, off-diagonal twice
      R w = R(1);
      if (!MEAN_ONLY && e >= D) {
        int idx = e - D, row = 0;
        while (idx >= D - row) {
          idx -= D - row;
          ++row;
        }
        w = idx == 0 ? R(1) : R(2);
      }
      sq = rfma(w * sc, sc, sq);
    }
    const R scaled = rsqrt_(sq / R(COUNT));
    const bool keep = scaled < R(1) || at_min;
    const R inv = (scaled == R(0)) ? R(__builtin_huge_val()) : R(1) / scaled;
    R factor = tb.safety * rpow(inv, tb.c1);
    if (tb.c2 != R(0)) factor *= rpow(inv1, tb.c2);
    if (tb.c3 != R(0)) factor *= rpow(inv2, tb.c3);
    // fmax / fmin semantics: a NaN estimate (a stage that left the domain because the attempted step was far too long)
    // rejects the step and shrinks it by factormin; diffrax's clip would propagate the NaN and end in its max_steps error
    factor = rmin(rmax(factor, keep ? R(1) : tb.fmin), tb.fmax);
    const R nt0 = keep ? tnext : tprev;
    R dtn = rmin(dt * factor, tb.dtmax);
    at_min = dtn <= tb.dtmin;
    dtn = rmax(dtn, tb.dtmin);
    const R nt1 = nt0 + dtn;
    if (keep) {
#pragma unroll
      for (int e = 0; e < NS; ++e) y[e] = yn[e];
      inv2 = inv1;
      inv1 = inv;
    }
    tprev = rmin(nt0, t1);
    tnext = (nt1 > t1 - Tol<R>::v) ? (keep ? t1 : rfma(R(0.5), t1 - tprev, tprev)) : nt1;
    ++steps;
  }
  return capped;
}

// CDKF_UNIFORM_INTEGRATE (run-time compiled register-resident kernels, launch_custom.hip): the step loop of the fixed-step solvers runs
// until EVERY lane of the wavefront has reached its interval's end -- a lane that is there already takes steps of length zero (y + 0 k = y;
// tnext = tprev = t1 by then) -- so that the loop, whose body is where these kernels spill, runs under a full execution mask.  The wavefront
// waits for its slowest lane either way.  NOTES.md R5.1.
#ifndef CDKF_UNIFORM_INTEGRATE
#define CDKF_UNIFORM_INTEGRATE 0
#endif
CDKF_DEV bool wave_any(bool c) {
#if defined(CDKF_HOST_SIM)
  return c;  // (the host run takes the lanes of these kernels one at a time)
#else
  return __builtin_amdgcn_ballot_w64(c) != 0;
#endif
}

template <typename R, int NS, int MEAN_ONLY = 0, int NERR = NS, typename Rhs>
CDKF_DEV bool integrate(R (&y)[NS], R t0, R t1, R dt0, long max_steps, const Rhs& rhs, const RkTab<R>& tb) {
  if (tb.adaptive) return integrate_adaptive<R, NS, MEAN_ONLY, NERR>(y, t0, t1, dt0, max_steps, rhs, tb);
  R tprev = t0;
  R tnext = rmin(t0 + dt0, t1);
  long steps = 0;
  bool capped = false;
#if CDKF_UNIFORM_INTEGRATE
  while (wave_any(tprev < t1 && !capped)) {
    const bool act = tprev < t1 && !capped;
    if (act && steps >= max_steps) capped = true;
    const bool go = act && !capped;
    rk_step<R, NS>(y, go ? tnext - tprev : R(0), rhs, tb, tprev);
    if (go) {
      tprev = rmin(tnext, t1);
      R tn = tnext + dt0;
      tnext = (tn > t1 - Tol<R>::v) ? t1 : tn;
      ++steps;
    }
  }
#else
  while (tprev < t1) {
    if (steps >= max_steps) {
      capped = true;
      break;
    }
    rk_step<R, NS>(y, tnext - tprev, rhs, tb, tprev);
    tprev = rmin(tnext, t1);
    R tn = tnext + dt0;
    tnext = (tn > t1 - Tol<R>::v) ? t1 : tn;
    ++steps;
  }
#endif
  return capped;
}

template <typename R, int NS, int MEAN_ONLY = 0, int NERR = NS, typename Rhs>
CDKF_DEV bool integrate(R (&y)[NS], R t0, R t1, R dt0, long max_steps, const Rhs& rhs, const Dp5V<R>& C) {
  R tprev = t0;
  R tnext = rmin(t0 + dt0, t1);
  long steps = 0;
  bool capped = false;
#if CDKF_UNIFORM_INTEGRATE
  while (wave_any(tprev < t1 && !capped)) {
    const bool act = tprev < t1 && !capped;
    if (act && steps >= max_steps) capped = true;
    const bool go = act && !capped;
    dopri5_step<R, NS>(y, go ? tnext - tprev : R(0), rhs, C, tprev);
    if (go) {
      tprev = rmin(tnext, t1);
      R tn = tnext + dt0;
      tnext = (tn > t1 - Tol<R>::v) ? t1 : tn;
      ++steps;
    }
  }
#else
  while (tprev < t1) {
    if (steps >= max_steps) {
      capped = true;
      break;
    }
    dopri5_step<R, NS>(y, tnext - tprev, rhs, C, tprev);
    tprev = rmin(tnext, t1);
    R tn = tnext + dt0;
    tnext = (tn > t1 - Tol<R>::v) ? t1 : tn;
    ++steps;
  }
#endif
  return capped;
}

}  // namespace cdkf
