/*
 * cdkf.h -- C ABI of the MI355X-native continuous-discrete Gaussian filtering engine.
 *
 * This is the drop-in boundary for the batched-trajectory hot path of hd-UQ/cd_dynamax.  The
 * reference is pure Python/JAX and has no FFI of its own, so each entry
 * point below names the reference *function* it replaces (paths relative to /root/reference).
 * Only plain pointers, sizes and PODs cross this boundary: no torch / numpy / HIP types.
 *
 * Conventions
 *  - All matrices are dense row-major.  N = trajectories, T = observations per trajectory,
 *    d = state_dim, m = emission_dim.
 *  - Reference layout, opts.layout = CDKF_LAYOUT_NT (what `jax.vmap` over the trajectory axis produces):
 *      t  [N,T]   (or [T] when opts.t_shared != 0; the reference's t_emissions[:,0])
 *      y  [N,T,m]
 *      means [N,T,d], covariances [N,T,d,d], ll [N], status [N]
 *    Time-major layouts, opts.layout = CDKF_LAYOUT_TN / CDKF_LAYOUT_TCN: see the CDKF_LAYOUT_* defines.
 *  - Output pointers may be NULL: that field is then not produced (the reference's
 *    `output_fields` filter, inference_ekf.py:209,315).
 *  - Functions return 0 on success or a negative CDKF_E* code; cdkf_last_error() returns a
 *    thread-local message.  Numerical failures are NOT errors: they are flagged in status[n]
 *    and the outputs go NaN from that step on, like the reference's silent NaN propagation.
 *  - `_dev` variants take DEVICE pointers (hipMalloc'ed / torch tensors' data_ptr) and a
 *    hipStream_t passed as void* (NULL = the default stream); they enqueue work and return without
 *    synchronising.  The host variants allocate, copy, run, copy back and free.
 *  - The model parameter block is always given in double on the host and converted to the
 *    compute type inside the library.
 */
#ifndef CDKF_H
#define CDKF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CDKF_VERSION 110 /* 0.5.0 */

/* error codes */
#define CDKF_OK 0
#define CDKF_EINVAL (-1)      /* bad argument (NULL pointer, negative size, ...) */
#define CDKF_EUNSUPPORTED (-2) /* no kernel for this (drift, d, m, option) combination */
#define CDKF_EHIP (-3)        /* a HIP runtime call failed (no device, out of memory, ...) */

/* status[n] bit flags */
#define CDKF_STATUS_NOT_PD 1    /* a Cholesky pivot was <= 0 (S, or P in the UKF / smoother) */
#define CDKF_STATUS_NAN 2       /* a NaN reached the filtered mean */
#define CDKF_STATUS_MAX_STEPS 4 /* an observation interval needed more than max_steps RK steps */

/* drift registry: the reference's `params.dynamics.drift` Learnable* classes
 * (src/continuous_discrete_nonlinear_gaussian_ssm/cdnlgssm_utils.py:50-83) plus two build-defined
 * families for BASELINE.json configs 4 and 5.  theta layouts: */
#define CDKF_DRIFT_LINEAR 0   /* LearnableLinear: theta = [W (d*d), b (d)]; f = W x + b */
#define CDKF_DRIFT_LORENZ63 1 /* LearnableLorenz63: theta = [sigma, rho, beta]; d = 3 */
#define CDKF_DRIFT_LORENZ96 2 /* theta = [F]; f_i = (x_{i+1} - x_{i-2}) x_{i-1} - x_i + F; d >= 4 */
#define CDKF_SOLVER_DOPRI5 0   /* Dormand-Prince 5(4): diffrax.Dopri5, the reference's default */
#define CDKF_SOLVER_TSIT5 1    /* Tsitouras 5(4): diffrax.Tsit5 */
#define CDKF_SOLVER_BOSH3 2    /* Bogacki-Shampine 3(2): diffrax.Bosh3 */
#define CDKF_SOLVER_HEUN 3     /* explicit trapezoid: diffrax.Heun */
#define CDKF_SOLVER_MIDPOINT 4 /* explicit midpoint: diffrax.Midpoint */
#define CDKF_SOLVER_RALSTON 5  /* Ralston's 2nd-order method: diffrax.Ralston */
#define CDKF_SOLVER_EULER 6    /* diffrax.Euler */

#define CDKF_DRIFT_CUSTOM_BASE 1000 /* kinds >= this come from cdkf_custom_drift_register() */
#define CDKF_EMISSION_CUSTOM_BASE 1000 /* emission kinds >= this come from cdkf_custom_emission_register() */
#define CDKF_DRIFT_MLP_TANH 3 /* theta = [W1 (h1*d), b1 (h1), W2 (h2*h1), b2 (h2), W3 (d*h2), b3 (d)];
                                 f = W3 tanh(W2 tanh(W1 x + b1) + b2) + b3 */

/* array layouts (opts.layout).  NT is what jax.vmap over trajectories produces in the reference;
 * TN / TCN are time-major: at step k the 64 trajectories of a wavefront touch one contiguous run of
 * memory. */
#define CDKF_LAYOUT_NT 0 /* t [N,T], y [N,T,m], means [N,T,d], covariances [N,T,d,d] */
#define CDKF_LAYOUT_TN 1 /* t [T,N], y [T,N,m], means [T,N,d], covariances [T,N,d,d] */
#define CDKF_LAYOUT_TCN 2 /* t [T,N], y [T,m,N], means [T,d,N], covariances [T,d,d,N]: trajectory index
                             fastest, so that lane n of a wavefront touches element n of a contiguous run in
                             EVERY load and store instruction (fully coalesced; the engine's native layout) */
#define CDKF_LAYOUT_SAME (-1) /* opts.layout_in: the inputs use opts.layout */

/* state_order of EKFHyperParams (inference_ekf.py:40) */
#define CDKF_ORDER_ZEROTH 0
#define CDKF_ORDER_FIRST 1
#define CDKF_ORDER_SECOND 2

/* ParamsCDNLGSSM (cdnlgssm_utils.py:191-209) restricted to what the hot path reads.  The
 * emission function is the reference's LearnableLinear: h(x) = H x + h_bias. */
typedef struct cdkf_model {
  int32_t drift_kind;   /* CDKF_DRIFT_* */
  int32_t state_dim;    /* d */
  int32_t emission_dim; /* m */
  int32_t hidden1;      /* MLP only */
  int32_t hidden2;      /* MLP only */
  int32_t emission_kind; /* 0: linear emission h(x) = H x + h_bias; >= CDKF_EMISSION_CUSTOM_BASE: a registered custom emission,
                            whose parameter vector eta is [H (m*d, row-major) | h_bias (m)] (custom drift kinds only) */
  int64_t n_theta;
  const double* theta; /* [n_theta] drift parameters */
  const double* L;     /* [d,d] diffusion coefficient  (params.dynamics.diffusion_coefficient) */
  const double* Qc;    /* [d,d] diffusion covariance   (params.dynamics.diffusion_cov) */
  const double* H;     /* [m,d] emission weights */
  const double* h_bias;/* [m]   emission bias */
  const double* R;     /* [m,m] emission covariance */
  const double* m0;    /* [d]   initial mean */
  const double* P0;    /* [d,d] initial covariance */
  int32_t input_dim;   /* (version 109) d_u: length of the inputs row u_k = inputs[t0_idx] the reference hands to the drift and the
                          emission function with every call, f(x, u, t), h(x, u, t) (inference_ekf.py:95, 101-114, 277-286;
                          inference_ukf.py:142, 189).  0: none.  Read by drifts / emissions given as source (their snippets see
                          `u[0 .. d_u-1]` and the time `t`); the registry drifts ignore inputs and time, as the reference's own
                          LearnableLinear / LearnableLorenz63 do (cdnlgssm_utils.py:50-83).  The array itself: cdkf_opts.inputs. */
  int32_t reserved0;
} cdkf_model;

/* EKFHyperParams / UKFHyperParams (inference_ekf.py:34-44, inference_ukf.py:25-33) and the
 * defaults of src/utils/diffrax_utils.py:40-52 (Dopri5, ConstantStepSize, dt0 = 0.01). */
/* cdkf_opts.flags */
#define CDKF_FLAG_UKF_SIGMA_POINTS 1 /* unscented filter: form the 2 d + 1 sigma points and factorise the covariance in EVERY Runge-Kutta
                                        stage, literally as inference_ukf.py:45-60, 93-159 do -- also for the drifts whose weighted sums
                                        this library otherwise evaluates in closed form (Lorenz-63 / linear on the sixteen-lane grid,
                                        DESIGN.md 3.2b: same numbers to rounding, 3.5x faster).  With the flag a trajectory turns NaN
                                        exactly where the reference's does: when a STAGE covariance inside an interval loses positive
                                        definiteness (jnp.linalg.cholesky, inference_ukf.py:57 called from :138), not only when the
                                        covariance at an observation does. */

typedef struct cdkf_opts {
  int32_t state_order;  /* CDKF_ORDER_*; default SECOND */
  int32_t num_iter;     /* EKF update re-linearisations; default 1 */
  int32_t t_shared;     /* 0: t is [N,T]; 1: t is [T], shared by all trajectories */
  int32_t device;       /* HIP device ordinal; -1 = current device */
  int32_t layout;       /* CDKF_LAYOUT_NT (default), _TN or _TCN; applies to t (unless shared), y and every
                           mean / covariance array.  ll and status are always [N]. */
  int32_t forecast;     /* 0 (default): filter.  1: forecast mode -- the measurement update and the log-likelihood are
                           skipped, so predicted_means[k] / predicted_covs[k] are the moments pushed from (m0, P0) at
                           t[0] to t[k+1]; y is ignored (may be NULL) and ll is set to 0.  Replaces
                           forecast_extended_kalman_filter / forecast_unscented_kalman_filter
                           (inference_ekf.py:679-766, inference_ukf.py:409-505). */
  int32_t solver;       /* CDKF_SOLVER_*: the explicit Runge-Kutta method of the predict step, fixed steps of dt0 (the
                           reference forwards diffeqsolve_settings['solver'] to diffrax, src/utils/diffrax_utils.py:40-57).
                           Default DOPRI5 (diffrax_utils.py:120-123).  The other methods run on the register-resident and the
                           workgroup kernels (incl. custom drifts), through both gradient paths and in the type-1 smoother. */
  int32_t adaptive;     /* 0 (default): fixed steps of dt0 (diffrax.ConstantStepSize).  1: diffrax.PIDController(rtol, atol, pcoeff,
                           icoeff, dcoeff) around the method's embedded error estimate (DOPRI5, TSIT5, BOSH3, HEUN), first step
                           dt0, every trajectory adapting on its own; max_steps then counts accepted and rejected steps.
                           Filters and smoothers of every shape (larger models on the workgroup kernels) and custom drifts;
                           the type-1 smoother's pushed-forward (A, Q) adapt as well; so does the reverse-sweep gradient: its forward
                           sweep logs the accepted step sizes (up to CDKF_ADJ_DT_CAP = 64 per interval, MAX_STEPS status beyond) and
                           the reverse sweep replays them as constants (the controller's factor carries no derivative). */
  int64_t max_steps;    /* RK steps per observation interval; default 100000 */
  double dt0;           /* default 0.01 */
  double dt_final;      /* default 1e-10 (inference_ekf.py:39) */
  double cov_rescaling; /* zeroth-order only; default 1.0 */
  double ukf_alpha;     /* default sqrt(3) */
  double ukf_beta;      /* default 2 */
  double ukf_kappa;     /* default 1 */
  double rtol;          /* adaptive only; defaults 1e-3 / 1e-6 (the values of the reference's tutorial notebook) */
  double atol;
  double pid_p;         /* PIDController pcoeff / icoeff / dcoeff; defaults 0 / 1 / 0 (an I-controller, diffrax's default) */
  double pid_i;
  double pid_d;
  int32_t layout_in;    /* layout of the INPUT arrays t and y: CDKF_LAYOUT_SAME (default: like `layout`) or a CDKF_LAYOUT_* value.
                           E.g. layout = TCN with layout_in = NT lets a host caller hand over the reference's [N,T,m] arrays
                           untransposed and still get the coalesced native output layout. */
  int32_t flags;        /* bit mask of CDKF_FLAG_*; default 0; unknown bits are refused (CDKF_EINVAL) */
  double dtmin;         /* adaptive only: PIDController(dtmin=, dtmax=) with force_dtmin=True (diffrax's default) -- every proposed step size,
                           the first included, is clipped to [dtmin, dtmax], and a step taken at dtmin is kept whatever its error estimate.
                           Defaults 0 and +infinity (= no bounds; src/utils/diffrax_utils.py:40-57 forwards the controller object) */
  double dtmax;          /* (0 is read as "no bound", so that a zero-initialised struct keeps working) */
  const void* inputs;    /* (version 109) inputs [N,T,d_u] (d_u = cdkf_model.input_dim), laid out like y under opts.layout_in, in the
                            real type of the entry point called and with the residence of its y (host memory for the host entry
                            points, device memory for the _dev ones).  NULL with input_dim > 0: zeros (the reference's
                            _process_input, inference_ekf.py:32, 260).  Clients built against version 108 must be rebuilt: the two
                            structs grew. */
  double pid_safety;     /* (version 110) adaptive only: PIDController(safety=, factormin=, factormax=) -- the next step size is the attempted */
  double pid_factormin;  /* one times clip(safety * e^-c1 e1^-c2 e2^-c3, [1 if the step was kept else factormin, factormax]).  Defaults 0.9, */
  double pid_factormax;  /* 0.2, 10 (diffrax's); 0 is read as the default, so that a zero-initialised struct keeps working.  cdkf_struct_sizes() */
                         /* tells a binding whether its mirror of this struct is the library's. */
} cdkf_opts;

/* Fill *opts with the reference defaults listed above. */
void cdkf_default_opts(cdkf_opts* opts);

int cdkf_version(void);
/* sizeof(cdkf_model) and sizeof(cdkf_opts) as THIS library was built.  Both structs have grown (versions 107, 109): a binding that mirrors
 * them (ctypes, cgo, JNI) compares these with its own sizes before the first call instead of letting the library read past a shorter
 * struct (cd_dynamax_amd/_ffi.py does, and raises). */
void cdkf_struct_sizes(int64_t* model_bytes, int64_t* opts_bytes);
const char* cdkf_last_error(void);
/* Number of HIP devices visible, or a negative error code. */
int cdkf_device_count(void);
/* 1 if a kernel exists for this model/algorithm/precision, else 0.  algo: 0 EKF filter,
 * 1 UKF filter, 2 EKF smoother.  bytes_per_real: 4 or 8. */
int cdkf_supported(const cdkf_model* mdl, const cdkf_opts* opts, int algo, int bytes_per_real);

/* The layout (CDKF_LAYOUT_*) in which this model's kernels move data fastest: _TCN for the lane-per-trajectory
 * kernels (small state_dim), _TN for the workgroup-per-trajectory kernels (a trajectory's d x d block contiguous). */
int cdkf_preferred_layout(const cdkf_model* mdl);

/* Informational: how many distinct trajectories the lane-per-trajectory sweeps (state_dim <= 4) put on one 64-lane
 * wavefront for a batch of N on the current device -- 64 once N fills the chip, fewer (a power of two; the remaining
 * lanes repeat them) for small batches, so that a long interval of one trajectory delays fewer others.  Results do not
 * depend on it.  The environment variable CDKF_LANES_PER_WAVE (1 ... 64, power of two) overrides the choice.
 * (Lorenz-63 EKF batches of at most 4096 trajectories (16 per CU) with H = I run on a third mapping, sixteen lanes per trajectory,
 * which agrees with the others to rounding; CDKF_NO_LPE=1 disables it.) */
int cdkf_trajectories_per_wavefront(int64_t N);

/* ---- device memory helpers (so that host code needs no other GPU runtime) ------------------- */
int cdkf_malloc(void** dev_ptr, int64_t bytes);
int cdkf_free(void* dev_ptr);
int cdkf_memcpy_h2d(void* dev_dst, const void* host_src, int64_t bytes);
int cdkf_memcpy_d2h(void* host_dst, const void* dev_src, int64_t bytes);
int cdkf_memset(void* dev_ptr, int value, int64_t bytes);
int cdkf_synchronize(void* stream);

/* ---- EKF filter: replaces extended_kalman_filter / iterated_extended_kalman_filter
 *      (src/continuous_discrete_nonlinear_gaussian_ssm/inference_ekf.py:202-361) under
 *      jax.vmap over trajectories (src/ssm_temissions.py:555-566). ------------------------------ */
int cdkf_ekf_filter_f64(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T,
                        const double* t, const double* y, double* ll, double* filtered_means,
                        double* filtered_covs, double* predicted_means, double* predicted_covs,
                        int32_t* status);
int cdkf_ekf_filter_f32(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T,
                        const float* t, const float* y, float* ll, float* filtered_means,
                        float* filtered_covs, float* predicted_means, float* predicted_covs,
                        int32_t* status);
int cdkf_ekf_filter_f64_dev(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T,
                            const double* t, const double* y, double* ll, double* filtered_means,
                            double* filtered_covs, double* predicted_means, double* predicted_covs,
                            int32_t* status, void* stream);
int cdkf_ekf_filter_f32_dev(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T,
                            const float* t, const float* y, float* ll, float* filtered_means,
                            float* filtered_covs, float* predicted_means, float* predicted_covs,
                            int32_t* status, void* stream);

/* ---- UKF filter: replaces unscented_kalman_filter (inference_ukf.py:206-308). ---------------- */
int cdkf_ukf_filter_f64(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T,
                        const double* t, const double* y, double* ll, double* filtered_means,
                        double* filtered_covs, double* predicted_means, double* predicted_covs,
                        int32_t* status);
int cdkf_ukf_filter_f32(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T,
                        const float* t, const float* y, float* ll, float* filtered_means,
                        float* filtered_covs, float* predicted_means, float* predicted_covs,
                        int32_t* status);
int cdkf_ukf_filter_f64_dev(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T,
                            const double* t, const double* y, double* ll, double* filtered_means,
                            double* filtered_covs, double* predicted_means, double* predicted_covs,
                            int32_t* status, void* stream);
int cdkf_ukf_filter_f32_dev(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T,
                            const float* t, const float* y, float* ll, float* filtered_means,
                            float* filtered_covs, float* predicted_means, float* predicted_covs,
                            int32_t* status, void* stream);

/* ---- EKF (RTS) smoother: replaces extended_kalman_smoother / iterated_extended_kalman_smoother
 *      (inference_ekf.py:450-593): runs the filter (num_iter = 1) and then the backward sweep
 *      of _smooth (inference_ekf.py:363-448).  filtered_* and smoothed_* are all required. ----- */
int cdkf_ekf_smoother_f64(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T,
                          const double* t, const double* y, double* ll, double* filtered_means,
                          double* filtered_covs, double* smoothed_means, double* smoothed_covs,
                          int32_t* status);
int cdkf_ekf_smoother_f32(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T,
                          const float* t, const float* y, float* ll, float* filtered_means,
                          float* filtered_covs, float* smoothed_means, float* smoothed_covs,
                          int32_t* status);
int cdkf_ekf_smoother_f64_dev(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T,
                              const double* t, const double* y, double* ll, double* filtered_means,
                              double* filtered_covs, double* smoothed_means, double* smoothed_covs,
                              int32_t* status, void* stream);
int cdkf_ekf_smoother_f32_dev(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T,
                              const float* t, const float* y, float* ll, float* filtered_means,
                              float* filtered_covs, float* smoothed_means, float* smoothed_covs,
                              int32_t* status, void* stream);

/* ---- emission moments of Gaussian state marginals: replaces emissions_extended_kalman_filter /
 *      emissions_unscented_kalman_filter (inference_ekf.py:768-855, inference_ukf.py:507-612; identical for the
 *      linear emission of the registry): out_mean[r] = H mean[r] + h_bias, out_cov[r] = H cov[r] H^T + R.
 *      `rows` state marginals stored contiguously: means [rows,d], covs [rows,d,d] (NULL: point estimates, out_cov is
 *      then not written), out_mean [rows,m], out_cov [rows,m,m]. ---------------------------------------------- */
int cdkf_emission_moments_f64(const cdkf_model* mdl, int64_t rows, const double* means, const double* covs,
                              double* out_mean, double* out_cov);
int cdkf_emission_moments_f32(const cdkf_model* mdl, int64_t rows, const float* means, const float* covs,
                              float* out_mean, float* out_cov);
int cdkf_emission_moments_f64_dev(const cdkf_model* mdl, int64_t rows, const double* means, const double* covs,
                                  double* out_mean, double* out_cov, void* stream);
int cdkf_emission_moments_f32_dev(const cdkf_model* mdl, int64_t rows, const float* means, const float* covs,
                                  float* out_mean, float* out_cov, void* stream);
/* The same for an emission given as SOURCE (emission_kind >= CDKF_EMISSION_CUSTOM_BASE; state_dim, emission_dim <= 16), where the two
 * reference functions differ: ukf = 0 -- emissions_extended_kalman_filter (inference_ekf.py:768-855): h(m, u, t) and
 * jacfwd(h) P jacfwd(h)^T + R, the Jacobian from h_src by dual numbers; ukf = 1 -- emissions_unscented_kalman_filter
 * (inference_ukf.py:507-612): the 2 d + 1 sigma points of (m, P) (opts->ukf_alpha / _beta / _kappa) through h, weighted mean and
 * covariance + R.  Host buffers: t [rows] (NULL: 0), inputs [rows, input_dim] (NULL when input_dim is 0), means, covs (NULL: point
 * estimates, out_cov not written) and outputs as above.  A kernel compiled at run time around the model's statements
 * (cdkf_custom_emission_moments_compile: the build check, no GPU needed). */
int cdkf_custom_emission_moments_f64(const cdkf_model* mdl, const cdkf_opts* opts, int ukf, int64_t rows, const double* t,
                                     const double* inputs, const double* means, const double* covs, double* out_mean, double* out_cov);
int cdkf_custom_emission_moments_f32(const cdkf_model* mdl, const cdkf_opts* opts, int ukf, int64_t rows, const float* t,
                                     const float* inputs, const float* means, const float* covs, float* out_mean, float* out_cov);
/* device pointers (same shapes), asynchronous on `stream` (NULL: the default stream) */
int cdkf_custom_emission_moments_f64_dev(const cdkf_model* mdl, const cdkf_opts* opts, int ukf, int64_t rows, const double* t,
                                         const double* inputs, const double* means, const double* covs, double* out_mean, double* out_cov,
                                         void* stream);
int cdkf_custom_emission_moments_f32_dev(const cdkf_model* mdl, const cdkf_opts* opts, int ukf, int64_t rows, const float* t,
                                         const float* inputs, const float* means, const float* covs, float* out_mean, float* out_cov,
                                         void* stream);
int cdkf_custom_emission_moments_compile(const cdkf_model* mdl, const cdkf_opts* opts, int bytes_per_real);

/* ---- user-supplied drifts.  The reference accepts any Python callable as ParamsCDNLGSSMDynamics.drift
 *      (src/continuous_discrete_nonlinear_gaussian_ssm/cdnlgssm_utils.py:38-61); across a C ABI the drift is C source,
 *      compiled at run time (hipRTC) into the same register-resident sweep kernels the built-in drifts use.
 *        f_src       body computing  fx[i] = f_i(x, theta)         (x: const R*, theta: const R*, R = float or double)
 *        jac_src     body assigning the NON-ZERO entries F[i][j] = d f_i / d x_j   (F is zeroed first)
 *        divgrad_src body assigning g[i] = d/dx_i sum_j d f_j/d x_j, or NULL: then EKF state_order 'second' is refused
 *                    (the reference's second-order mean term 0.5 P grad(div f), inference_ekf.py:108-116)
 *      e.g. a pendulum:  f_src "fx[0] = x[1]; fx[1] = -theta[0] * sin(x[0]);"
 *                        jac_src "F[0][1] = R(1); F[1][0] = -theta[0] * cos(x[0]);"
 *      Derivatives by dual numbers (round 3; the reference calls jacfwd / value_and_grad on the callable it is given,
 *      inference_ekf.py:95, 108-116, ssm_temissions.py:550-568): jac_src NULL or empty -- the Jacobian is derived from f_src;
 *      divgrad_src "auto" -- so is grad(div f); and cdkf_ekf_loglik_grad_* differentiates the log-likelihood w.r.t. theta
 *      (forward sensitivities, a lane per (trajectory, parameter); linear emission, num_iter 1, state_order 'first', or 'second'
 *      with an empty divgrad_src = identically zero).  f_src is then compiled a second time with the scalar type T = a dual
 *      number in place of R: x, theta and fx are arrays of T there -- declare temporaries `auto` or `T`, not `R`
 *      (R(...) constants, +, -, *, /, sin, cos, tan, tanh, sinh, cosh, exp, log, sqrt, pow, fabs, atan and comparisons are provided).
 *      state_dim, emission_dim <= 6: the register-resident kernels.  Beyond that (state_dim <= 64, as far as the workgroup kernels'
 *      LDS plan holds the shape: e.g. d = m = 40 in fp64, 60 in fp32) the same f_src is compiled into the workgroup-per-trajectory
 *      sweeps (round 3): jac_src must then be NULL and divgrad_src NULL, "" or "auto" -- a thread of the workgroup evaluates ONE
 *      direction of the Jacobian / one (i, k) pair of the second derivatives / one pair of sigma points, all by dual numbers; linear
 *      emission only; 10 - 20 s of compilation per variant on first use.  cdkf_ekf_loglik_grad[_all]_* take these drifts on the
 *      shape-generic reverse sweep compiled with the source (state_order 'first', or 'second' with divgrad_src "" or "auto" -- third
 *      derivatives by triply nested dual numbers; state_dim + n_theta <= 256; max(state_dim, emission_dim) <= 43 in fp64, 62 in
 *      fp32) -- the _all variant at any state_dim.
 *      Returns the drift_kind to put in cdkf_model (>= CDKF_DRIFT_CUSTOM_BASE; the
 *      same sources give the same kind) or a negative CDKF_E* code.  Filters (EKF all orders, UKF), EKF smoother,
 *      forecast mode and the gradients are available for custom kinds; kernels compile
 *      on first use (seconds) and are cached.  Compile errors in the snippets surface through cdkf_last_error(). */
int cdkf_custom_drift_register(int state_dim, int n_theta, const char* f_src, const char* jac_src,
                               const char* divgrad_src);
/* A user-supplied EMISSION function (the reference accepts any callable h and linearises it with jacfwd,
 * inference_ekf.py:258-259, or pushes the sigma points through it, inference_ukf.py:162-203):
 *   h_src     body computing  hx[r] = h_r(x, eta)                 (x: const R*, eta: const R*, R = float or double)
 *   hjac_src  body assigning the NON-ZERO entries H[r][k] = d h_r / d x_k   (H is zeroed first); NULL or empty: derived from h_src
 *             by dual numbers (h_src is then also compiled with T = a dual number in place of R: temporaries `auto` / `T`)
 * eta is the model's emission block read as a flat vector: eta[r*d + k] = cdkf_model.H[r][k], eta[m*d + r] = h_bias[r].
 * e.g. observing the sine of a pendulum angle: h_src "hx[0] = eta[0] * sin(x[0]);"  hjac_src "H[0][0] = eta[0] * cos(x[0]);"
 * Returns the value for cdkf_model.emission_kind (>= CDKF_EMISSION_CUSTOM_BASE) or a negative CDKF_E* code.  Runs on the
 * run-time compiled kernels: the drift must be a custom kind too.  state_dim, emission_dim <= 6: EKF (re-linearised in every update
 * iteration) and UKF filters, EKF smoother, the log-likelihood gradients of both filters (cdkf_*_loglik_grad[_all]_*: every leaf,
 * eta included, forward mode through the literal recursions).  Up to 16: the same entry points on other kernels (the literal recursions
 * of csrc/cdkf_ukf_tangent_kernels.h; the smoother's backward sweep on the workgroup kernels) -- h_src is then always compiled over dual numbers.  Emission moments: cdkf_custom_emission_moments_*. */
int cdkf_custom_emission_register(int state_dim, int emission_dim, const char* h_src, const char* hjac_src);
/* compile (without loading: no GPU needed) the kernel for one variant -- algo 0 EKF filter, 1 UKF filter, 2 EKF
 * smoother, 3 the log-likelihood gradient (the forward-sensitivity sweep up to six dimensions, the reverse sweep beyond);
 * emission_kind 0 or a registered custom emission -- to check the snippets early; 0 or a negative CDKF_E* code
 * with the compiler log in cdkf_last_error().  algo + 16 (state / emission dimension <= 6): the variant that takes its Runge-Kutta
 * tableau and step-size controller from the arguments (opts.solver other than Dormand-Prince, opts.adaptive); algo + 256 * d_u:
 * the variant for a model with cdkf_model.input_dim = d_u. */
int cdkf_custom_drift_compile(int drift_kind, int bytes_per_real, int emission_dim, int algo, int state_order,
                              int emission_kind);
/* directory holding the kernel headers (cdkf_reg_kernels.h ...) for run-time compilation; default: <dir of this
 * library>/../csrc */
void cdkf_set_kernel_source_dir(const char* dir);
/* how many run-time compiled variants THIS process loaded from the on-disk code-object cache (<library dir>/rtc_cache, else
 * $CDKF_RTC_CACHE_DIR / ~/.cache/cdkf_rtc; key = source + options + target + hipRTC and runtime versions + the kernel headers) and how
 * many it had to compile.  The cache directory's MANIFEST lists what each stored object is. */
void cdkf_rtc_cache_stats(int64_t* hits, int64_t* misses);

/* ---- diagnostics: the argument blocks a launch would hand its kernel, built WITHOUT touching the GPU -- for the CPU-sanitizer
 *      builds of the same device templates (cd_dynamax_amd/csrc/hostsim/, tests/test_hostsim.py: the kernels of this library compiled
 *      for the host under ASan / UBSan / MSan / TSan and run on exactly the arguments the launcher forms).  Replaces nothing in the
 *      reference; not part of the drop-in surface. --------------------------------------------------------------------------- */
/* run-time compiled register-resident kernel (custom drift, state / emission dimension <= 6): algo as cdkf_custom_drift_compile;
 * par_out receives the real-valued block (bytes_per_real each), ip_out[27] the integer block (26) followed by the grid size.
 * Returns the number of reals written or a negative CDKF_E* code. */
/* the tangent sweep above: args_out receives its argument struct (UtArgs<real>, pointers null), par_out its parameter block; returns the
 * number of reals in the block or a negative CDKF_E* code */
int cdkf_debug_ukf_tangent_args(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, int bytes_per_real, int all,
                                void* args_out, int64_t args_cap_bytes, void* par_out, int64_t par_cap_bytes);
/* the check the run-time compiled kernels get before they are trusted at -O3 (launch_custom.hip: rtc_exec_prologue_defect): 1 when the
 * code object (an ELF for `arch`, e.g. "gfx950") shows vector spill code in front of an execution-mask restore -- the ROCm 7.2
 * register-allocation defect of NOTES.md R5.1 --, 0 clean, -1 could not look (no comgr disassembler in the process) */
int cdkf_debug_exec_prologue_check(const void* code, int64_t bytes, const char* arch, char* where, int64_t where_cap);
int cdkf_debug_custom_reg_blob(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, int algo, int bytes_per_real,
                               void* par_out, int64_t par_cap_bytes, int64_t* ip_out);
/* workgroup-per-trajectory kernels (cdkf_wg2_kernels.h: any drift, state / emission dimension <= 64): args_out receives the kernel's
 * argument struct (WgArgs<real>, every pointer null), blob_out its parameter block, geom_out[6] = {covariance entries per thread,
 * threads per workgroup, LDS bytes, sizeof(WgArgs<real>), 0, 0}; ukf / smoother = 1 select the unscented filter / the backward sweep;
 * smoother = 2: the reverse sweep of the gradient (ekf_adjoint_wg_kernel), geom_out[4], [5] = its scratch reals per trajectory and the
 * number of step starts it keeps per replay chunk.
 * Returns the number of reals in the block or a negative CDKF_E* code. */
int cdkf_debug_wg_args(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, int bytes_per_real, int ukf, int smoother,
                       void* args_out, int64_t args_cap_bytes, void* blob_out, int64_t blob_cap_bytes, int64_t* geom_out);

/* ---- linear model, smoother type 1: replaces cdlgssm_smoother(..., smoother_type='cd_smoother_1') -- the reference's default --
 *      src/continuous_discrete_linear_gaussian_ssm/inference.py:694-823 (_step_1 :746-773, compute_pushforward :105-143):
 *      filter sweep, the pushed-forward (A, Q) of every interval (Dopri5, dt0), discrete RTS (Sarkka Alg. 3.17).
 *      drift_kind LINEAR with zero bias, state_dim <= 8: cdkf_kf_smoother1_supported().  Arrays follow opts.layout;
 *      smoothed_cross (optional, NULL to skip) has the shape and strides of smoothed_covs, entries k = 0 .. T-2 written
 *      (cross[k] = C_k P_s[k+1] + m_s[k] m_s[k+1]^T, inference.py:769). --------------------------------------------------- */
int cdkf_kf_smoother1_f64(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, const double* t, const double* y,
                          double* ll, double* filtered_means, double* filtered_covs, double* smoothed_means,
                          double* smoothed_covs, double* smoothed_cross, int32_t* status);
int cdkf_kf_smoother1_f32(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, const float* t, const float* y,
                          float* ll, float* filtered_means, float* filtered_covs, float* smoothed_means, float* smoothed_covs,
                          float* smoothed_cross, int32_t* status);
int cdkf_kf_smoother1_f64_dev(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, const double* t,
                              const double* y, double* ll, double* filtered_means, double* filtered_covs,
                              double* smoothed_means, double* smoothed_covs, double* smoothed_cross, int32_t* status,
                              void* stream);
int cdkf_kf_smoother1_f32_dev(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, const float* t,
                              const float* y, float* ll, float* filtered_means, float* filtered_covs, float* smoothed_means,
                              float* smoothed_covs, float* smoothed_cross, int32_t* status, void* stream);
int cdkf_kf_smoother1_supported(const cdkf_model* mdl);
/* The pushed-forward (A, Q) of every observation interval by themselves -- compute_pushforward, inference.py:105-143 -- for the
 * same shapes (cdkf_kf_smoother1_supported): AQ [N, T-1, 2, d, d] row-major (A then Q per interval), host arrays; t follows
 * opts.layout_in / opts.t_shared.  The host side uses it for what the reference adds to the pushed-forward mean WITHOUT
 * integrating it (dynamics bias and inputs, inference.py:185-205 _predict: mu = F m + B u + b): the offsets s_k+1 = A_k s_k + B u_k + b
 * are formed from these A_k, the filter runs on y - H s - D u - d, and s is added back to the means. */
int cdkf_kf_pushforward_f64(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, const double* t, double* AQ);
int cdkf_kf_pushforward_f32(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, const float* t, float* AQ);

/* ---- marginal log-likelihood AND its gradient w.r.t. the drift parameters theta (ordering of cdkf_model.theta):
 *      replaces jax.value_and_grad of the fit_sgd loss, src/ssm_temissions.py:550-568, for the drift block of
 *      the parameters (what the Lorenz-63 parameter-estimation tutorials learn).  EKF, num_iter 1; state_order
 *      first/second (MLP drift, 'second': the mean term 0.5 P grad(div f) is differentiated too); shapes:
 *      cdkf_grad_supported().  ll [N], grad [N, n_theta] row-major whatever opts.layout is
 *      (t and y follow opts.layout).  Exact derivative of the discretised recursion (forward sensitivities; small
 *      Lorenz-63 batches with H = I: a forward and a reverse sweep, moments in a grow-only device workspace). -- */
int cdkf_ekf_loglik_grad_f64(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, const double* t,
                             const double* y, double* ll, double* grad, int32_t* status);
int cdkf_ekf_loglik_grad_f32(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, const float* t,
                             const float* y, float* ll, float* grad, int32_t* status);
int cdkf_ekf_loglik_grad_f64_dev(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, const double* t,
                                 const double* y, double* ll, double* grad, int32_t* status, void* stream);
int cdkf_ekf_loglik_grad_f32_dev(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, const float* t,
                                 const float* y, float* ll, float* grad, int32_t* status, void* stream);
/* ---- the same for the UNSCENTED filter's marginal log-likelihood: value_and_grad of the fit_sgd / fit_mcmc objective with
 *      filter_hyperparams = UKFHyperParams() (src/ssm_temissions.py:500, 555-568 -> models.py:393-408 -> the UKF branch of
 *      cdnlgssm_filter, models.py:708 -> inference_ukf.py:206-308).  Forward sensitivities through the closed form of the
 *      sigma-point sums (exact for the Lorenz-63 and linear drifts: the derivative of a function does not depend on how it is
 *      written down); ll [N] equals cdkf_ukf_filter_*'s to rounding; a predicted covariance that is not positive definite gives
 *      NaN and the NOT_PD flag as the filter does.  cdkf_ukf_grad_supported(): register-resident Lorenz-63 / linear shapes. */
int cdkf_ukf_loglik_grad_f64(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, const double* t,
                             const double* y, double* ll, double* grad, int32_t* status);
int cdkf_ukf_loglik_grad_f32(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, const float* t,
                             const float* y, float* ll, float* grad, int32_t* status);
int cdkf_ukf_loglik_grad_f64_dev(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, const double* t,
                                 const double* y, double* ll, double* grad, int32_t* status, void* stream);
int cdkf_ukf_loglik_grad_f32_dev(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, const float* t,
                                 const float* y, float* ll, float* grad, int32_t* status, void* stream);
int cdkf_ukf_grad_supported(const cdkf_model* mdl, const cdkf_opts* opts);
/* ---- ... and w.r.t. EVERY leaf (grad_model as in cdkf_ekf_loglik_grad_all_*): the reverse sweeps over the unscented filter's moment
 *      equations in closed form -- the extended filter's plus the curvature term 0.5 sum_jk (d^2 f / dx_j dx_k) P_jk in the mean, exact
 *      for the quadratic Lorenz-63 / Lorenz-96 drifts and (no curvature) the linear one, any shape the reverse sweeps take (state and
 *      emission dimension <= 8 on the wavefront sweep, Lorenz-96 / linear beyond on the workgroup sweep), linear emission, default
 *      solver.  What fit_sgd / fit_mcmc differentiate in the reference with filter_hyperparams = UKFHyperParams() when any leaf is
 *      trainable (src/ssm_temissions.py:500-568, 601-679 -> inference_ukf.py:93-203). */
int cdkf_ukf_loglik_grad_all_f64(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, const double* t,
                                 const double* y, double* ll, double* grad, double* grad_model, int32_t* status);
int cdkf_ukf_loglik_grad_all_f32(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, const float* t,
                                 const float* y, float* ll, float* grad, float* grad_model, int32_t* status);
int cdkf_ukf_loglik_grad_all_f64_dev(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, const double* t,
                                     const double* y, double* ll, double* grad, double* grad_model, int32_t* status,
                                     void* stream);
int cdkf_ukf_loglik_grad_all_f32_dev(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, const float* t,
                                     const float* y, float* ll, float* grad, float* grad_model, int32_t* status,
                                     void* stream);
int cdkf_ukf_grad_all_supported(const cdkf_model* mdl, const cdkf_opts* opts);
/* ... and for every model the closed forms do not cover -- an MLP drift, a drift or an emission given as source, the built-in drifts at
 *      other shapes -- behind the SAME entry points (cdkf_ukf_loglik_grad_*, cdkf_ukf_loglik_grad_all_*): forward mode through the literal
 *      sigma-point recursion, the Cholesky factor's derivative included, by dual numbers over the drift's / emission's statements
 *      (cdkf_ukf_tangent_kernels.h; state_dim, emission_dim <= 16, fixed-step Dormand-Prince, inputs and time-dependent f / h as the
 *      filters; a lane per (trajectory, leaf entry): a fallback, not a fast path).  With a custom emission the H / h_bias slots of
 *      grad_model hold the gradient w.r.t. its parameter vector eta = [H | h_bias].  Replaces jax.value_and_grad through
 *      unscented_kalman_filter, /root/reference/src/ssm_temissions.py:500, 555-568 -> inference_ukf.py:93-203.
 *      cdkf_ukf_tangent_compile: the kernel a model would get, compiled for gfx950 without a GPU (CDKF_OK or the compiler's message). */
int cdkf_ukf_tangent_compile(const cdkf_model* mdl, const cdkf_opts* opts, int bytes_per_real);
/* ... and the EXTENDED filter's gradient on the same plan, behind cdkf_ekf_loglik_grad[_all]_* where no forward-sensitivity kernel and no
 *      reverse sweep exists: num_iter > 1 above eight dimensions, emissions given as source (d ll / d eta in the H / h_bias slots), the
 *      MLP drift beyond its LDS plan -- state_dim, emission_dim <= 16, state_order first (second: state_dim <= 8), fixed-step
 *      Dormand-Prince.  jacfwd(f), jacfwd(h) by an outer dual level over the parameter tangent (inference_ekf.py:95, 258), the iterated
 *      update as inference_ekf.py:153-199 runs it. */
int cdkf_ekf_tangent_compile(const cdkf_model* mdl, const cdkf_opts* opts, int bytes_per_real);
/* ---- the same plus the gradient w.r.t. every other model parameter (the remaining leaves of the pytree jax.grad returns
 *      for ParamsCDNLGSSM): grad_model [N, d + 2 d^2 + m d + m + m^2] row-major, per trajectory
 *          m0 [d] | P0 [d,d] | LQL [d,d] | H [m,d] | h_bias [m] | R [m,m]
 *      LQL is the cotangent of L Qc L^T: dL = 2 LQL L Qc and dQc = L^T LQL L for symmetric Qc (the caller chains these;
 *      the Python host does).  Cotangents of the symmetric matrices (P0, LQL, R) are symmetric: they pair with
 *      symmetric perturbations, which is what a symmetric parametrisation (dynamax RealToPSDBijector) produces.
 *      Reverse sweep (discrete adjoint): cdkf_grad_all_supported() -- state_dim and emission_dim <= 8 for every registry drift
 *      (wavefront per trajectory); beyond that the Lorenz-96 and linear drifts on the workgroup-per-trajectory reverse sweep, as far as
 *      its LDS plan goes (max(state_dim, emission_dim) <= 43 in fp64, 62 in fp32; fixed steps).  Replaces jax.value_and_grad of
 *      marginal_log_prob, /root/reference/src/ssm_temissions.py:550-568. ---------------------------------------------------- */
int cdkf_ekf_loglik_grad_all_f64(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, const double* t,
                                 const double* y, double* ll, double* grad, double* grad_model, int32_t* status);
int cdkf_ekf_loglik_grad_all_f32(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, const float* t,
                                 const float* y, float* ll, float* grad, float* grad_model, int32_t* status);
int cdkf_ekf_loglik_grad_all_f64_dev(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, const double* t,
                                     const double* y, double* ll, double* grad, double* grad_model, int32_t* status,
                                     void* stream);
int cdkf_ekf_loglik_grad_all_f32_dev(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, const float* t,
                                     const float* y, float* ll, float* grad, float* grad_model, int32_t* status,
                                     void* stream);
/* ... the same for a model whose PREDICTED mean takes a jump behind every interval: jumps[n][k][:] ([N,T,d], contiguous, whatever
 * opts.layout says about t and y) is added to the mean predicted from t_k to t_{k+1}.  That is how the reference's linear model applies
 * its dynamics bias and inputs -- `mu_pred = A m + B u + b`, un-integrated
 * (/root/reference/src/continuous_discrete_linear_gaussian_ssm/inference.py:185-205, 596-620) -- so with jumps = u B^T + b this is
 * jax.value_and_grad of that model's marginal_log_prob (models.py:116-139, 167: bias and input weights are ordinary trainable leaves):
 * grad_jumps[n][k][:] = d ll_n / d jumps[n][k][:] (chain to b and B on the host), grad_y[n][k][:] = d ll_n / d y[n][k][:] (chain to the
 * emission input weights: y enters as y - D u).  state_dim, emission_dim <= 8, fixed-step Dopri5; host buffers. */
int cdkf_ekf_loglik_grad_jumps_f64(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, const double* t,
                                   const double* y, const double* jumps, double* ll, double* grad, double* grad_model,
                                   double* grad_jumps, double* grad_y, int32_t* status);
int cdkf_ekf_loglik_grad_jumps_f32(const cdkf_model* mdl, const cdkf_opts* opts, int64_t N, int64_t T, const float* t,
                                   const float* y, const float* jumps, float* ll, float* grad, float* grad_model,
                                   float* grad_jumps, float* grad_y, int32_t* status);
int cdkf_grad_all_supported(const cdkf_model* mdl, const cdkf_opts* opts);
/* The reverse sweeps behind cdkf_ekf_loglik_grad_* keep the forward sweep's moments -- and, where it pays, stage checkpoints: up to
 * CDKF_ADJ_CKPT_GB (environment, default 128) GB -- in ONE per-process device workspace that only grows and is reused by every later
 * call (no allocation inside an SGD / HMC loop).  This call waits for the last launch that uses it and returns the memory to the
 * device (e.g. before handing the GPU to another library); the next gradient call allocates again.  Counterpart in the reference:
 * none (XLA owns the buffers of jax.value_and_grad, src/ssm_temissions.py:550-568). */
int cdkf_release_workspace(void);
/* 1 if cdkf_ekf_loglik_grad_* has a kernel for this model/options, else 0 */
int cdkf_grad_supported(const cdkf_model* mdl, const cdkf_opts* opts);
/* out[p] = sum_n grad[n, p] on the device (the `.sum()` of ssm_temissions.py:567 applied to the gradient), so a
 * multi-GPU caller all-reduces n_theta scalars */
int cdkf_grad_sum_f64_dev(const double* grad, int64_t N, int64_t n_theta, double* out_sum, void* stream);
int cdkf_grad_sum_f32_dev(const float* grad, int64_t N, int64_t n_theta, double* out_sum, void* stream);

/* ---- sum_n ll[n]: the reduction of src/ssm_temissions.py:567 (`vmap(...)(...).sum()`), done on
 *      the device so that the multi-GPU caller can all-reduce ONE scalar over RCCL. ------------- */
int cdkf_ll_sum_f64_dev(const double* ll, int64_t N, double* out_sum, void* stream);
int cdkf_ll_sum_f32_dev(const float* ll, int64_t N, double* out_sum, void* stream);

/* ---- data-parallel reduction: the ONE collective of the path -------------------------------------------------------------------
 * Trajectories share the parameters and nothing else, so the reference's training losses are `vmap(...)(...).sum()` over the batch
 * (src/ssm_temissions.py:555-568 for fit_sgd, :665-679 for fit_mcmc).  Sharded over GPUs, each rank sweeps its block, reduces on
 * the device (cdkf_ll_sum_*_dev, cdkf_grad_sum_*_dev) and the 1 (+ n_theta) doubles are summed in place over RCCL / xGMI
 * (ncclAllReduce, double, sum) on the same stream -- no host round trip between the sweep and the reduced sum.  RCCL is loaded at
 * first use; without it these return CDKF_EUNSUPPORTED and everything else works. */
typedef struct cdkf_comm cdkf_comm;
#define CDKF_COMM_ID_BYTES 128 /* sizeof(ncclUniqueId) */
/* one process per GPU: rank 0 makes the id, the ranks exchange it (cdkf_rdv_broadcast below, or any channel of the caller's),
 * every rank joins with the device it sweeps on */
/* (before any of it: what this rank can check alone -- RCCL loads and resolves, `device` is one of the visible devices.  No collective
 * inside; the ranks agree on the outcomes over the rendezvous and enter ncclCommInitRank, whose bootstrap has no timeout, only if all
 * passed.  Replaces nothing in the reference: it runs one device.) */
int cdkf_comm_preflight(int device);
int cdkf_comm_unique_id(void* id /* [CDKF_COMM_ID_BYTES] */);
int cdkf_comm_init_rank(cdkf_comm** comm, const void* id, int rank, int world, int device);
/* one process driving ndev GPUs: comms[i] lives on devices[i] (NULL: 0 .. ndev-1) */
int cdkf_comm_init_all(cdkf_comm** comms /* [ndev] */, int ndev, const int* devices);
int cdkf_comm_rank(const cdkf_comm* comm);
int cdkf_comm_world(const cdkf_comm* comm);
/* sums[0 .. count) (device memory) <- the sum over all ranks, in place, enqueued on `stream` */
int cdkf_ll_allreduce(const cdkf_comm* comm, double* sums, int64_t count, void* stream);
/* the same with max (the wall-clock maximum over ranks that a benchmark reports) */
int cdkf_comm_allreduce_max(const cdkf_comm* comm, double* values, int64_t count, void* stream);
/* single-process form: one grouped call for the ndev communicators of cdkf_comm_init_all */
int cdkf_ll_allreduce_all(cdkf_comm* const* comms, int ndev, double* const* sums, int64_t count, void* const* streams);
int cdkf_comm_destroy(cdkf_comm* comm);

/* Host-only rendezvous for process-per-GPU launches (a star over TCP, rank 0 listening on addr:port): hands the RCCL id around
 * and sums / maximises a few host doubles (rank order: deterministic).  Needs no GPU. */
typedef struct cdkf_rdv cdkf_rdv;
int cdkf_rdv_create(cdkf_rdv** rdv, const char* addr, int port, int rank, int world, int timeout_ms);
int cdkf_rdv_broadcast(cdkf_rdv* rdv, void* buf, int64_t bytes); /* from rank 0 */
int cdkf_rdv_allreduce(cdkf_rdv* rdv, double* values, int64_t count, int op /* 0 sum, 1 max */);
int cdkf_rdv_barrier(cdkf_rdv* rdv);
int cdkf_rdv_destroy(cdkf_rdv* rdv);

/* ---- measurement plumbing ------------------------------------------------------------------------------------------------------
 * Name of the sweep kernel the calling thread's last filter / smoother / gradient call launched (the one a profile of that call is
 * dominated by), e.g. "filter_lpe_l63_kernel<double, 3, 1, true>"; "" before the first call.  bench.py matches it against the
 * kernel names of the committed rocprofv3 summaries. */
const char* cdkf_last_kernel(void);
/* HIP events for timing on a stream of the caller's: create, record on `stream`, elapsed milliseconds between two recorded events
 * (synchronises on the second), destroy.  Streams: create / destroy a non-blocking stream. */
int cdkf_event_create(void** event);
int cdkf_event_record(void* event, void* stream);
int cdkf_event_elapsed_ms(void* start, void* stop, float* ms);
int cdkf_event_destroy(void* event);
int cdkf_stream_create(void** stream);
int cdkf_stream_destroy(void* stream);
int cdkf_set_device(int device);

#ifdef __cplusplus
}
#endif
#endif /* CDKF_H */
