"""The device templates compiled for the HOST and run under the CPU sanitizers (VERDICT r4 "do this" 1a): the same headers the GPU
runs -- cdkf_reg_kernels.h / cdkf_grad_kernels.h / cdkf_dual.h behind a run-time generated drift source, the workgroup kernels of
cdkf_wg2_kernels.h -- built with clang++ for x86-64 (cd_dynamax_amd/csrc/hostsim/cdkf_hostsim.h: a thread per GPU thread, barriers
ThreadSanitizer understands, the matrix instruction and the cross-lane reads as exchanges) and driven with the argument blocks the
library's own launcher forms (cdkf_debug_custom_reg_blob, cdkf_debug_wg_args).  Every case is one that returned wrong numbers at
-O2 / -O3 on the GPU in rounds 3 / 4 and is fenced with -O1: an out-of-bounds private-array index, an
uninitialised read, a signed overflow or a missing barrier in the repository's own code would show here; the outputs are compared
with the oracle as well.  Reference behaviour at stake: inference_ekf.py:46-199, inference_ukf.py:93-203, ssm_temissions.py:550-568."""
import ctypes as C
import os
import shutil
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import cdkf_oracle as o  # noqa: E402
import cd_dynamax_amd as cd  # noqa: E402
from cd_dynamax_amd import _ffi, models  # noqa: E402
import hostsim_util as hs  # noqa: E402
from helpers import lorenz96_model, params_from, random_quadratic_drift, relerr  # noqa: E402

pytestmark = pytest.mark.skipif(hs.clang() is None or shutil.which("hipcc") is None, reason="needs clang++ and the HIP library (hipRTC)")

FILTER_KEYS = ("marginal_loglik", "filtered_means", "filtered_covariances", "predicted_means", "predicted_covariances")


def _custom_params(mdl, drift):
    return cd.ParamsCDNLGSSM(
        initial=cd.ParamsLGSSMInitial(cd.LearnableVector(mdl.m0), cd.LearnableMatrix(mdl.P0)),
        dynamics=cd.ParamsCDNLGSSMDynamics(drift, cd.LearnableMatrix(mdl.L), cd.LearnableMatrix(mdl.Qc), 2.0),
        emissions=cd.ParamsCDNLGSSMEmissions(cd.LearnableLinear(mdl.H, mdl.bias), cd.LearnableMatrix(mdl.R)))


def _pow_drift_model(d, m, seed0=5000):
    """tests/test_custom_drift.py::test_forward_sensitivities_of_a_source_drift_with_powers' problem: the first random quadratic drift
    whose source squares a component through pow()."""
    for seed in range(400):
        rng = np.random.default_rng(seed0 + seed)
        src, make = random_quadratic_drift(rng, d)
        if "pow(" in src:
            break
    theta = np.array([0.7, -0.15])
    A = rng.standard_normal((d, d))
    B = rng.standard_normal((m, m))
    mdl = o.Model(make(theta), np.eye(d) + 0.1 * rng.standard_normal((d, d)), A @ A.T / d * 0.3 + 0.3 * np.eye(d),
                  rng.standard_normal((m, d)) / np.sqrt(d), 0.1 * rng.standard_normal(m), B @ B.T / m * 0.5 + 0.3 * np.eye(m),
                  0.5 * rng.standard_normal(d), 0.3 * np.eye(d))
    return rng, src, theta, mdl


FULL = os.environ.get("CDKF_HOSTSIM_FULL") == "1"   # the slowest legs (minutes of sanitizer-instrumented compilation) only on request


@pytest.mark.parametrize("san", ["asan", "msan"])
@pytest.mark.parametrize("d,m", [(6, 1), (2, 1)])
def test_forward_sensitivity_sweep_of_a_source_drift_on_the_host(san, d, m):
    if d == 6 and not FULL:
        pytest.skip("d = 6 under the sanitizers takes 1 - 2 minutes to compile: CDKF_HOSTSIM_FULL=1 (clean in round 5: NOTES.md R5.1)")
    """ekf_grad_reg_body + nested dual numbers behind a generated drift source (the d = 6 instantiation: wrong d ll / d theta at
    -O2 / -O3 on the GPU, flipping between spellings of x^2; the d = 2 one: a zero column): clean under ASan + UBSan and under MSan
    (the outputs written to a file: an uninitialised value reaching one would be reported), gradient equal to the oracle's."""
    rng, src, theta, mdl = _pow_drift_model(d, m)
    N, T = 3, 6
    t = o.irregular_times(rng, N, T, 0.03 * T)
    y = o.simulate(mdl, t, rng)
    ll_ref, g_ref = o.ekf_loglik_grad_adjoint(mdl, t, y, state_order="first")
    mb = models._model_block(_custom_params(mdl, cd.LearnableCustomDrift(theta, src, None, None)))
    opts = models._opts(cd.EKFHyperParams(state_order="first"))
    opts.layout, opts.layout_in, opts.t_shared = _ffi.LAYOUT_TCN, _ffi.LAYOUT_NT, 0
    ddir, srcs = hs.dump_custom_source(mb.c.drift_kind, 8, m, 3, 1)
    ll, g, *_ = hs.reg_run(os.path.join(ddir, srcs[0]), mb, opts, t, y, 3, np.float64, san, (N, N * 2, 0, 0, 0, N, 0, 0))
    np.testing.assert_allclose(ll, ll_ref, rtol=1e-11)
    assert np.abs(g.reshape(N, 2) - g_ref).max() < 1e-9 * np.abs(g_ref).max()


def test_adaptive_source_drift_reads_its_step_bounds_on_the_host():
    """Round 5 finding of this build: the run-time compiled register kernels never received PIDController's dtmin / dtmax (two fields of
    the argument struct the generated `unpack` left unset).  Under MSan the adaptive sweep of a source drift must run clean, and dtmax
    must bind: with dtmax below the observation gaps the log-likelihood equals the oracle's with the same bound."""
    rng, src, theta, mdl = _pow_drift_model(3, 2)
    N, T = 2, 5
    t = o.irregular_times(rng, N, T, 0.08 * T)
    y = o.simulate(mdl, t, rng)
    hyp = cd.EKFHyperParams(state_order="first", diffeqsolve_settings={"solver": "heun", "stepsize_controller": cd.PIDController(rtol=1e-1, atol=1e-2, dtmax=0.01), "dt0": 0.1})
    mb = models._model_block(_custom_params(mdl, cd.LearnableCustomDrift(theta, src, None, None)))
    opts = models._opts(hyp)
    opts.layout, opts.layout_in, opts.t_shared = _ffi.LAYOUT_TCN, _ffi.LAYOUT_NT, 0
    ddir, srcs = hs.dump_custom_source(mb.c.drift_kind, 8, 2, 0 + 16, 1)   # (+ 16: the run-time tableau / controller variant)
    d = 3
    ll, fm, fP, pm, pP, status, *_ = hs.reg_run(os.path.join(ddir, srcs[0]), mb, opts, t, y, 0, np.float64, "msan",
                                                (N, N * T * d, N * T * d * d, N * T * d, N * T * d * d, N, 0, 0))
    with o.use_solver("heun", adaptive=dict(rtol=1e-1, atol=1e-2, dtmax=0.01)):
        ref = o.ekf_filter(mdl, t, y, state_order="first", dt0=0.1)
    with o.use_solver("heun", adaptive=dict(rtol=1e-1, atol=1e-2)):
        ref_free = o.ekf_filter(mdl, t, y, state_order="first", dt0=0.1)
    np.testing.assert_allclose(ll, ref["marginal_loglik"], rtol=1e-10)
    assert np.abs(ref["marginal_loglik"] - ref_free["marginal_loglik"]).max() > 1e-6   # (the bound changes the result: it was read)


def _l96_problem(d, T=3):
    rng = np.random.default_rng(5)
    mdl = lorenz96_model(d, d)
    t = o.irregular_times(rng, 1, T, 0.012 * T)
    y = o.simulate(mdl, t, rng)
    return mdl, t, y


@pytest.mark.parametrize("san", ["asan", "msan", "tsan"])
def test_workgroup_kernel_eight_entries_per_thread_on_the_host(san, monkeypatch):
    if san == "msan" and not FULL:
        pytest.skip("the MemorySanitizer leg of this kernel: CDKF_HOSTSIM_FULL=1 (clean in round 5)")
    """ekf_filter_wg_kernel<double, 8, false, Lorenz-96> at d = 46 -- NaN at -O2 / -O3 on the GPU since round 3 (launch_wg8.hip ships at
    -O1; pass bisection names si-shrink-instructions) -- on the host with 512 threads: no out-of-bounds LDS or private-array access, no
    undefined arithmetic, no uninitialised value reaching an output or a branch, no pair of LDS accesses without a barrier between them;
    1e-12 of the oracle."""
    mdl, t, y = _l96_problem(46)
    ref = o.ekf_filter(mdl, t, y)
    mb = models._model_block(params_from(mdl))
    opts = models._opts(cd.EKFHyperParams(diffeqsolve_settings={"max_steps": 50}))
    if san == "tsan":
        monkeypatch.setenv("HOSTSIM_JITTER", "4")
    out = hs.wg_run(os.path.join(hs.HARNESS, "wg_builtin_tu.h"), mb, opts, t, y, np.float64, san, kind=2)
    assert out["geom"][:2] == (8, 512)
    for k in FILTER_KEYS:
        assert relerr(out[k], ref[k]) < 1e-12, k


def test_thread_sanitizer_reports_a_removed_barrier(monkeypatch):
    """The instrument checked on the kernel it guards: with one of the workgroup's barriers skipped (HOSTSIM_SKIP_BARRIER: the 5th is in
    the Cholesky panel loop, the 60th in a Runge-Kutta stage), ThreadSanitizer names the two LDS accesses that barrier kept apart.  Its
    shadow checks are lock-free, so a single run can miss a pair that happens at the same moment: a few barriers are tried, one report
    is required."""
    mdl, t, y = _l96_problem(46)
    mb = models._model_block(params_from(mdl))
    opts = models._opts(cd.EKFHyperParams(diffeqsolve_settings={"max_steps": 50}))
    monkeypatch.setenv("HOSTSIM_JITTER", "4")
    reported = []
    for skip in (5, 60, 6, 61, 7, 62, 8, 63):
        monkeypatch.setenv("HOSTSIM_SKIP_BARRIER", str(skip))
        try:
            hs.wg_run(os.path.join(hs.HARNESS, "wg_builtin_tu.h"), mb, opts, t, y, np.float64, "tsan", kind=2)
        except AssertionError as e:
            if "ThreadSanitizer: data race" in str(e) and ("wg_cholesky2" in str(e) or "wg_stage" in str(e) or "wg_ekf_update" in str(e)):
                reported.append(skip)
                break
    assert reported, "no skipped barrier was reported"


@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_unscented_workgroup_kernel_of_a_source_drift_on_the_host(san, monkeypatch):
    if san == "tsan" and not FULL:
        pytest.skip("the ThreadSanitizer leg of this kernel: CDKF_HOSTSIM_FULL=1 (clean in round 5)")
    """ekf_filter_wg_kernel<double, 4, true, any> with a d = 15 source drift that squares a component through pow() -- 2-3 % off at -O3
    on the GPU (gpu_fuzz_custom.py 62626 case 11, also on the round-3 library) -- on the host: the generic sigma-point path, a thread
    per sigma-point pair calling the drift's source."""
    d, m = 15, 10
    rng, src, theta, mdl = _pow_drift_model(d, m, seed0=7000)
    N, T = 1, 3
    t = o.irregular_times(rng, N, T, 0.02 * T)
    y = o.simulate(mdl, t, rng)
    ref = o.ukf_filter(mdl, t, y)
    mb = models._model_block(_custom_params(mdl, cd.LearnableCustomDrift(theta, src, None, None)))
    opts = models._opts(cd.UKFHyperParams())
    ddir, srcs = hs.dump_custom_source(mb.c.drift_kind, 8, m, 1, 1)
    if san == "tsan":
        monkeypatch.setenv("HOSTSIM_JITTER", "4")
    out = hs.wg_run(os.path.join(ddir, [s for s in srcs if "_1_0" in s][0]), mb, opts, t, y, np.float64, san, ukf=True, kind=-1)
    for k in FILTER_KEYS:
        assert relerr(out[k], ref[k]) < 1e-9, k
