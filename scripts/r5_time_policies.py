"""A/B of the run-time compiled kernels' build policies (launch_custom.hip: rtc_policy) and of the two builds of launch_wg8.o: each
leg in a child process; prints ms per call.  gpurun -- 'python scripts/r5_time_policies.py'"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORK = r'''
import sys, os, time
sys.path[:0] = [os.environ["ROOT"], os.path.join(os.environ["ROOT"], "oracle"), os.path.join(os.environ["ROOT"], "tests")]
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from helpers import lorenz96_model, params_from, random_quadratic_drift
what = sys.argv[1]
rng = np.random.default_rng(3)
def timed(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e3
if what == "wg8":
    os.environ["CDKF_NO_WAVE40"] = "1"
    d = 48
    mdl = lorenz96_model(d, d)
    N, T = 256, 40
    t = o.irregular_times(rng, N, T, 0.005 * T); y = 8.0 + rng.standard_normal((N, T, d))
    P = params_from(mdl)
    ms = timed(lambda: cd.cdnlgssm_filter(P, y, t[..., None], output_fields=[]))
    ref = o.ekf_filter(mdl, t[:2], y[:2])["marginal_loglik"]
    got = cd.cdnlgssm_filter(P, y[:2], t[:2, :, None], output_fields=[]).marginal_loglik
    print("RESULT wg8 d=48 fp64 N=256 T=40 filter %.1f ms, ll rel err %.2e (%s)" % (ms, np.abs(got - ref).max() / np.abs(ref).max(), cd._ffi.lib().cdkf_last_kernel().decode()[:50]))
else:
    d = int(what[3:])
    for seed in range(50):
        r2 = np.random.default_rng(100 + seed)
        src, make = random_quadratic_drift(r2, d)
        if "pow(" in src: break
    theta = np.array([0.7, -0.15])
    m = max(1, d // 2)
    mdl = o.Model(make(theta), np.eye(d), 0.3 * np.eye(d), rng.standard_normal((m, d)) / np.sqrt(d), np.zeros(m), 0.5 * np.eye(m), 0.5 * rng.standard_normal(d), 0.3 * np.eye(d))
    N, T = 512, 100
    t = o.irregular_times(rng, N, T, 0.005 * T); y = rng.standard_normal((N, T, m))
    P = cd.ParamsCDNLGSSM(initial=cd.ParamsLGSSMInitial(cd.LearnableVector(mdl.m0), cd.LearnableMatrix(mdl.P0)),
        dynamics=cd.ParamsCDNLGSSMDynamics(cd.LearnableCustomDrift(theta, src, None, ""), cd.LearnableMatrix(mdl.L), cd.LearnableMatrix(mdl.Qc), 2.0),
        emissions=cd.ParamsCDNLGSSMEmissions(cd.LearnableLinear(mdl.H, mdl.bias), cd.LearnableMatrix(mdl.R)))
    hyp = cd.EKFHyperParams(state_order="first")
    out = []
    out.append("ekf %.1f" % timed(lambda: cd.cdnlgssm_filter(P, y, t[..., None], hyp, output_fields=[])))
    out.append("ukf %.1f" % timed(lambda: cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams(), output_fields=[])))
    out.append("grad %.1f" % timed(lambda: cd.cdnlgssm_loglik_and_grad(P, y, t[..., None], hyp)))
    out.append("grad_all %.1f" % timed(lambda: cd.cdnlgssm_loglik_and_grad_all(P, y, t[..., None], hyp)))
    ref = o.ekf_filter(mdl, t[:2], y[:2], "first")["marginal_loglik"]
    got = cd.cdnlgssm_filter(P, y[:2], t[:2, :, None], hyp, output_fields=[]).marginal_loglik
    print("RESULT custom d=%d fp64 N=512 T=100 ms: %s | ll rel err %.2e" % (d, ", ".join(out), np.abs(got - ref).max() / np.abs(ref).max()))
'''
env0 = dict(os.environ, ROOT=ROOT, CDKF_RTC_CACHE_DIR="/tmp/r5_rtc_cache")
os.makedirs("/tmp/r5_rtc_cache", exist_ok=True)
for what in os.environ.get("R5_WHAT", "rtc4,rtc6,rtc12,rtc24").split(","):
    for pol in (os.environ.get("R5_POLICIES", ",o1").split(",")):
        env = dict(env0)
        if pol: env["CDKF_RTC_POLICY"] = pol
        p = subprocess.run([sys.executable, "-c", WORK, what], env=env, capture_output=True, text=True, timeout=3000)
        print("[policy %s]" % (pol or "shipped: -O3, -O1 past the spill limit"), ([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")] or [p.stderr[-400:]])[-1], flush=True)
for lib in ([] if os.environ.get("R5_SKIP_WG8") else ["", os.path.join(ROOT, "build", "alt", "libcdkf_hip_wg8basic.so")]):
    env = dict(env0)
    if lib: env["CDKF_LIB_PATH"] = lib
    p = subprocess.run([sys.executable, "-c", WORK, "wg8"], env=env, capture_output=True, text=True, timeout=3000)
    print("[launch_wg8.o %s]" % ("-O3 basic allocator" if lib else "-O1 (shipped)"), ([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")] or [p.stderr[-400:]])[-1], flush=True)
