"""Round 5: which compiler pass flips the run-time compiled kernels that are right at -O1 and wrong at -O3.

  python scripts/r5_o3_probe.py case <name>          one run of the case under the current environment: prints RESULT GOOD|BAD ...
  python scripts/r5_o3_probe.py scan <name>          -O1, -O3 and a few single-switch variants of -O3, each in a child process
  python scripts/r5_o3_probe.py bisect <name>        LLVM -opt-bisect-limit binary search on the ONE kernel the case names (the other
                                                     kernels of the run stay as shipped: CDKF_RTC_EXTRA_OPTS_ONLY), then the pass name

Cases (each: the kernel's tag for CDKF_RTC_EXTRA_OPTS_ONLY, and a check against the oracle):
  d2grad  forward-sensitivity sweep of the d = 2 drift of tests/test_custom_drift.py (NL_F: sin, exp, tanh, cos, sqrt, pow(theta, 2)):
          a zero gradient column at -O3 (round 4, the night the a^2 of cdkf_dual.h changed shape) -- the SMALL member of the family
  fs6     forward-sensitivity sweep of a random quadratic d = 6 drift with pow(x, 2) (gpu_fuzz_custom.py 40404 case 7)
  ukf15   unscented filter of a d = 15 source drift on the workgroup kernel (gpu_fuzz_custom.py 62626 case 11)
Output under gpurun_out/r5_o3/.  Runs on the GPU box (gpurun -- 'python scripts/r5_o3_probe.py bisect d2grad')."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
OUT = os.path.join(ROOT, "gpurun_out", "r5_o3")

TAGS = {"d2grad": "reg ukf=0 algo=2", "fs6": "reg ukf=0 algo=2", "ukf15": "true, cdkf::kDriftAny"}


def case_d2grad():
    import numpy as np
    import cdkf_oracle as o
    import cd_dynamax_amd as cd
    from test_custom_drift import NL_F, make_model, params_for
    rng = np.random.default_rng(90)
    theta = np.array([1.7, 0.25, 0.4])

    def f_np(x, thv):
        s, e = np.sin(x[..., 0]), np.exp(-thv[2] * x[..., 1] ** 2)
        return np.stack([x[..., 1] + thv[1] * np.tanh(x[..., 0] * x[..., 1]),
                         -thv[0] * s * e - thv[1] * x[..., 1] + 0.3 * np.cos(2 * x[..., 0]) / (1 + x[..., 0] ** 2)
                         + np.sqrt(1 + x[..., 1] ** 2) * thv[2] ** 2], -1)

    def jac_np(x, thv, h=1e-6):
        return np.stack([(f_np(x + h * np.eye(2)[j], thv) - f_np(x - h * np.eye(2)[j], thv)) / (2 * h) for j in range(2)], -1)

    mdl = make_model(o.CallableDrift(theta, f_np, jac_np, None), 1)
    N, T = 2, 10   # (small: the oracle's finite differences dominate a probe's time, and the wrong builds are wrong everywhere)
    t = o.irregular_times(rng, N, T, 0.15)
    y = o.simulate(mdl, t, rng)

    def ll_of(th):
        return o.ekf_filter(make_model(o.CallableDrift(th, f_np, jac_np, None), 1), t, y, state_order="first")["marginal_loglik"]
    fd = np.stack([(ll_of(theta + 1e-5 * np.eye(3)[p]) - ll_of(theta - 1e-5 * np.eye(3)[p])) / 2e-5 for p in range(3)], -1)
    P = params_for(mdl, cd.LearnableCustomDrift(theta, NL_F, None, None))
    ll, g = cd.cdnlgssm_loglik_and_grad(P, y, t[..., None], cd.EKFHyperParams(state_order="first"))
    g = np.asarray(g.theta)
    err = np.abs(g - fd).max(0) / np.abs(fd).max(0)
    return bool(err.max() < 1e-5), "per-parameter relative error %s; g[0] %s fd[0] %s" % (err.tolist(), g[0].tolist(), fd[0].tolist())


def case_fuzz(seed, cases, only, what):
    env = dict(os.environ, PYTHONUNBUFFERED="1", CDKF_FUZZ_ONLY_CASE=str(only))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gpu_fuzz_custom.py"), str(seed), str(cases)], capture_output=True,
                       text=True, timeout=1500, env=env)
    out = p.stdout + p.stderr
    mism = [ln for ln in out.splitlines() if "MISMATCH" in ln]
    ran = ("'%s'" % what) in out or what in out
    bis = " | ".join(ln for ln in out.splitlines() if ln.startswith("BISECT:"))
    return p.returncode == 0 and not mism and ran, ("; ".join(mism)[:400] or out[-300:].replace("\n", " | ")) + (" | " + bis if bis else "")


CASES = {"d2grad": case_d2grad, "fs6": lambda: case_fuzz(40404, 8, 7, "grad_theta"), "ukf15": lambda: case_fuzz(62626, 12, 11, "ukf")}


def child(name, extra, log):
    env = dict(os.environ, CDKF_RTC_EXTRA_OPTS=extra, CDKF_RTC_EXTRA_OPTS_ONLY=TAGS[name])
    with open(log, "w") as f:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "case", name], stdout=subprocess.PIPE, stderr=f, text=True, env=env, timeout=3000)
    res = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")]
    line = res[-1] if res else "RESULT BAD (no result line, rc %d) %s" % (p.returncode, p.stdout[-200:].replace("\n", " | "))
    with open(log, "a") as f:   # (a case that runs a grandchild -- the fuzz replays -- reports the compiler's stderr in its result line)
        f.write("\n" + line.replace(" | ", "\n") + "\n")
    return line.split()[1] == "GOOD", line


def main():
    mode, name = sys.argv[1], sys.argv[2]
    if mode == "case":
        ok, detail = CASES[name]()
        print("RESULT", "GOOD" if ok else "BAD", detail, flush=True)
        return
    os.makedirs(OUT, exist_ok=True)
    report = open(os.path.join(OUT, f"{name}_{mode}.txt"), "a")

    def say(*a):
        s = " ".join(str(x) for x in a)
        print(s, flush=True)
        report.write(s + "\n")
        report.flush()
    if mode == "scan":
        # (round 5, first pass: -no-stack-coloring, -no-stack-slot-sharing, -amdgpu-spill-vgpr-to-agpr=0, -amdgpu-spill-sgpr-to-vgpr=0 and a
        #  256-register budget without accumulator registers all stay BAD; the BASIC register allocator and -amdgpu-function-calls=0 are GOOD
        #  on the unscented workgroup kernel: the suspects below are the greedy allocator's features)
        variants = ["-O1", "-O3", "-O3 -mllvm -vgpr-regalloc=basic -mllvm -sgpr-regalloc=basic", "-O3 -mllvm -vgpr-regalloc=basic",
                    "-O3 -mllvm -split-threshold-for-reg-with-hint=0", "-O3 -mllvm -amdgpu-dce-in-ra=0", "-O3 -mllvm -enable-subreg-liveness=0",
                    "-O3 -mllvm -join-liveintervals=0", "-O3 -mllvm -split-spill-mode=size", "-O3 -mllvm -disable-spill-fusing",
                    "-O3 -mllvm -enable-deferred-spilling", "-O3 -mllvm -amdgpu-enable-rewrite-partial-reg-uses=0", "-O3 -mllvm -enable-misched=0",
                    "-O3 -mllvm -amdgpu-waitcnt-forcezero"]
        variants = sys.argv[3:] or variants
        for k, v in enumerate(variants):
            ok, line = child(name, v, os.path.join(OUT, f"{name}_scan_{k}.err"))
            say("[%s]" % v, line[:600])
        return
    # bisect: a limit of -1 is "no limit"; find the smallest N whose build is BAD given that N - 1 is GOOD
    base = sys.argv[3] if len(sys.argv) > 3 else "-O3"
    ok_all, line = child(name, base, os.path.join(OUT, f"{name}_bis_all.err"))
    say("[%s, no limit]" % base, line[:400])
    if ok_all:
        say("the unlimited build is GOOD: nothing to bisect")
        return
    log0 = os.path.join(OUT, f"{name}_bis_count.err")
    ok0, line = child(name, base + " -mllvm -opt-bisect-limit=0", log0)   # (every optional pass is listed as NOT running: the count)
    total = 0
    for ln in open(log0, errors="replace"):
        for m in re.finditer(r"BISECT: (?:NOT )?running pass \((\d+)\)", ln):
            total = max(total, int(m.group(1)))
    say("optional passes:", total)
    lo, hi = 0, total   # invariant: limit lo GOOD, limit hi BAD
    say("[limit 0]", line[:300])
    if not ok0:
        say("limit 0 is already BAD: not an optional pass")
        return
    while hi - lo > 1:
        mid = (lo + hi) // 2
        ok, line = child(name, base + f" -mllvm -opt-bisect-limit={mid}", os.path.join(OUT, f"{name}_bis_{mid}.err"))
        say("[limit %d]" % mid, line[:200])
        lo, hi = (mid, hi) if ok else (lo, mid)
    say("first BAD limit:", hi)
    for ln in open(os.path.join(OUT, f"{name}_bis_count.err"), errors="replace"):
        if re.match(r"BISECT: (?:NOT )?running pass \((%d|%d)\)" % (hi - 1, hi), ln):
            say(ln.strip()[:300])


if __name__ == "__main__":
    main()
