"""Round 5: register / spill figures of the run-time compiled register-resident kernels under each build policy (no GPU needed: hipRTC
cross-compiles for gfx950).  For every (drift, algorithm, policy): VGPRs, AGPRs, spilled VGPRs / SGPRs and scratch bytes per lane, read from
the code object's metadata (llvm-readelf --notes).  Output: a table on stdout (profiles/r05_b_spill_table.txt).
  python scripts/r5_spill_table.py [policy ...]        default policies: o1 o3 auto (auto = the shipped rule; -mllvm policies only as a process's first compile)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def notes(co):
    out = subprocess.run([READELF, "--notes", co], capture_output=True, text=True).stdout
    g = lambda k: (re.findall(r"\.%s:\s+(\d+)" % k, out) or ["?"])[0]
    return {k: g(k) for k in ("vgpr_count", "agpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size")}


def main():
    from cd_dynamax_amd import _ffi
    from test_custom_drift import NL_F, L63_F
    quad = lambda d: " ".join("fx[%d] = -theta[0] * x[%d] + theta[1] * x[%d] * x[%d] - pow(x[%d], 2) * theta[2];" % (i, i, (i + 1) % d, (i + 2) % d, (i + 3) % d)
                              for i in range(d))
    drifts = [("NL d=2", 2, 3, NL_F), ("L63 d=3", 3, 3, L63_F), ("quad d=4", 4, 3, quad(4)), ("quad d=6", 6, 3, quad(6))]
    algos = [(0, "ekf filter"), (1, "ukf filter"), (2, "ekf filter + smoother"), (3, "forward-sensitivity gradient")]
    pols = sys.argv[1:] or ["o1", "o3", "auto"]
    print("%-10s %-30s %-9s %5s %5s %7s %7s %9s" % ("drift", "kernel", "policy", "vgpr", "agpr", "v-spill", "s-spill", "scratch B"))
    for name, d, nth, src in drifts:
        kind = _ffi.register_custom_drift(d, nth, src, None, None)
        for algo, what in algos:
            for pol in pols:
                os.environ["CDKF_RTC_POLICY"] = pol
                if pol == "auto":   # the shipped rule: -O3, rebuilt at -O1 past the spill limit
                    del os.environ["CDKF_RTC_POLICY"]
                work = tempfile.mkdtemp(prefix="r5_spill_")
                os.environ["CDKF_CUSTOM_DUMP"] = work
                rc = _ffi.lib().cdkf_custom_drift_compile(kind, 8, min(d, 2), algo, 1, 0)
                del os.environ["CDKF_CUSTOM_DUMP"]
                if rc:
                    print(name, what, pol, "compile failed:", _ffi.lib().cdkf_last_error().decode()[:200])
                    continue
                cos = sorted((os.path.getmtime(os.path.join(work, f)), f) for f in os.listdir(work) if f.endswith(".co"))
                n = notes(os.path.join(work, cos[-1][1]))   # (algo 2 compiles two kernels: the last one is the backward sweep)
                print("%-10s %-30s %-9s %5s %5s %7s %7s %9s" % (name, what, pol, n["vgpr_count"], n["agpr_count"], n["vgpr_spill_count"], n["sgpr_spill_count"],
                                                              n["private_segment_fixed_size"]), flush=True)


if __name__ == "__main__":
    main()
