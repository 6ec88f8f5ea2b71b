"""The compiler defect behind rounds 3 / 4's "-O1 fences" (launch_custom.hip: rtc_policy; NOTES.md R5.1), held by a reproducer and a
canary.  ROCm 7.2's compiler miscompiles spill-heavy double-precision kernels at the register limit at -O2 / -O3 (located in round 5:
vector spill code placed in front of an execution-mask restore, profiles/r05_j_root_cause.txt); the smallest member of
the family is the forward-sensitivity sweep of NL_F of tests/test_custom_drift.py (its generated translation unit is ~90 lines on top of
the kernel headers: CDKF_CUSTOM_DUMP writes it; scripts/r5_o3_probe.py and scripts/r5_mir_delta.py are the drivers of the investigation).
  * under the shipped policy (-O3, rebuilt at -O1 when the machine code shows the defect's shape or reports more than 300 spilled vector
    registers: this kernel's -O3 build does both) the gradient equals finite differences of the oracle;
  * CANARY: under plain -O3 (CDKF_RTC_POLICY=o3) it is still WRONG on this toolchain -- the day a ROCm release makes this leg pass, the
    test fails with the message to revisit the rule;
  * -O1 (the round-4 fence, CDKF_RTC_POLICY=o1) is right, as it always was.
Every leg runs in a process of its own (hipRTC freezes the first compilation's -mllvm options for the whole process)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _case(policy, tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("CDKF_RTC_POLICY", "CDKF_RTC_EXTRA_OPTS", "CDKF_RTC_EXTRA_OPTS_ONLY", "CDKF_RTC_UNIFORM")}
    if policy:
        env["CDKF_RTC_POLICY"] = policy
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "r5_o3_probe.py"), "case", "d2grad"], capture_output=True, text=True,
                       env=env, timeout=900)
    res = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")]
    assert res, p.stdout[-2000:] + p.stderr[-2000:]
    return res[-1].split()[1] == "GOOD", res[-1]


def test_shipped_policy_is_right_and_the_old_fence_too(hip_lib, tmp_path):
    for policy in (None, "o1"):
        ok, line = _case(policy, tmp_path)
        assert ok, (policy, line)


def test_canary_plain_o3_is_still_miscompiled_on_this_toolchain(hip_lib, tmp_path):
    ok, line = _case("o3", tmp_path)
    assert not ok, ("plain -O3 now compiles the forward-sensitivity sweep correctly on this toolchain: the spill-limit rule "
                    "(launch_custom.hip rtc_policy, csrc/Makefile launch_wg8.o) can be retired -- rerun scripts/r5_o3_probe.py scan ukf15 and "
                    "the d = 46 case first. " + line)
