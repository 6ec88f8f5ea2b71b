"""A fixed-seed slice of the random-problem fuzzers (scripts/gpu_fuzz_*.py) inside the GPU suite.

The scripts compare the Python surface with the oracle on random drifts, shapes, emission matrices, orders, solver settings, batch sizes
and layouts across every kernel family; in round 3 they found five defects the fixed-shape tests had not.  They are
development aids with run-dependent seeds; here each runs ONCE with a pinned seed and a small case count -- about forty cases in all --
as a child process (its own seeds, its own registered drifts), and a single `MISMATCH` line, a status flag or a non-zero exit fails the
test.  Reference behaviour under test: inference_ekf.py:46-326, 363-539; inference_ukf.py:45-308; ssm_temissions.py:550-568."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (script, seed, cases): seeds chosen once and kept; together 53 cases (round 5: the tangent sweeps' fuzzer, and the models whose
# emission is given as source above six dimensions)
SLICE = [("filters", 20261, 9), ("grads", 20262, 6), ("batches", 20263, 6), ("solvers", 20264, 6), ("misc", 20265, 5),
         ("r03", 20266, 4), ("custom", 20267, 4), ("tangent", 20268, 8), ("wide_emission", 20269, 5)]


# cases the fuzzers caught with run-dependent seeds, replayed alone (the generator's draws in its order, the other cases skipped):
# round 4 -- custom drifts with pow(x, 2): the forward-sensitivity sweep at d = 6 (seed 40404, case 7) and the unscented filter on the
# workgroup kernel at d = 15 (seed 62626, case 11), both wrong at -O3 and right at -O1 (launch_custom.hip builds them at -O1 since)
REPLAY = [("custom", 40404, 8, 7), ("custom", 62626, 12, 11)]


@pytest.mark.parametrize("script,seed,cases,only", REPLAY, ids=[f"{s[0]}-{s[1]}-{s[3]}" for s in REPLAY])
def test_fuzz_case_replayed(hip_lib, script, seed, cases, only):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", f"gpu_fuzz_{script}.py"), str(seed), str(cases)],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, PYTHONUNBUFFERED="1", CDKF_FUZZ_ONLY_CASE=str(only)))
    out = p.stdout + p.stderr
    assert p.returncode == 0, out[-3000:]
    assert "MISMATCH" not in out, out[-3000:]
    assert f"seed {seed}" in p.stdout and "'ukf'" in p.stdout or "'grad_theta'" in p.stdout, out[-1500:]   # the case ran its checks


@pytest.mark.parametrize("script,seed,cases", SLICE, ids=[s[0] for s in SLICE])
def test_fuzz_slice(hip_lib, script, seed, cases):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", f"gpu_fuzz_{script}.py"), str(seed), str(cases)],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, PYTHONUNBUFFERED="1"))
    out = p.stdout + p.stderr
    assert p.returncode == 0, out[-3000:]
    assert "MISMATCH" not in out, out[-3000:]
    assert f"seed {seed}" in p.stdout and f"cases {cases}" in p.stdout, out[-1500:]   # the summary line: the run reached its end
