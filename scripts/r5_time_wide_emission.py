"""Round 5: what the value mode of the tangent kernels costs (filters of a model whose emission is given as source above six dimensions:
a lane per trajectory, state in private memory, dual arithmetic computed and dropped) -- wall time of the host-array entry points after
a warm-up call, Lorenz-96 d = 8 / 12 under the m = 7 / 5 emission of tests/test_wide_emission.py.
gpurun -- 'python scripts/r5_time_wide_emission.py'"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi
from test_wide_emission import wide_problem

for d, m, N, T in ((8, 7, 64, 100), (8, 7, 4096, 100), (8, 7, 16384, 100), (12, 5, 4096, 100)):
    mdl, P, t, y, _ = wide_problem(5, d, m, N, T, span=1.0)
    for name, call in (("ekf 'first'", lambda: cd.cdnlgssm_filter(P, y, t[..., None], cd.EKFHyperParams(state_order="first"))),
                       ("ukf", lambda: cd.cdnlgssm_filter(P, y, t[..., None], cd.UKFHyperParams())),
                       ("ekf smoother", lambda: cd.cdnlgssm_smoother(P, y, t[..., None], cd.EKFHyperParams(state_order="first")))):
        call()
        t0 = time.perf_counter()
        for _ in range(3):
            call()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        print(f"d={d} m={m} N={N} T={T} {name:14s} {ms:9.1f} ms  ({_ffi.lib().cdkf_last_kernel().decode()[:60]})", flush=True)
