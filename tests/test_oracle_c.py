"""The C restatement (oracle/cdkf_oracle.c, the timed CPU baseline) agrees with the NumPy oracle.  CPU only."""
import numpy as np
import pytest

import cdkf_oracle as o
import cdkf_oracle_c as oc
from helpers import FILTER_KEYS, linear_model, relerr


@pytest.mark.parametrize("case", ["l63_m3", "l63_m1", "lin_2_6", "lin_4_2", "l96_6"])
def test_c_oracle_matches_numpy_oracle(case):
    rng = np.random.default_rng(abs(hash(case)) % 1000)
    if case.startswith("l63"):
        mdl = o.lorenz63_model(int(case[-1]))
    elif case.startswith("lin"):
        _, d, m = case.split("_")
        mdl = linear_model(rng, int(d), int(m))
    else:
        d = 6
        mdl = o.Model(o.Lorenz96Drift(8.0), np.eye(d), 0.5 * np.eye(d), np.eye(d)[::2], np.zeros(3), np.eye(3),
                      8.0 * np.ones(d), np.eye(d))
    N, T = 5, 40
    t = o.irregular_times(rng, N, T, 0.5)
    t[2, 10] = t[2, 9]
    y = o.simulate(mdl, t, rng)
    for order, it in (("second", 1), ("first", 2), ("zeroth", 1)):
        ref = o.ekf_filter(mdl, t, y, state_order=order, num_iter=it, cov_rescaling=0.8)
        got = oc.ekf_filter(mdl, t, y, state_order=order, num_iter=it, cov_rescaling=0.8, nthreads=2)
        assert relerr(got["marginal_loglik"], ref["marginal_loglik"]) < 1e-11
        for k in FILTER_KEYS:
            assert relerr(got[k], ref[k]) < 1e-11, (order, k)
    ref32 = o.ekf_filter(mdl, t, y, dtype=np.float32)
    got32 = oc.ekf_filter(mdl, t, y, dtype=np.float32)
    assert got32["filtered_means"].dtype == np.float32
    assert relerr(got32["filtered_means"], ref32["filtered_means"]) < 2e-4


def test_c_oracle_fp32_known_answer():
    """Same reference constants as tests/test_oracle.py::test_dopri5_known_answer_constants_fp32, through the C
    filter: with R huge the update is a no-op, so predicted moments after one unit interval are A m0 and
    A P0 A^T + Q."""
    mdl = o.Model(o.LinearDrift(-0.1 * np.eye(2), np.zeros(2)), 0.5 * np.eye(2), 0.5 * np.eye(2), np.eye(2),
                  np.zeros(2), 1e30 * np.eye(2), np.array([1.0, 0.0]), np.zeros((2, 2)))
    t = np.array([[0.0, 1.0]])
    y = np.zeros((1, 2, 2))
    r = oc.ekf_filter(mdl, t, y, state_order="first", dtype=np.float32, dt_final=1.0)
    assert r["predicted_means"][0, 0, 0] == np.float32(0.9048373699188232421875)
    assert r["predicted_covariances"][0, 0, 0, 0] == np.float32(0.11329327523708343505859375)
