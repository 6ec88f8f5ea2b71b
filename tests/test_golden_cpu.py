"""The committed golden vectors are reproduced by the oracle (guards against silent oracle drift), and the
linear fixtures agree with the exact closed-form Kalman filter (ties the fixtures to an oracle-independent
answer).  CPU only."""
import numpy as np
import pytest

import cdkf_oracle as o
from helpers import FILTER_KEYS, GOLDEN, GOLDEN_WIDE, closed_form_kf, load_golden, model_from_fixture, relerr


@pytest.mark.parametrize("name", GOLDEN)
def test_oracle_reproduces_golden(name):
    g = load_golden(name)
    mdl = model_from_fixture(g)
    s = int(g["stride"])
    order = "second"
    r = o.ekf_filter(mdl, g["t"], g["y"], state_order=order, dt_final=float(g["dt_final"]))
    np.testing.assert_allclose(r["marginal_loglik"], g[f"ekf_{order}_ll"], rtol=1e-12)
    for k in FILTER_KEYS:
        np.testing.assert_allclose(r[k][:, ::s], g[f"ekf_{order}_{k}"], rtol=1e-11, atol=1e-13)
    r = o.ukf_filter(mdl, g["t"], g["y"], dt_final=float(g["dt_final"]))
    np.testing.assert_allclose(r["marginal_loglik"], g["ukf_ll"], rtol=1e-12)
    r = o.ekf_smoother(mdl, g["t"], g["y"], dt_final=float(g["dt_final"]))
    np.testing.assert_allclose(r["smoothed_means"][:, ::s], g["eks_smoothed_means"], rtol=1e-11, atol=1e-13)


@pytest.mark.parametrize("name", GOLDEN[:2])
def test_linear_golden_equals_closed_form(name):
    g = load_golden(name)
    mdl = model_from_fixture(g)
    s = int(g["stride"])
    ref = closed_form_kf(mdl, g["t"][0], g["y"][0], dt_final=float(g["dt_final"]))
    for algo in ("ekf_first", "ekf_second", "ukf"):
        for k in FILTER_KEYS:
            assert relerr(g[f"{algo}_{k}"][0], ref[k][::s]) < 1e-7, (algo, k)
        assert abs(g[f"{algo}_ll"][0] - ref["marginal_loglik"]) < 1e-6 * abs(ref["marginal_loglik"])


@pytest.mark.parametrize("name", GOLDEN_WIDE)
def test_oracle_reproduces_wide_golden(name):
    g = load_golden(name)
    mdl = model_from_fixture(g)
    orders = [k[4:-3] for k in g.files if k.startswith("ekf_") and k.endswith("_ll")]
    for order in orders:
        r = o.ekf_filter(mdl, g["t"], g["y"], state_order=order)
        np.testing.assert_allclose(r["marginal_loglik"], g[f"ekf_{order}_ll"], rtol=1e-12)
        np.testing.assert_allclose(r["filtered_means"], g[f"ekf_{order}_filtered_means"], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(r["predicted_covariances"][:, -1], g[f"ekf_{order}_predicted_cov_last"], rtol=1e-10,
                                   atol=1e-12)
