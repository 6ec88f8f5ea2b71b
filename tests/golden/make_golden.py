"""Generates the committed golden vectors in tests/golden/*.npz from the CPU oracle.

The JAX reference cannot be imported in the build container, so the vectors
come from oracle/cdkf_oracle.py, which tests/test_oracle.py pins to the reference's own known-answer
constants and test equalities.  Re-run with:  python tests/golden/make_golden.py
Outputs are sub-sampled in time (every STRIDE-th step) to keep the fixtures small.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import cdkf_oracle as o  # noqa: E402

STRIDE = 4
FILTER_KEYS = ["filtered_means", "filtered_covariances", "predicted_means", "predicted_covariances"]


def model_arrays(mdl):
    return dict(drift_kind=mdl.drift.kind, theta=mdl.drift.theta(), L=mdl.L, Qc=mdl.Qc, H=mdl.H, bias=mdl.bias, R=mdl.R,
                m0=mdl.m0, P0=mdl.P0)


def case(name, mdl, t, y, dt_final, orders=("first", "second"), ukf=True, eks=True, extra=None):
    out = dict(model_arrays(mdl), t=t, y=y, dt_final=dt_final, stride=STRIDE)
    if extra:
        out.update(extra)
    for order in orders:
        r = o.ekf_filter(mdl, t, y, state_order=order, dt_final=dt_final)
        out[f"ekf_{order}_ll"] = r["marginal_loglik"]
        for k in FILTER_KEYS:
            out[f"ekf_{order}_{k}"] = r[k][:, ::STRIDE]
    if ukf:
        r = o.ukf_filter(mdl, t, y, dt_final=dt_final)
        out["ukf_ll"] = r["marginal_loglik"]
        for k in FILTER_KEYS:
            out[f"ukf_{k}"] = r[k][:, ::STRIDE]
    if eks:
        r = o.ekf_smoother(mdl, t, y, dt_final=dt_final)
        out["eks_smoothed_means"] = r["smoothed_means"][:, ::STRIDE]
        out["eks_smoothed_covariances"] = r["smoothed_covariances"][:, ::STRIDE]
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, os.path.getsize(path) // 1024, "KiB")


def case_wide(name, mdl, t, y, orders):
    """Large-state fixtures (Lorenz-96 d=40, MLP d=8): inputs, log-likelihoods, all filtered / smoothed means,
    and the covariances of the LAST step only (a d x d block per step would not be a small fixture)."""
    out = dict(model_arrays(mdl), t=t, y=y, dt_final=1e-10)
    if mdl.drift.kind == "mlp":
        out["hidden"] = np.array([mdl.drift.W1.shape[0], mdl.drift.W2.shape[0]])
    for order in orders:
        r = o.ekf_filter(mdl, t, y, state_order=order)
        out[f"ekf_{order}_ll"] = r["marginal_loglik"]
        out[f"ekf_{order}_filtered_means"] = r["filtered_means"]
        out[f"ekf_{order}_filtered_cov_last"] = r["filtered_covariances"][:, -1]
        out[f"ekf_{order}_predicted_cov_last"] = r["predicted_covariances"][:, -1]
    r = o.ekf_smoother(mdl, t, y, state_order=orders[-1])
    out["eks_smoothed_means"] = r["smoothed_means"]
    out["eks_smoothed_cov_first"] = r["smoothed_covariances"][:, 0]
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, os.path.getsize(path) // 1024, "KiB")


def main():
    from helpers import linear_model, lorenz96_model, mlp_model
    # 1. the reference test scripts' shape: STATE_DIM=2, EMISSION_DIM=6, T=100 regular integer times, dt_final=1
    rng = np.random.default_rng(100)
    mdl = linear_model(rng, 2, 6)
    t = np.arange(100, dtype=float)[None]
    case("linear_d2_m6_regular", mdl, t, o.simulate(mdl, t, rng), 1.0)
    # 2. BASELINE config 1: tracking model of cdlgssm_tracking.ipynb (d=4, m=2), regular unit steps
    rng = np.random.default_rng(101)
    F = np.zeros((4, 4))
    F[0, 2] = F[1, 3] = 1.0
    H = np.eye(4)[:2]
    mdl = o.Model(o.LinearDrift(F, np.zeros(4)), np.eye(4), 0.1 * np.eye(4), H, np.zeros(2), 0.5 * np.eye(2),
                  np.array([8.0, 10.0, 1.0, 0.0]), np.eye(4))
    t = np.arange(60, dtype=float)[None]
    case("tracking_d4_m2_regular", mdl, t, o.simulate(mdl, t, rng), 1.0)
    # 3./4. Lorenz-63, irregular per-trajectory grids with long gaps (n_k up to ~5), a duplicated time stamp
    for m_obs, seed in ((3, 102), (1, 103)):
        rng = np.random.default_rng(seed)
        mdl = o.lorenz63_model(m_obs)
        N, T = 3, 80
        t = o.irregular_times(rng, N, T, 0.012 * T)
        t[1, 40] = t[1, 39]  # zero-length interval
        y = o.simulate(mdl, t, rng)
        case(f"lorenz63_m{m_obs}_irregular", mdl, t, y, 1e-10, orders=("second",))
    # 5. BASELINE config 4 shape: Lorenz-96, d = 40, fully observed (N=2, T=20)
    rng = np.random.default_rng(104)
    mdl = lorenz96_model(40, 40)
    t = o.irregular_times(rng, 2, 20, 0.012 * 20)
    case_wide("lorenz96_d40_m40", mdl, t, o.simulate(mdl, t, rng), orders=("second",))
    # 6. BASELINE config 5 shape: MLP(8 -> 64 -> 64 -> 8, tanh) drift, d = 8, m = 4 (N=4, T=50); 'second' carries the
    #    reference's 0.5*trace(H_t @ P) quirk (non-zero here), 'first' does not
    rng = np.random.default_rng(105)
    mdl = mlp_model(rng, 8, 4, 64)
    t = o.irregular_times(rng, 4, 50, 0.02 * 50)
    case_wide("mlp_d8_m4", mdl, t, o.simulate(mdl, t, rng), orders=("first", "second"))


if __name__ == "__main__":
    main()
