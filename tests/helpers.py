"""Shared test helpers: oracle-model -> drop-in params, closed-form linear Kalman filter / RTS smoother
(an oracle-independent cross-check), comparison utilities in the style of the reference's
src/utils/test_utils.py:160-179 (compare_structs tolerance ladder)."""
import numpy as np
import scipy.linalg as sla

import cd_dynamax_amd as cd
import cdkf_oracle as o


def params_from(mdl: o.Model) -> cd.ParamsCDNLGSSM:
    dr = mdl.drift
    if dr.kind == "lorenz63":
        drift = cd.LearnableLorenz63(float(dr.sigma), float(dr.rho), float(dr.beta))
    elif dr.kind == "linear":
        drift = cd.LearnableLinear(dr.W, dr.b)
    elif dr.kind == "lorenz96":
        drift = cd.LearnableLorenz96(float(dr.F))
    elif dr.kind == "mlp":
        drift = cd.LearnableMLP(dr.W1, dr.b1, dr.W2, dr.b2, dr.W3, dr.b3)
    else:
        raise ValueError(dr.kind)
    return cd.ParamsCDNLGSSM(
        initial=cd.ParamsLGSSMInitial(cd.LearnableVector(mdl.m0), cd.LearnableMatrix(mdl.P0)),
        dynamics=cd.ParamsCDNLGSSMDynamics(drift, cd.LearnableMatrix(mdl.L), cd.LearnableMatrix(mdl.Qc), 2.0),
        emissions=cd.ParamsCDNLGSSMEmissions(cd.LearnableLinear(mdl.H, mdl.bias), cd.LearnableMatrix(mdl.R)))


def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))


def assert_close_structs(got: dict, ref: dict, rtol, names=None):
    for k in names or ref.keys():
        if k in got and got[k] is not None:
            e = relerr(got[k], ref[k])
            assert e <= rtol, f"{k}: max-relative error {e:.3e} > {rtol:.1e}"


def linear_model(rng, d, m, stable=True):
    W = -0.5 * np.eye(d) + 0.3 * rng.standard_normal((d, d)) / np.sqrt(d)
    b = 0.2 * rng.standard_normal(d)
    L = np.eye(d) + 0.1 * rng.standard_normal((d, d))
    A = rng.standard_normal((d, d))
    Qc = 0.3 * (A @ A.T / d + np.eye(d))
    H = rng.standard_normal((m, d))
    hb = 0.1 * rng.standard_normal(m)
    B = rng.standard_normal((m, m))
    R = 0.2 * (B @ B.T / m + np.eye(m))
    m0 = rng.standard_normal(d)
    C = rng.standard_normal((d, d))
    P0 = C @ C.T / d + 0.5 * np.eye(d)
    return o.Model(o.LinearDrift(W, b), L, Qc, H, hb, R, m0, P0)


def van_loan(F, LQL, b, dt):
    """Exact discretisation of dx = (F x + b) dt + L dW over dt: x' = A x + c + N(0, Q)."""
    d = F.shape[0]
    Mx = np.zeros((2 * d, 2 * d))
    Mx[:d, :d] = -F
    Mx[:d, d:] = LQL
    Mx[d:, d:] = F.T
    E = sla.expm(Mx * dt)
    A = E[d:, d:].T
    Q = A @ E[:d, d:]
    Ma = np.zeros((d + 1, d + 1))
    Ma[:d, :d] = F
    Ma[:d, d] = b
    c = sla.expm(Ma * dt)[:d, d]
    return A, 0.5 * (Q + Q.T), c


def closed_form_kf(mdl: o.Model, t, y, dt_final=1e-10):
    """Exact continuous-discrete Kalman filter + RTS smoother for ONE trajectory with a linear drift."""
    F, b = mdl.drift.W.astype(np.float64), mdl.drift.b.astype(np.float64)
    LQL = mdl.L @ mdl.Qc @ mdl.L.T
    T, d = len(t), mdl.d
    fm, fP, pm, pP = np.zeros((T, d)), np.zeros((T, d, d)), np.zeros((T, d)), np.zeros((T, d, d))
    As = np.zeros((T, d, d))
    m, P, ll = mdl.m0.copy(), mdl.P0.copy(), 0.0
    for k in range(T):
        S = mdl.H @ P @ mdl.H.T + mdl.R
        v = y[k] - (mdl.H @ m + mdl.bias)
        ll += -0.5 * v @ np.linalg.solve(S, v) - 0.5 * np.linalg.slogdet(S)[1] - 0.5 * mdl.m * np.log(2 * np.pi)
        K = np.linalg.solve(S, mdl.H @ P).T
        m = m + K @ v
        P = P - K @ S @ K.T
        P = 0.5 * (P + P.T)
        fm[k], fP[k] = m, P
        dt = (t[k + 1] - t[k]) if k + 1 < T else dt_final
        A, Q, c = van_loan(F, LQL, b, dt)
        As[k] = A
        m = A @ m + c
        P = A @ P @ A.T + Q
        pm[k], pP[k] = m, P
    sm, sP = fm.copy(), fP.copy()
    for k in range(T - 2, -1, -1):
        G = np.linalg.solve(pP[k], As[k] @ fP[k]).T
        sm[k] = fm[k] + G @ (sm[k + 1] - pm[k])
        sP[k] = fP[k] + G @ (sP[k + 1] - pP[k]) @ G.T
    return dict(marginal_loglik=ll, filtered_means=fm, filtered_covariances=fP, predicted_means=pm,
                predicted_covariances=pP, smoothed_means=sm, smoothed_covariances=sP)


def model_from_fixture(g) -> o.Model:
    kind = str(g["drift_kind"])
    th = g["theta"]
    d = g["m0"].shape[0]
    if kind == "linear":
        drift = o.LinearDrift(th[: d * d].reshape(d, d), th[d * d:])
    elif kind == "lorenz63":
        drift = o.Lorenz63Drift(*th)
    elif kind == "lorenz96":
        drift = o.Lorenz96Drift(th[0])
    elif kind == "mlp":
        h1, h2 = (int(v) for v in g["hidden"])
        sizes = [(h1, d), (h1,), (h2, h1), (h2,), (d, h2), (d,)]
        parts, off = [], 0
        for shp in sizes:
            n = int(np.prod(shp))
            parts.append(th[off:off + n].reshape(shp))
            off += n
        drift = o.MLPDrift(*parts)
    else:
        raise ValueError(kind)
    return o.Model(drift, g["L"], g["Qc"], g["H"], g["bias"], g["R"], g["m0"], g["P0"])


GOLDEN = ["linear_d2_m6_regular", "tracking_d4_m2_regular", "lorenz63_m3_irregular", "lorenz63_m1_irregular"]
GOLDEN_WIDE = ["lorenz96_d40_m40", "mlp_d8_m4"]
FILTER_KEYS = ["filtered_means", "filtered_covariances", "predicted_means", "predicted_covariances"]


def load_golden(name):
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))


def lorenz96_model(d, m_obs=None, forcing=8.0):
    """f_i = (x_{i+1} - x_{i-2}) x_{i-1} - x_i + F, L = Qc = I, R = I, m0 = F 1, P0 = I;
    m_obs < d observes every (d // m_obs)-th coordinate."""
    m_obs = d if m_obs is None else m_obs
    H = np.eye(d)[:: max(1, d // m_obs)][:m_obs]
    return o.Model(o.Lorenz96Drift(forcing), np.eye(d), np.eye(d), H, np.zeros(m_obs), np.eye(m_obs),
                   forcing * np.ones(d), np.eye(d))


def mlp_model(rng, d=8, m_obs=4, h=64):
    """MLP(d -> h -> h -> d, tanh), weights N(0, 1/fan_in), H = first m rows of I.
    ``h``: one hidden width or a pair (h1, h2)."""
    h1, h2 = (h, h) if np.isscalar(h) else h
    W1 = rng.standard_normal((h1, d)) / np.sqrt(d)
    W2 = rng.standard_normal((h2, h1)) / np.sqrt(h1)
    W3 = rng.standard_normal((d, h2)) / np.sqrt(h2)
    b1, b2, b3 = (0.1 * rng.standard_normal(k) for k in (h1, h2, d))
    return o.Model(o.MLPDrift(W1, b1, W2, b2, W3, b3), np.eye(d), 0.5 * np.eye(d), np.eye(d)[:m_obs], np.zeros(m_obs),
                   0.5 * np.eye(m_obs), np.zeros(d), np.eye(d))


def random_quadratic_drift(rng, d):
    """A random sparse quadratic drift f = c - x + theta_0 B x + theta_1 Q(x), Q_i = sum a_i,jk x_j x_k, as C statements of random shape
    (plain expressions, pow, temporaries) for LearnableCustomDrift and as the oracle's callables (f, Jacobian, grad(div f), and the
    vector-Jacobian products the reverse sweep needs, from the coefficient arrays): returns (f_src, make) with make(theta) -> o.CallableDrift."""
    B = np.where(rng.random((d, d)) < min(1.0, 3.0 / d), rng.standard_normal((d, d)), 0.0) * 0.5
    terms = []   # (i, j, k, a): a x_j x_k in f_i
    for i in range(d):
        for _ in range(int(rng.integers(0, 3))):
            terms.append((i, int(rng.integers(d)), int(rng.integers(d)), 0.3 * rng.standard_normal()))
    c = 0.3 * rng.standard_normal(d)
    A3 = np.zeros((d, d, d))
    for i, j, k, a in terms:
        A3[i, j, k] += a
    lines = []
    style = rng.integers(3)
    for i in range(d):
        lin = " + ".join(f"R({float(B[i, j])!r}) * x[{j}]" for j in range(d) if B[i, j] != 0) or "R(0)"
        qs = []
        for ii, j, k, a in terms:
            if ii != i:
                continue
            if j == k and style == 1:
                qs.append(f"R({float(a)!r}) * pow(x[{j}], 2)")
            else:
                qs.append(f"R({float(a)!r}) * x[{j}] * x[{k}]")
        quad = " + ".join(qs) or "R(0)"
        if style == 2:
            lines.append(f"{{ auto l_ = {lin}; auto q_ = {quad}; fx[{i}] = R({float(c[i])!r}) - x[{i}] + theta[0] * l_ + theta[1] * q_; }}")
        else:
            lines.append(f"fx[{i}] = R({float(c[i])!r}) - x[{i}] + theta[0] * ({lin}) + theta[1] * ({quad});")
    src = "\n".join(lines)
    S3 = A3 + A3.transpose(0, 2, 1)            # d Q_i / d x_j = S3[i, j, :] . x

    def f(x, th):
        return c - x + th[0] * x @ B.T + th[1] * np.einsum("ijk,...j,...k->...i", A3, x, x)

    def jac(x, th):
        return -np.eye(d) + th[0] * B + th[1] * np.einsum("ijk,...k->...ij", S3, x)

    def g(x, th):                              # d/dx_k sum_i dF_ii: th1 sum_i S3[i, i, k]
        return np.broadcast_to(th[1] * np.einsum("iik->k", S3), x.shape).copy()

    def vjp(x, lam, G, th):
        F = jac(x[None], th)[0]
        xb = F.T @ lam + th[1] * np.einsum("ij,ijk->k", G, S3)
        Q = np.einsum("ijk,j,k->i", A3, x, x)
        JQ = np.einsum("ijk,k->ij", S3, x)
        return xb, np.array([lam @ (B @ x) + (G * B).sum(), lam @ Q + (G * JQ).sum()])
    c3 = np.einsum("iik->k", S3)                 # g = theta_1 c3: gradient of u . g is (0, [0, u . c3])
    gvjp = lambda x, u, th: (np.zeros_like(x), np.array([0.0, u @ c3]))
    return src, lambda th: o.CallableDrift(th, f, jac, g, vjp=vjp, gvjp=gvjp)
