"""Constrainers for ``ParameterProperties.constrainer``: the PSD bijector the reference attaches to its covariance
parameters (/root/reference/dynamax/utils/bijectors.py:5-35, built from TFP's FillTriangular, TransformDiagonal(Exp) and
CholeskyOuterProduct), restated in NumPy with the vector-Jacobian product fit_sgd needs.

    RealToPSDBijector.forward :  x [n(n+1)/2]  ->  L0 = fill_triangular(x)  ->  L = L0 with exp on the diagonal  ->  L L^T
    inverse                   :  P -> cholesky -> log on the diagonal -> fill_triangular_inverse

fill_triangular follows tfp.math.fill_triangular (lower): the vector is laid out as
``concat([x[n:], reverse(x)])`` reshaped to [n, n] and the lower triangle is kept; e.g. [1..6] -> [[4,0,0],[6,5,0],[3,2,1]].
"""
from __future__ import annotations

import numpy as np


def _tri_index(n: int) -> np.ndarray:
    """index[r, c] (c <= r) = position in x of the entry fill_triangular puts at (r, c)."""
    m = n * (n + 1) // 2
    return np.concatenate([np.arange(n, m), np.arange(m - 1, -1, -1)]).reshape(n, n)


def fill_triangular(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x)
    m = x.shape[-1]
    n = int(round(np.sqrt(0.25 + 2 * m) - 0.5))
    if n * (n + 1) // 2 != m:
        raise ValueError(f"fill_triangular: {m} is not a triangular number")
    return np.tril(x[..., _tri_index(n)])


def fill_triangular_inverse(L: np.ndarray) -> np.ndarray:
    L = np.asarray(L)
    n = L.shape[-1]
    idx = _tri_index(n)
    out = np.zeros(L.shape[:-2] + (n * (n + 1) // 2,), L.dtype)
    r, c = np.tril_indices(n)
    out[..., idx[r, c]] = L[..., r, c]
    return out


class RealToPSDBijector:
    """dynamax.utils.bijectors.RealToPSDBijector (bijectors.py:21-35)."""

    def forward(self, x):
        L = fill_triangular(np.asarray(x, np.float64))
        n = L.shape[-1]
        di = np.arange(n)
        L[..., di, di] = np.exp(L[..., di, di])
        return L @ np.swapaxes(L, -1, -2)

    def inverse(self, P):
        L = np.linalg.cholesky(np.asarray(P, np.float64))
        n = L.shape[-1]
        di = np.arange(n)
        L[..., di, di] = np.log(L[..., di, di])
        return fill_triangular_inverse(L)

    def forward_vjp(self, x, g_P):
        """Cotangent of x given the cotangent g_P of forward(x) (any g_P; only its symmetric part matters)."""
        L = fill_triangular(np.asarray(x, np.float64))
        n = L.shape[-1]
        di = np.arange(n)
        L[..., di, di] = np.exp(L[..., di, di])
        g = np.asarray(g_P, np.float64)
        gL = np.tril((g + np.swapaxes(g, -1, -2)) @ L)
        gL[..., di, di] *= L[..., di, di]
        return fill_triangular_inverse(gL)

    def forward_log_det_jacobian_and_grad(self, x):
        """log |det d vech(forward(x)) / dx| and its gradient: TFP's FillScaleTriL(Exp) contributes sum_j x_jj (the
        diagonal entries of fill_triangular(x), i.e. log L_jj) and CholeskyOuterProduct n log 2 + sum_j (n - j) log L_jj,
        j = 0..n-1 -- what ``prop.constrainer.forward_log_det_jacobian(unc_value)`` returns in dynamax/parameters.py:118-121."""
        x = np.asarray(x, np.float64)
        m = x.shape[-1]
        n = int(round(np.sqrt(0.25 + 2 * m) - 0.5))
        idx = _tri_index(n)[np.arange(n), np.arange(n)]
        w = n - np.arange(n) + 1.0
        grad = np.zeros(m)
        grad[idx] = w
        return n * np.log(2.0) + float(np.dot(w, x[idx])), grad
