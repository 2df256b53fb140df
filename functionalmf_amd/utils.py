"""Host-side helpers of the hot path: trend-filtering penalty and horseshoe
initial scales.  Counterpart of the slice of functionalmf/utils.py the path uses
(utils.py:56-124); same function names and return types."""
import numpy as np
from scipy.sparse import csc_matrix, diags, vstack, issparse


def get_1d_penalty_matrix(N):
    """(N-1) x N first-difference operator, sparse csc (utils.py:93-98)."""
    return diags([-np.ones(N - 1), np.ones(N - 1)], [0, 1], shape=(N - 1, N), format="csc")


def get_delta(D, k):
    """k-th order trend-filtering operator from the oriented incidence matrix
    (utils.py:56-64): alternate left-multiplication by D' and D."""
    if k < 0:
        raise Exception("k must be at least 0th order.")
    out = D
    for i in range(k):
        out = (D.T if i % 2 == 0 else D).dot(out)
    return out


def bayes_delta(D, K, anchor=0):
    """Stack an anchor row e_anchor' on top of the operators of order 0..K
    (utils.py:66-81)."""
    top = np.zeros((1, D.shape[1]))
    top[0, anchor] = 1
    blocks = [csc_matrix(top) if issparse(D) else top]
    blocks += [get_delta(D, k) for k in range(K + 1)]
    if issparse(D):
        return vstack(blocks).tocsc()
    return np.concatenate(blocks, axis=0)


def bayes_grid_penalty(dims, k, anchor=0):
    """Bayesian trend-filtering penalty for a 1-D grid of `dims` points
    (utils.py:83-90).  Multi-dimensional grids are outside the hot path."""
    if hasattr(dims, "__len__"):
        if len(dims) != 1:
            raise NotImplementedError("only 1-D depth grids are on the accelerated path")
        dims = dims[0]
    return bayes_delta(get_1d_penalty_matrix(int(dims)), k, anchor=anchor)


def ilogit(x):
    return 1 / (1 + np.exp(-x))


def mse(x, y):
    return np.nanmean((x - y) ** 2)


def mae(x, y):
    return np.nanmean(np.abs(x - y))


def sample_horseshoe_plus(size=1):
    """Four-level half-Cauchy scale mixture drawn as nested inverse gammas
    (utils.py:115-120); returns (tau2, c, b, a)."""
    a = 1 / np.random.gamma(0.5, 1, size=size)
    b = 1 / np.random.gamma(0.5, a)
    c = 1 / np.random.gamma(0.5, b)
    d = 1 / np.random.gamma(0.5, c)
    return d, c, b, a


def sample_horseshoe(size=1):
    """(utils.py:122-124)"""
    a = 1 / np.random.gamma(0.5, 1, size=size)
    return 1 / np.random.gamma(0.5, a), a
