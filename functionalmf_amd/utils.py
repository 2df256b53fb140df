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


def posterior_summary(Ws, Vs, q=(5, 95), transform=None, device=0):
    """Mean and percentiles over the kept samples of f(w_s[i] . v_s[j,t]) for every cell, on the GPU.

    The reference's example scripts do this on the host
    (examples/gaussian_tensor_filtering.py:82-85):

        Mu_hat = np.einsum('znk,zmtk->znmt', Ws, Vs)        # (S, N, M, T): 67 GB at S=1000, (512,256,64)
        mean, lo, hi = Mu_hat.mean(0), np.percentile(Mu_hat, 5, axis=0), np.percentile(Mu_hat, 95, axis=0)

    Here the (S, N, M, T) tensor is never materialised (btf_posterior_summary: per-cell bitonic sort in
    LDS).  `transform`: None, "ilogit" (the Binomial examples) or "square".  Returns (mean, quantiles)
    with shapes (N, M, T) and (len(q), N, M, T); percentiles use numpy's default linear interpolation.
    There is no CPU fallback."""
    from . import _native
    Ws, Vs = _native.as_f64(Ws), _native.as_f64(Vs)
    if Ws.ndim != 3 or Vs.ndim != 4 or Ws.shape[0] != Vs.shape[0] or Ws.shape[2] != Vs.shape[3]:
        raise ValueError("Ws must be (S, N, K) and Vs (S, M, T, K)")
    code = {None: 0, "identity": 0, "ilogit": 1, "square": 2}[transform]
    S, N, K = Ws.shape
    M, T = Vs.shape[1:3]
    qs = _native.as_f64(np.atleast_1d(q))
    mean = np.zeros((N, M, T))
    quant = np.zeros((len(qs), N, M, T))
    lib = _native.load()
    rc = lib.btf_posterior_summary(int(device), S, N, M, T, K, _native.dptr(Ws), _native.dptr(Vs), code,
                                   _native.dptr(qs), len(qs), _native.dptr(mean), _native.dptr(quant))
    if rc != _native.BTF_OK:
        raise _native.BTFError(rc, lib.btf_last_error(None).decode())
    return mean, quant
