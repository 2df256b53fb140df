"""Gaussian samplers on the GPU: counterpart of functionalmf/fast_mvn.py - ``sample_mvn_from_precision``
(:10-74), ``sample_mvn_from_covariance`` (:77-142) and the dispatcher ``sample_mvn`` (:145-179).

Sparse (banded) precisions:

    x = Q^-1 mu_part + P' L^-T z,   L L' = P Q P',   z ~ N(0, I)

The reference delegates the factorisation (and the choice of P) to CHOLMOD; here
the ordering is *declared* by the caller (``perm``: new index -> old index, e.g.
the depth-major ordering of the V step) and the factor is a banded Cholesky in
that ordering, one wavefront per system.  The jitter retry schedule is the
reference's (eps, then +10 eps, ... cumulatively, at most ``force_psd_attempts``);
where the reference would warn forever (fast_mvn.py:69-72) this raises.

Dense matrices (``sparse=False``; precision or covariance, plain or pre-factored, ``mu`` or ``mu_part``) go to the
dense kernel (``btf_mvn_dense``): one workgroup per system, right-looking Cholesky, the same jitter schedule.
There is no host computation on either path.
"""
import ctypes as C
import numpy as np

from . import _native


def sample_banded_batch(band, mu_part=None, z=None, seed=0, device=0,
                        force_psd=False, force_psd_eps=1e-6, force_psd_attempts=4):
    """Batched draw for precisions given as lower bands by column:
    band[b, c, a] = Q_b[c+a, c], a = 0..bw.  Returns (x, tries)."""
    lib = _native.load()
    band = _native.as_f64(band)
    B, n, R1 = band.shape
    mu = None if mu_part is None else _native.as_f64(mu_part).reshape(B, n)
    zz = None if z is None else _native.as_f64(z).reshape(B, n)
    x = np.empty((B, n))
    tries = np.zeros(B, dtype=np.int32)
    rc = lib.btf_mvn_banded(device, B, n, R1 - 1, _native.dptr(band), _native.dptr(mu), _native.dptr(zz),
                            C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), float(force_psd_eps),
                            int(force_psd_attempts) if force_psd else 0, _native.dptr(x),
                            tries.ctypes.data_as(_native._c_ip))
    if rc != _native.BTF_OK:
        msg = lib.btf_last_error(None).decode()
        if rc == _native.BTF_ENOTPD:
            raise _native.NotPositiveDefiniteError(rc, msg)
        raise _native.BTFError(rc, msg)
    return x, tries


def dense_to_band(Q, bw):
    n = Q.shape[0]
    band = np.zeros((n, bw + 1))
    for a in range(bw + 1):
        band[:n - a, a] = np.diagonal(Q, -a)
    return band


def sample_mvn_from_precision(Q, mu=None, mu_part=None, sparse=True, chol_factor=False, Q_shape=None,
                              force_psd=False, force_psd_eps=1e-6, force_psd_attempts=4,
                              perm=None, z=None, seed=0, device=0):
    """Draw from N(Q^-1 mu_part, Q^-1) (or N(mu, Q^-1)) for one SPD matrix Q (scipy
    sparse or dense) on the GPU.  ``perm`` (new -> old) is the factor ordering; the
    half-bandwidth of P Q P' must be <= 63.  z, if given, is indexed in the permuted
    order, exactly as the normals enter solve_Lt in fast_mvn.py:41-44; otherwise it
    is drawn from the legacy global numpy generator *after* nothing else (one call
    of size n), as the reference does."""
    if not sparse or chol_factor:
        # fast_mvn.py:49-60 (dense precision; chol_factor: Q is the lower factor of the precision).  A pre-factored
        # *sparse* input is a CHOLMOD factor object in the reference; here a dense lower factor serves both.
        Ld = Q.toarray() if hasattr(Q, "toarray") else np.asarray(Q, dtype=float)
        return sample_dense_batch(Ld[None], precision=True, chol_factor=chol_factor,
                                  mu=None if mu is None or mu_part is not None else np.asarray(mu, float)[None],
                                  mu_part=None if mu_part is None else np.asarray(mu_part, float)[None],
                                  z=(np.random.normal(size=Ld.shape[0]) if z is None else np.asarray(z, float))[None],
                                  device=device, force_psd=force_psd, force_psd_eps=force_psd_eps,
                                  force_psd_attempts=force_psd_attempts)[0][0]
    Qd = Q.toarray() if hasattr(Q, "toarray") else np.asarray(Q, dtype=float)
    n = Qd.shape[0]
    p = np.arange(n) if perm is None else np.asarray(perm)
    Qp = Qd[np.ix_(p, p)]
    r, c = np.nonzero(Qp)
    bw = int(np.max(np.abs(r - c))) if r.size else 0
    if bw > 63:
        raise ValueError("half-bandwidth %d > 63 in the given ordering" % bw)
    band = dense_to_band(Qp, bw)[None]
    if z is None:
        z = np.random.normal(size=n)
    mp = None if mu_part is None else np.asarray(mu_part, float)[p][None]
    x, _ = sample_banded_batch(band, mu_part=mp, z=np.asarray(z, float)[None], seed=seed, device=device,
                               force_psd=force_psd, force_psd_eps=force_psd_eps,
                               force_psd_attempts=force_psd_attempts)
    out = np.empty(n)
    out[p] = x[0]
    if mu_part is None and mu is not None:
        out = out + mu
    return out


def sample_dense_batch(A, precision=False, chol_factor=False, mu=None, mu_part=None, z=None, seed=0, device=0,
                       force_psd=False, force_psd_eps=1e-6, force_psd_attempts=4):
    """Batched dense draws: A[b] an n x n precision (``precision=True``) or covariance matrix, or - with
    ``chol_factor`` - its lower Cholesky factor.  Returns (x, tries).  z: (B, n) standard normals, or None for the
    device generator (Philox keyed by ``seed``)."""
    lib = _native.load()
    A = _native.as_f64(A)
    B, n, n2 = A.shape
    if n != n2:
        raise ValueError("square matrices expected")
    if mu is not None and mu_part is not None:
        raise ValueError("mu and mu_part are mutually exclusive (fast_mvn.py:157)")
    mu = None if mu is None else _native.as_f64(mu).reshape(B, n)
    mp = None if mu_part is None else _native.as_f64(mu_part).reshape(B, n)
    zz = None if z is None else _native.as_f64(z).reshape(B, n)
    x = np.empty((B, n))
    tries = np.zeros(B, dtype=np.int32)
    form = (1 if precision else 0) | (2 if chol_factor else 0)
    rc = lib.btf_mvn_dense(device, B, n, _native.dptr(A), form, _native.dptr(mu), _native.dptr(mp), _native.dptr(zz),
                           C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), float(force_psd_eps),
                           int(force_psd_attempts) if force_psd else 0, _native.dptr(x), tries.ctypes.data_as(_native._c_ip))
    if rc != _native.BTF_OK:
        msg = lib.btf_last_error(None).decode()
        if rc == _native.BTF_ENOTPD:
            raise _native.NotPositiveDefiniteError(rc, msg)
        raise _native.BTFError(rc, msg)
    return x, tries


def sample_mvn_from_covariance(Q, mu=None, mu_part=None, sparse=True, chol_factor=False,
                               force_psd=False, force_psd_eps=1e-6, force_psd_attempts=4, z=None, device=0):
    """Draw from N(mu, Q) or N(Q mu_part, Q) for a covariance matrix Q (fast_mvn.py:77-142): x = L z (+ Q mu_part |
    + mu), L L' = Q; ``chol_factor``: Q is that lower factor.  A scipy-sparse Q is densified (the reference lets
    CHOLMOD permute it: its noise term L z differs by that permutation, the distribution does not).  z: the n normals
    (default: one np.random.normal(size=n) call, as the reference draws them after the factorisation)."""
    Qd = Q.toarray() if hasattr(Q, "toarray") else np.asarray(Q, dtype=float)
    zz = np.random.normal(size=Qd.shape[0]) if z is None else np.asarray(z, float)
    return sample_dense_batch(Qd[None], precision=False, chol_factor=chol_factor,
                              mu=None if mu is None or mu_part is not None else np.asarray(mu, float)[None],
                              mu_part=None if mu_part is None else np.asarray(mu_part, float)[None], z=zz[None],
                              device=device, force_psd=force_psd, force_psd_eps=force_psd_eps,
                              force_psd_attempts=force_psd_attempts)[0][0]


def sample_mvn(Q, mu=None, mu_part=None, sparse=True, precision=False, chol_factor=False, Q_shape=None, **kwargs):
    """The reference's dispatcher (fast_mvn.py:145-179): a scalar or vector Q means Q*I; ``precision`` picks the
    parameterisation; everything else is handed on."""
    if mu is not None and mu_part is not None:
        raise AssertionError("the mean and the mean-part are mutually exclusive")
    if not chol_factor and (np.isscalar(Q) or np.ndim(Q) == 1):
        dim = len(mu) if mu is not None else len(mu_part)
        Q = np.eye(dim) * Q
        sparse = False                     # (a diagonal matrix: the dense kernel; the reference wraps it in csc)
    if precision:
        return sample_mvn_from_precision(Q, mu=mu, mu_part=mu_part, sparse=sparse, chol_factor=chol_factor,
                                         Q_shape=Q_shape, **kwargs)
    return sample_mvn_from_covariance(Q, mu=mu, mu_part=mu_part, sparse=sparse, chol_factor=chol_factor, **kwargs)
