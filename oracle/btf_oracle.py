"""CPU oracle for the Bayesian Tensor Filtering Gibbs hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``functionalmf_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / the CPU number reported
beside the GPU one.

This is a numpy/scipy *restatement* (not a copy) of the algorithm of
tansey/functionalmf for the path ``run_gibbs -> resample -> _resample_W /
_resample_V / _resample_nu2`` (+ the hyper-parameter steps and the
``fast_mvn`` precision sampler those call).  Every function cites the
reference file:line whose arithmetic it follows.

Parity status (see DESIGN.md "Oracle"):
  * pinned by the real reference code run in the build container (fixtures in
    tests/golden/, generator tests/golden/make_golden.py): trend-filter
    penalty, W step (draw included), V-step precision/mean assembly, the mean
    term Q^-1 mu, the RNG stream order, nu2/sigma2/Tau2/lam2 steps, run_gibbs
    result layout, quirks Q1-Q4 of SURVEY.md section 8.
  * NOT pinned ("parity unpinned"): the V-step *noise* term under CHOLMOD's own
    fill-reducing ordering (scikit-sparse is absent from the image, version
    unpinned in reference setup.py:51) and the Polya-Gamma draws (pypolyagamma
    absent).  The build declares the factor ordering instead (``perm``
    argument below: depth-major) and validates PG draws distributionally.

The state is a plain dict so the oracle shares no class structure with the
product code:  W (N,K)  V (M,T,K)  Tau2 (M,nD)  Tau2_a/b/c (M,nD)  lam2  lam2_a
sigma2  nu2 (scalar or (N,M,T)).
All random numbers come from the *global legacy* numpy generator, in the
reference's order, unless a ``z`` array is injected.
"""
import numpy as np
import scipy.linalg as sla

# --------------------------------------------------------------------------
# trend-filtering penalty            (reference functionalmf/utils.py:56-98)
# --------------------------------------------------------------------------

def first_difference(T):
    """(T-1) x T first-difference operator, rows (-1, +1).  utils.py:93-98."""
    D = np.zeros((T - 1, T))
    idx = np.arange(T - 1)
    D[idx, idx] = -1.0
    D[idx, idx + 1] = 1.0
    return D


def difference_power(D, k):
    """k-th order operator built by alternating D' and D.  utils.py:56-64."""
    if k < 0:
        raise ValueError("order must be >= 0")
    out = D
    for step in range(k):
        out = D.T @ out if step % 2 == 0 else D @ out
    return out


def trend_penalty(T, order, anchor=0):
    """Dense Delta = [e_anchor'; D^(0); ...; D^(order)].  utils.py:66-90."""
    D = first_difference(T)
    top = np.zeros((1, T))
    top[0, anchor] = 1.0
    blocks = [top] + [difference_power(D, k) for k in range(order + 1)]
    return np.concatenate(blocks, axis=0)


def depth_major_perm(K, T):
    """perm[t*K + k] = k*T + t : new (depth-major) index -> reference (k-major)
    index.  This is the factor ordering the build declares."""
    t, k = np.meshgrid(np.arange(T), np.arange(K), indexing="ij")
    return (k * T + t).reshape(-1)


def twisted_order(K, T, tf):
    """Elimination order of the build's default ("twisted") V-step kernel as depth-major
    indices g = t*K + k: depths 0..ts-1 ascending, depths T-1..ts+S descending (k
    descending), then the separator depths ts..ts+S-1; S = tf+1, ts = (T-S)//2."""
    S = tf + 1
    ts = (T - S) // 2
    n, nl = K * T, ts * K
    left = np.arange(nl)
    right = np.arange(n - 1, nl + S * K - 1, -1)
    sep = np.arange(nl, nl + S * K)
    return np.concatenate([left, right, sep])


def perm_from_order(order, K, T):
    """Depth-major pivot order -> the reference's k-major unknown indices (CHOLMOD's P())."""
    order = np.asarray(order)
    return (order % K) * T + order // K


def twisted_perm(K, T, tf):
    return perm_from_order(twisted_order(K, T, tf), K, T)


# --------------------------------------------------------------------------
# sufficient statistics              (factor.py:323-330 and :368-375)
# --------------------------------------------------------------------------

def replicate_stats(Y):
    """Observed-replicate count and NaN-mean over the last axis."""
    if Y.ndim == 3:
        Y = Y[..., None]
    obs = ~np.isnan(Y)
    cnt = obs.sum(axis=-1)
    tot = np.where(obs, Y, 0.0).sum(axis=-1)
    with np.errstate(invalid="ignore", divide="ignore"):
        ybar = tot / cnt
    return cnt, ybar


# --------------------------------------------------------------------------
# precision-form Gaussian draw       (fast_mvn.py:10-74)
# --------------------------------------------------------------------------

class NotPositiveDefinite(Exception):
    pass


def mvn_from_precision(Q, mu_part=None, perm=None, z=None,
                       force_psd=True, eps0=1e-6, attempts=4, info=None):
    """x = Q^-1 mu_part + P' L^-T z  with  L L' = P Q P'.

    fast_mvn.py:35-47 (sparse branch).  ``perm`` plays the role of CHOLMOD's
    P(); None = identity.  z is drawn (global legacy RNG) only after a
    factorisation succeeded (fast_mvn.py:38 then :41), so failed attempts use
    no random numbers.  On failure eps0*10^a is added to the diagonal
    cumulatively, at most ``attempts`` times (fast_mvn.py:62-68); where the
    reference would then warn forever (:69-72) this raises instead.
    """
    Q = np.array(Q, dtype=float)
    n = Q.shape[0]
    p = np.arange(n) if perm is None else np.asarray(perm)
    eps = eps0
    tried = 0
    while True:
        Qp = Q[np.ix_(p, p)]
        try:
            L = np.linalg.cholesky(Qp)
            break
        except np.linalg.LinAlgError:
            if force_psd and tried < attempts:
                Q[np.diag_indices(n)] += eps
                eps *= 10.0
                tried += 1
            else:
                raise NotPositiveDefinite("precision not PD after %d shifts" % tried)
    if info is not None:
        info["attempts"] = tried
    if z is None:
        z = np.random.normal(size=n)
    x = np.empty(n)
    x[p] = sla.solve_triangular(L.T, z, lower=False)      # P' L^-T z
    if mu_part is not None:
        y = sla.cho_solve((L, True), np.asarray(mu_part)[p])
        mean = np.empty(n)
        mean[p] = y
        x = x + mean
    return x


def gram_eigensystem(G):
    """Eigenvalues ascending, eigenvectors as columns, each vector's largest-magnitude entry
    positive: the convention the build declares for its spectral V sampler (include/btf.h)."""
    g, U = np.linalg.eigh(np.asarray(G, float))
    big = np.argmax(np.abs(U), axis=0)
    U = U * np.where(U[big, np.arange(U.shape[1])] < 0, -1.0, 1.0)[None, :]
    return g, U


def spectral_pivot_order(T, S):
    """Depth of pivot i inside one system of the spectral sampler ("burn at both ends", include/btf.h):
    depths 0..ts-1 ascending, depths T-1..ts+S descending, then the separator ts..ts+S-1, ts = (T-S)//2;
    natural order when T < 2S+2."""
    if T < 2 * S + 2:
        return np.arange(T)
    ts = (T - S) // 2
    return np.concatenate([np.arange(ts), np.arange(T - 1, ts + S - 1, -1), np.arange(ts, ts + S)])


def band_halfwidth(P):
    """Half-bandwidth of a symmetric banded matrix (tf_order + 1 for the trend-filter prior)."""
    r, c = np.nonzero(np.abs(P) > 0)
    return int(np.max(np.abs(r - c))) if r.size else 0


def kronecker_sum_parts(Q, K, T):
    """Split a k-major precision of the form  G (x) I_T + I_K (x) P  (factor.py:396-405 with constant
    likelihood weights) into a K x K matrix with G's eigenvectors and the T x T remainder:
    Gs = G + P[0,0] I (entry (k T, k' T) of Q), Ps = P - P[0,0] I; raises if Q is not of that form."""
    Q = np.asarray(Q, float)
    idx = np.arange(K) * T
    Gs = Q[np.ix_(idx, idx)].copy()
    Ps = Q[:T, :T] - Gs[0, 0] * np.eye(T)
    if np.abs(np.kron(Gs, np.eye(T)) + np.kron(np.eye(K), Ps) - Q).max() > 1e-9 * np.abs(Q).max():
        raise ValueError("precision is not a Kronecker sum: the spectral sampler does not apply")
    return Gs, Ps


def mvn_from_precision_spectral(Q, K, T, mu_part=None, z=None, force_psd=True, eps0=1e-6, attempts=4, info=None, S=None):
    """The square root the build's "spectral" V sampler declares in place of CHOLMOD's P' L^-T
    (fast_mvn.py:35-47):  x = Q^-1 mu_part + (U (x) I_T) blockdiag_k(C_k^-T) z,  C_k C_k' = g_k I + P in
    the pivot order o = spectral_pivot_order (C_k C_k' = (g_k I + P)[o, o]), Q = G (x) I_T + I_K (x) P
    (k-major), G = U diag(g) U'.  z[k*T + i] multiplies pivot i of system k.  Jitter as fast_mvn.py:62-68: eps
    added to the diagonal of Q cumulatively; z is drawn only after the factorisation succeeded."""
    Gs, Ps = kronecker_sum_parts(Q, K, T)
    g, U = gram_eigensystem(Gs)
    o = spectral_pivot_order(T, band_halfwidth(Q[:T, :T]) if S is None else S)
    Po = Ps[np.ix_(o, o)]
    shift, eps, tried = 0.0, eps0, 0
    while True:
        try:
            C = [np.linalg.cholesky(Po + (g[k] + shift) * np.eye(T)) for k in range(K)]
            break
        except np.linalg.LinAlgError:
            if force_psd and tried < attempts:
                shift += eps
                eps *= 10.0
                tried += 1
            else:
                raise NotPositiveDefinite("precision not PD after %d shifts" % tried)
    if info is not None:
        info["attempts"] = tried
    if z is None:
        z = np.random.normal(size=K * T)
    zt = np.asarray(z, float).reshape(K, T)
    mt = np.zeros((K, T)) if mu_part is None else U.T @ np.asarray(mu_part, float).reshape(K, T)
    xt = np.empty((K, T))
    for k in range(K):
        xt[k, o] = sla.cho_solve((C[k], True), mt[k, o]) + sla.solve_triangular(C[k].T, zt[k], lower=False)
    return (U @ xt).reshape(-1)


# --------------------------------------------------------------------------
# W half-sweep                       (factor.py:313-362)
# --------------------------------------------------------------------------

def w_step(st, Y, z=None, row0=0):
    """Row-by-row conjugate draw of W.  (row0: global index of the first row when
    st["W"] / Y hold only a block of rows - used by the sharding tests.)  Reproduces quirk Q1: the design/factor
    cache is refreshed only for rows < K or when the data holds any NaN
    (factor.py:320, :349).  ``z``: optional flat array of the sum_i min(i+1,K)
    normals, consumed in row order; else the legacy global RNG is used.
    """
    W, V = st["W"], st["V"]
    N, K = W.shape
    any_nan = bool(np.isnan(Y).any())
    cnt, ybar = replicate_stats(Y)
    Vflat = V.reshape(-1, K)
    nu2 = st["nu2"]
    zpos = 0
    Xt = Lt = None
    if z is not None and row0:
        zpos = row0 * (row0 + 1) // 2 if row0 < K else K * (K + 1) // 2 + (row0 - K) * K
    for i in range(N):
        d = min(row0 + i + 1, K)
        yb = ybar[i].reshape(-1)
        keep = ~np.isnan(yb)
        if np.isscalar(nu2) or np.ndim(nu2) == 0:
            c = cnt[i].reshape(-1)[keep] / nu2
        else:
            c = cnt[i].reshape(-1)[keep] / nu2[i].reshape(-1)[keep]
        if row0 + i < K or any_nan or Xt is None:
            Vd = Vflat[keep][:, :d]
            Xt = (Vd * c[:, None]).T
            Q = Xt @ Vd + np.eye(d) / st["sigma2"]
            Lt = np.linalg.cholesky(Q).T
        m = Xt @ yb[keep]
        if z is None:
            zi = np.random.normal(size=d)
        else:
            zi = z[zpos:zpos + d]
            zpos += d
        W[i, :d] = sla.cho_solve((Lt, False), m) + sla.solve_triangular(Lt, zi, lower=False)
    return W


# --------------------------------------------------------------------------
# V half-sweep                       (factor.py:364-409)
# --------------------------------------------------------------------------

def prior_precision_1d(Delta, lam2, tau2_row):
    """Delta' diag(1/(lam2*tau2)) Delta  (T x T).  factor.py:404-405."""
    return Delta.T @ (Delta / (lam2 * tau2_row)[:, None])


def v_step_system(st, Y, Delta, j, src):
    """Precision (k-major, dense) and mean-part of column j with the likelihood
    weights taken from column ``src`` (Q2: the reference re-uses the cached
    design of the last column whose NaN pattern differed, factor.py:394-401)."""
    W = st["W"]
    N, K = W.shape
    T = Delta.shape[1]
    cnt, ybar = st["_cnt"], st["_ybar"]
    nu2 = st["nu2"]
    yb = ybar[:, j, :]                      # (N,T)
    keep = ~np.isnan(yb)
    if np.isscalar(nu2) or np.ndim(nu2) == 0:
        c = cnt[:, src, :] / nu2
    else:
        with np.errstate(divide="ignore", invalid="ignore"):
            c = cnt[:, src, :] / nu2[:, src, :]
    c = np.where(keep, c, 0.0)
    y0 = np.where(keep, yb, 0.0)
    # Q_lik[(k,t),(k',t)] = sum_i c_it W_ik W_ik'      (kron(W,I_T)' C kron(W,I_T))
    G = np.einsum("it,ik,il->tkl", c, W, W)             # (T,K,K)
    mu = np.einsum("it,ik->kt", c * y0, W).reshape(-1)   # k-major
    Q = np.zeros((K * T, K * T))
    tt = np.arange(T)
    for k in range(K):
        for l in range(K):
            Q[k * T + tt, l * T + tt] = G[:, k, l]
    P1 = prior_precision_1d(Delta, st["lam2"], st["Tau2"][j])
    for k in range(K):
        Q[k * T:(k + 1) * T, k * T:(k + 1) * T] += P1
    return Q, mu


def stale_column_sources(ybar):
    """src[j] = column whose cached weights column j uses (quirk Q2)."""
    M = ybar.shape[1]
    src = np.zeros(M, dtype=np.int64)
    pat = None
    for j in range(M):
        miss = np.isnan(ybar[:, j, :]).reshape(-1)
        if j == 0 or np.any(pat != miss):
            pat = miss
            src[j] = j
        else:
            src[j] = src[j - 1]
    return src


def v_step(st, Y, Delta, perm="depth", z=None, compat="reference",
           force_psd=True, eps0=1e-6, attempts=4, info=None, cols=None):
    """Column-by-column conjugate draw of V (factor.py:364-409 + fast_mvn).
    perm: "depth" (declared ordering), "identity", "twist", an explicit array, or "spectral"
    (mvn_from_precision_spectral: complete-data columns only).
    z: optional (M, K*T) normals, row j used for column j, indexed in the
    *permuted* order (as solve_Lt sees them, fast_mvn.py:44)."""
    V = st["V"]
    M, T, K = V.shape
    st["_cnt"], st["_ybar"] = replicate_stats(Y)
    spectral = isinstance(perm, str) and perm == "spectral"
    if isinstance(perm, str):
        if perm == "twist":
            p = twisted_perm(K, T, (Delta.shape[0] + 1) // T - 1 if Delta.shape[0] != T else 0)
        else:
            p = depth_major_perm(K, T) if perm == "depth" else np.arange(K * T)
    else:
        p = np.asarray(perm)
    src = stale_column_sources(st["_ybar"]) if compat == "reference" else np.arange(M)
    tries = np.zeros(M, dtype=np.int64)
    for j in (range(M) if cols is None else cols):
        Q, mu = v_step_system(st, Y, Delta, j, int(src[j]))
        inf = {}
        if spectral:
            x = mvn_from_precision_spectral(Q, K, T, mu_part=mu, z=None if z is None else z[j], force_psd=force_psd,
                                            eps0=eps0, attempts=attempts, info=inf, S=band_halfwidth(Delta.T @ Delta))
        else:
            x = mvn_from_precision(Q, mu_part=mu, perm=p,
                                   z=None if z is None else z[j],
                                   force_psd=force_psd, eps0=eps0, attempts=attempts, info=inf)
        tries[j] = inf.get("attempts", 0)
        V[j] = x.reshape(K, T).T
    if info is not None:
        info["attempts"] = tries
        info["src"] = src
    st.pop("_cnt"), st.pop("_ybar")
    return V


# --------------------------------------------------------------------------
# variance / shrinkage steps   (genlasso.py:149-168, factor.py:130-153,411-416)
# --------------------------------------------------------------------------

def inv_gamma_precision_draw(means, obs, shape, rate):
    """One Gamma(a + n/2, scale=1/(b + sse/2)) precision draw.  genlasso.py:149-168."""
    miss = np.isnan(obs)
    sse = np.nansum((means - obs) ** 2)
    return np.random.gamma(shape + np.sum(~miss) / 2.0, 1.0 / (rate + sse / 2.0))


def sse_and_count(st, Y):
    """Residual sum of squares and number of observed values (the two numbers
    the nu2 update needs).  factor.py:411-416."""
    Mu = np.einsum("nk,mtk->nmt", st["W"], st["V"])
    if Y.ndim == 4:
        Mu = Mu[..., None]
    r = Mu - Y
    return float(np.nansum(r * r)), int(np.sum(~np.isnan(Y)))


def nu2_step(st, Y, a=0.1, b=0.1):
    sse, n = sse_and_count(st, Y)
    st["nu2"] = 1.0 / np.random.gamma(a + n / 2.0, 1.0 / (b + sse / 2.0))
    return st["nu2"]


def free_w_entries(W):
    """Lower-triangular head + dense tail of W as one vector.  factor.py:155-174."""
    N, K = W.shape
    h = min(N, K)
    return np.concatenate([W[np.tril_indices(h)], W[h:].reshape(-1)])


def sigma2_step(st, a=0.1, b=0.1):
    w = free_w_entries(st["W"])
    st["sigma2"] = 1.0 / inv_gamma_precision_draw(np.zeros_like(w), w, a, b)
    return st["sigma2"]


def tau2_step(st, Delta, stability=1e-6):
    """Horseshoe+ chain per column, 4 vector gamma draws each.  factor.py:134-141."""
    V = st["V"]
    M, T, K = V.shape
    lo, hi = stability, 1.0 / stability
    for j in range(M):
        d = Delta @ V[j]
        rate = (d * d).sum(axis=1) / (2.0 * st["lam2"]) + 1.0 / np.clip(st["Tau2_c"][j], lo, hi)
        st["Tau2"][j] = 1.0 / np.random.gamma((K + 1) / 2.0, 1.0 / np.clip(rate, lo, hi))
        st["Tau2_c"][j] = 1.0 / np.random.gamma(1.0, 1.0 / np.clip(1.0 / st["Tau2"][j] + 1.0 / st["Tau2_b"][j], lo, hi))
        st["Tau2_b"][j] = 1.0 / np.random.gamma(1.0, 1.0 / np.clip(1.0 / st["Tau2_c"][j] + 1.0 / st["Tau2_a"][j], lo, hi))
        st["Tau2_a"][j] = 1.0 / np.random.gamma(1.0, 1.0 / np.clip(1.0 / st["Tau2_b"][j] + 1.0, lo, hi))


def lam2_step(st, Delta, compat="reference"):
    """Global shrinkage.  compat="reference" keeps quirk Q3 (factor.py:147-150:
    the rate is overwritten each column, so only the last column counts);
    "exact" accumulates 1/lam2_a + sum_j."""
    V = st["V"]
    M, T, K = V.shape
    rate = 1.0 / st["lam2_a"]
    for j in range(M):
        d = Delta @ V[j]
        term = ((d / np.sqrt(st["Tau2"][j])[:, None]) ** 2).sum() / 2.0
        rate = term if compat == "reference" else rate + term
    shape = Delta.shape[0] * M * K + 1
    st["lam2"] = max(1e-5, 1.0 / np.random.gamma(shape / 2.0, 1.0 / rate))
    st["lam2_a"] = 1.0 / np.random.gamma(1.0, 1.0 / (1.0 / st["lam2"] + 1.0))


# --------------------------------------------------------------------------
# construction draws                 (factor.py:53-110, 230-253; utils.py:115-124)
# --------------------------------------------------------------------------

def horseshoe_plus(size):
    a = 1.0 / np.random.gamma(0.5, 1.0, size=size)
    b = 1.0 / np.random.gamma(0.5, a)
    c = 1.0 / np.random.gamma(0.5, b)
    d = 1.0 / np.random.gamma(0.5, c)
    return d, c, b, a


def init_state(N, M, T, K=5, tf_order=2, sigma2_init=None, lam2_init=None,
               nu2_init=None, sigma2_a=0.1, sigma2_b=0.1, nu2_a=0.1, nu2_b=0.1,
               perm="depth", force_psd=True, eps0=1e-6, attempts=4):
    """Initial state in the reference's construction order (factor.py:49-110,
    then :292-304): sigma2, lam2(+a), Tau2(+a,b,c), W, V, then nu2."""
    Delta = trend_penalty(T, tf_order)
    st = {}
    st["sigma2"] = sigma2_init if sigma2_init is not None else \
        1.0 / np.random.gamma(sigma2_a, 1.0 / sigma2_b, size=1)
    a = 1.0 / np.random.gamma(0.5, 1.0, size=1)
    lam2 = 1.0 / np.random.gamma(0.5, a)
    st["lam2"], st["lam2_a"] = np.clip(lam2, 0, 4), a
    if lam2_init is not None:
        st["lam2"] = lam2_init
    t2, c, b, a = horseshoe_plus((M, Delta.shape[0]))
    st["Tau2"], st["Tau2_c"], st["Tau2_b"], st["Tau2_a"] = np.clip(t2, 0, 9), c, b, a
    W = np.random.normal(0, np.sqrt(st["sigma2"]), size=(N, K))
    if N > 1:
        W[np.triu_indices(K, k=1)] = 0
    st["W"] = W
    V = np.full((M, T, K), np.nan)
    p = depth_major_perm(K, T) if perm == "depth" else np.arange(K * T)
    for j in range(M):
        P1 = prior_precision_1d(Delta, st["lam2"], st["Tau2"][j])
        Q = np.kron(np.eye(K), P1)
        V[j] = mvn_from_precision(Q, perm=p, force_psd=force_psd, eps0=eps0,
                                  attempts=attempts).reshape(K, T).T
    st["V"] = np.clip(V, -10, 10)
    st["nu2"] = nu2_init if nu2_init is not None else \
        1.0 / np.random.gamma(nu2_a, 1.0 / nu2_b, size=1)
    return st, Delta


# --------------------------------------------------------------------------
# sweep + driver                     (factor.py:306-311, :112-128; genlasso.py:37-66)
# --------------------------------------------------------------------------

def gaussian_sweep(st, Y, Delta, perm="depth", compat="reference", flags=None):
    f = dict(nu2=True, sigma2=True, Tau2=True, lam2=True, W=True, V=True)
    if flags:
        f.update(flags)
    if f["nu2"]:
        nu2_step(st, Y)
    if f["sigma2"]:
        sigma2_step(st)
    if f["Tau2"]:
        tau2_step(st, Delta)
    if f["lam2"]:
        lam2_step(st, Delta, compat=compat)
    if f["W"]:
        w_step(st, Y)
    if f["V"]:
        v_step(st, Y, Delta, perm=perm, compat=compat)


def run_gibbs(st, Y, Delta, nburn, nthin, nsamples, perm="depth", compat="reference", flags=None):
    """genlasso.py:37-66: result arrays [nsamples]+shape, scalars as [nsamples,1]."""
    keys = ("W", "V", "sigma2", "lam2", "Tau2", "nu2")
    out = None
    for step in range(nburn + nthin * nsamples):
        gaussian_sweep(st, Y, Delta, perm=perm, compat=compat, flags=flags)
        if step >= nburn and (step - nburn) % nthin == 0:
            s = (step - nburn) // nthin
            if s == 0:
                out = {k: np.zeros([nsamples] + ([1] if np.isscalar(st[k]) else list(np.shape(st[k]))))
                       for k in keys}
            for k in keys:
                out[k][s] = st[k]
    return out


# --------------------------------------------------------------------------
# Binomial / Polya-Gamma path        (factor.py:425-460)
# --------------------------------------------------------------------------

def binomial_kappa(Ysucc, Ntrials, nu2):
    """kappa = (Y - N/2) * nu2 : the pseudo-observations handed to the Gaussian
    steps (factor.py:437-445)."""
    return (Ysucc - Ntrials / 2.0) * nu2


def binomial_w_step(st, Ysucc, Ntrials, z=None):
    return w_step(st, binomial_kappa(Ysucc, Ntrials, st["nu2"]), z=z)


def binomial_v_step(st, Ysucc, Ntrials, Delta, **kw):
    return v_step(st, binomial_kappa(Ysucc, Ntrials, st["nu2"]), Delta, **kw)


def pg_mean(b, c):
    """E[PG(b,c)] = b/(2c) tanh(c/2)  (b/4 at c=0)."""
    b = np.asarray(b, float)
    c = np.abs(np.asarray(c, float))
    with np.errstate(invalid="ignore", divide="ignore"):
        m = np.where(c > 1e-8, b / (2 * c) * np.tanh(c / 2), b / 4.0 * (1 - c * c / 12.0))
    return m


def pg_var(b, c):
    """Var[PG(b,c)] = b/(4c^3) (sinh c - c) sech^2(c/2)  (b/24 at c=0)."""
    b = np.asarray(b, float)
    c = np.abs(np.asarray(c, float))
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        v = np.where(c > 1e-3,
                     b / (4 * c ** 3) * (np.sinh(c) - c) / np.cosh(c / 2) ** 2,
                     b / 24.0 * (1 - c * c / 5.0))
    return v


def pg_draw_series(b, c, size, rng, nterms=256):
    """Reference sampler straight from the definition (Polson, Scott & Windle
    2013, eq. 2):  PG(b,c) = 1/(2 pi^2) sum_k g_k / ((k-1/2)^2 + c^2/(4 pi^2)),
    g_k ~ Gamma(b,1).  Truncated at ``nterms`` with the tail replaced by its
    mean.  Slow; used only to validate the device sampler's distribution."""
    k = np.arange(1, nterms + 1)
    den = (k - 0.5) ** 2 + (c * c) / (4 * np.pi ** 2)
    g = rng.gamma(b, 1.0, size=(size, nterms))
    x = (g / den).sum(axis=1)
    kk = np.arange(nterms + 1, nterms + 200001)
    tail = (b / ((kk - 0.5) ** 2 + (c * c) / (4 * np.pi ** 2))).sum()
    return (x + tail) / (2 * np.pi ** 2)


def pg_draw_series_cells(b, c, rng, nterms=200):
    """The same definition-based sampler for MANY cells at once (b: common shape, c: one tilt per cell), the truncated
    tail replaced by its mean - the CPU stand-in for `pgdrawv` (factor.py:459) that bench.py's Binomial cpu_baseline
    times on a sample of the cells (pypolyagamma itself is absent: DESIGN.md section 2)."""
    c = np.abs(np.asarray(c, float)).reshape(-1)
    k = np.arange(1, nterms + 1)
    c2 = (c * c / (4 * np.pi ** 2))[:, None]
    den = (k - 0.5) ** 2 + c2
    x = (rng.gamma(b, 1.0, size=(c.size, nterms)) / den).sum(axis=1)
    # tail mean sum_{k > nterms} b / ((k - 1/2)^2 + c2) ~ b * integral: (1/sqrt(c2)) (pi/2 - atan(nterms / sqrt(c2))), 1/nterms at c2 = 0
    r = np.sqrt(c2[:, 0])
    with np.errstate(divide="ignore", invalid="ignore"):
        tail = np.where(r > 1e-12, (np.pi / 2 - np.arctan(nterms / r)) / r, 1.0 / nterms) * b
    return (x + tail) / (2 * np.pi ** 2)


# --------------------------------------------------------------------------
# Negative-Binomial rate update      (factor.py:462-563)
#   counts y ~ NB(R, p), p = ilogit(w.v): given R the augmented model is the
#   Binomial one with Y = sum_r y_r successes out of N = sum_r (y_r + R) trials.
#   R is updated by `nmetropolis` random-walk MH steps on log R; rdims are the
#   dims of (rows, cols, depth) a single R value is shared across.
# --------------------------------------------------------------------------

def nb_rate_shape(dims, rdims):
    """Shape of R: 1 along shared dims (factor.py:560-561)."""
    return tuple(1 if i in rdims else c for i, c in enumerate(dims))


def nb_init_rate(dims, rdims, rstdev=1.0):
    """exp(N(0, rstdev)) + 1  (factor.py:560-563)."""
    return np.exp(np.random.normal(0, rstdev, size=nb_rate_shape(dims, rdims))) + 1


def nb_log1m_p(W, V):
    """log(1 - ilogit(clip(w.v, -10, 10)))  (factor.py:519, :534)."""
    x = np.einsum("nk,mtk->nmt", W, V).clip(-10, 10)
    return -np.log1p(np.exp(x))


def nb_loglik_ratio(data, R, candR, l1p, rdims):
    """Log-likelihood ratio candidate/current, summed over replicates and the shared dims;
    NaN observations drop out with their whole term (factor.py:533-538).  data (N,M,T,r);
    R, candR broadcastable (.,.,.); returns the array over the unshared dims."""
    from scipy.special import gammaln
    R4, C4 = R[..., None], candR[..., None]
    term = (gammaln(data + C4) - gammaln(C4) - gammaln(data + R4) + gammaln(R4)
            + (C4 - R4) * l1p[..., None])
    for dim in [3] + sorted(rdims)[::-1]:
        term = np.nansum(term, axis=dim)
    return term


def nb_resample_rate(st, data, rdims=(0, 1, 2), nmetropolis=30, rpropstdev=0.1, rstdev=1.0):
    """factor.py:513-554: st["R"] updated in place (global legacy RNG: per step one normal and
    one uniform array of R's shape); returns the Binomial trial counts N = nansum(data + R)."""
    from scipy.stats import norm
    if data.ndim == 3:
        data = data[..., None]
    R = st["R"]
    logR = np.log(R)
    l1p = nb_log1m_p(st["W"], st["V"])
    for _ in range(nmetropolis):
        cand_log = logR + np.random.normal(0, rpropstdev, size=logR.shape)
        cand = np.exp(cand_log)
        a_prior = norm.logpdf(cand_log, loc=0, scale=rstdev) - norm.logpdf(logR, loc=0, scale=rstdev)
        a_lik = nb_loglik_ratio(data, R, cand, l1p, rdims)
        a_prior = np.squeeze(a_prior).reshape(a_lik.shape)
        prob = np.exp(np.clip(a_prior + a_lik, -10, 1)).reshape(R.shape)
        acc = np.random.random(size=prob.shape) <= prob
        acc = acc & (cand > 1)
        logR[acc] = cand_log[acc]
        R[acc] = np.exp(cand_log[acc])
    return nb_trials(data, R)


def nb_trials(data, R):
    """Binomial pseudo-data of the augmented model (factor.py:503-505, :552):
    Y = nansum(data) with NaN where every replicate is missing; N = nansum(data + R)
    (0, not NaN, where every replicate is missing)."""
    if data.ndim == 3:
        data = data[..., None]
    missing = np.all(np.isnan(data), axis=-1)
    Y = np.nansum(data, axis=-1)
    Y[missing] = np.nan
    return Y, np.nansum(data + R[..., None], axis=-1)


# --------------------------------------------------------------------------
# "strong" CPU path: same conditionals, restructured for
# the CPU - statistics hoisted, BLAS for the Gram / mean accumulation, banded
# LAPACK per column.  Complete Gaussian data only (the headline workload).
# Used by bench.py as the second CPU number; validated against w_step / v_step
# in tests/test_oracle_golden.py.
# --------------------------------------------------------------------------

def hoisted_stats(Y):
    """cnt, sum over replicates - computed once, outside the sweep."""
    cnt, ybar = replicate_stats(Y)
    if np.isnan(ybar).any() or (cnt != cnt.flat[0]).any():
        raise ValueError("strong CPU path: complete data only")
    return int(cnt.flat[0]), ybar


def w_step_strong(st, R, ybar, z=None, row0=0):
    """(row0: global index of the first row when st["W"] / ybar hold a block of rows; z then holds
    that block's normals only.)"""
    W, V = st["W"], st["V"]
    N, K = W.shape
    if row0 >= K:                                                 # no triangular head in this block
        Vf = V.reshape(-1, K)
        s = R / st["nu2"]
        L = np.linalg.cholesky(s * (Vf.T @ Vf) + np.eye(K) / st["sigma2"])
        Mpart = s * (ybar.reshape(N, -1) @ Vf)
        Z = (np.random.normal(size=N * K) if z is None else np.asarray(z)).reshape(N, K)
        W[:] = sla.cho_solve((L, True), Mpart.T).T + sla.solve_triangular(L.T, Z.T, lower=False).T
        return W
    assert row0 == 0
    Vf = V.reshape(-1, K)
    s = R / st["nu2"]
    G = s * (Vf.T @ Vf)
    Mpart = s * (ybar.reshape(N, -1) @ Vf)                       # (N,K)   one GEMM
    nz = sum(min(i + 1, K) for i in range(N))
    if z is None:
        z = np.random.normal(size=nz)
    zpos = 0
    for i in range(min(N, K)):                                    # triangular head
        d = i + 1
        L = np.linalg.cholesky(G[:d, :d] + np.eye(d) / st["sigma2"])
        W[i, :d] = sla.cho_solve((L, True), Mpart[i, :d]) + sla.solve_triangular(L.T, z[zpos:zpos + d], lower=False)
        zpos += d
    if N > K:                                                     # all other rows share one factor
        L = np.linalg.cholesky(G + np.eye(K) / st["sigma2"])
        Z = z[zpos:].reshape(N - K, K)
        W[K:] = sla.cho_solve((L, True), Mpart[K:].T).T + sla.solve_triangular(L.T, Z.T, lower=False).T
    return W


def v_step_strong(st, R, ybar, Delta, z=None, order=None):
    """Depth-major banded Cholesky per column (scipy.linalg.cholesky_banded).
    order: None (depth-major), an elimination order as depth-major indices (dense Cholesky of the
    permuted precision: the twisted kernel's declared order), or "spectral"
    (mvn_from_precision_spectral's square root, evaluated from G and the prior band directly)."""
    W, V = st["W"], st["V"]
    M, T, K = V.shape
    N = W.shape[0]
    s = R / st["nu2"]
    G = s * (W.T @ W)                                             # shared likelihood block
    mu = s * np.einsum("nk,nmt->mtk", W, ybar)                    # (M,T,K) depth-major
    tf1 = int(np.max(np.abs(np.subtract(*np.nonzero(Delta.T @ Delta)))))    # tf+1
    bw = tf1 * K
    n = T * K
    if z is None:
        z = np.random.normal(size=(M, n))
    DtD = [Delta[:, :T - d] * Delta[:, d:] for d in range(tf1 + 1)]         # (nD, T-d) coefficient products
    kk = np.arange(K)
    if isinstance(order, str) and order == "spectral":
        g, U = gram_eigensystem(G)
        mt = np.einsum("kl,mtk->mlt", U, mu)                      # (M,K,T) rotated right-hand sides
        zt = np.asarray(z).reshape(M, K, T)
        o = spectral_pivot_order(T, tf1)
        for j in range(M):
            lam = 1.0 / (st["lam2"] * st["Tau2"][j])
            P1 = np.zeros((T, T))
            for d in range(tf1 + 1):
                pd = lam @ DtD[d]
                P1[np.arange(d, T), np.arange(T - d)] = pd
                P1[np.arange(T - d), np.arange(d, T)] = pd
            Po = P1[np.ix_(o, o)]
            xt = np.empty((K, T))
            for k in range(K):
                C = np.linalg.cholesky(Po + g[k] * np.eye(T))
                xt[k, o] = sla.cho_solve((C, True), mt[j, k, o]) + sla.solve_triangular(C.T, zt[j, k], lower=False)
            V[j] = (U @ xt).T
        return V
    pvt = None if order is None else np.asarray(order)
    for j in range(M):
        lam = 1.0 / (st["lam2"] * st["Tau2"][j])
        ab = np.zeros((bw + 1, n))                                # lower form: ab[i-j, j] = A[i, j]
        for d in range(tf1 + 1):
            pd = lam @ DtD[d]                                     # prior band entry (t+d, t)
            for k in range(K):
                ab[d * K, k:(T - d) * K:K] += pd
        for a in range(K):                                        # likelihood block on every depth
            for k in range(K - a):
                ab[a, k::K] += G[k + a, k]
        if pvt is not None:                                       # declared elimination order: dense, permuted
            Qd = np.zeros((n, n))
            for a in range(bw + 1):
                Qd[np.arange(a, n), np.arange(n - a)] = ab[a, :n - a]
            Qd = Qd + np.tril(Qd, -1).T
            L = np.linalg.cholesky(Qd[np.ix_(pvt, pvt)])
            x = np.empty(n)
            x[pvt] = sla.cho_solve((L, True), mu[j].reshape(-1)[pvt]) + sla.solve_triangular(L.T, z[j], lower=False)
            V[j] = x.reshape(T, K)
            continue
        cb = sla.cholesky_banded(ab, lower=True)
        y = sla.cho_solve_banded((cb, True), mu[j].reshape(-1))
        # L' x = z  with L lower banded: L' is upper banded
        up = np.zeros_like(cb)
        for a in range(bw + 1):
            up[bw - a, a:] = cb[a, :n - a]
        x = sla.solve_banded((0, bw), up, z[j])
        V[j] = (y + x).reshape(T, K)
    return V


def v_column_system_strong(st, R, ybar, Delta, j):
    """Dense depth-major precision Q_j and mean part of column j for complete data (the system v_step_strong
    factorises: factor.py:377-408 with the hoisted statistics): for conditioning bounds and whitening checks."""
    W = st["W"]
    T = ybar.shape[2]
    K = W.shape[1]
    s = R / st["nu2"]
    G = s * (W.T @ W)
    lam = 1.0 / (st["lam2"] * st["Tau2"][j])
    Pm = np.asarray((Delta.T @ np.diag(lam) @ Delta) if not hasattr(Delta, "toarray") else (Delta.T.toarray() * lam) @ Delta.toarray())
    Q = np.kron(Pm, np.eye(K)) + np.kron(np.eye(T), G)            # depth-major: index t*K + k
    mu = s * np.einsum("nk,nt->tk", W, ybar[:, j, :]).reshape(-1)
    return Q, mu


def v_column_conds(st, R, ybar, Delta, cols=None):
    """2-norm condition numbers of the column systems (what bounds the agreement of two correct fp64 solves)."""
    M = st["V"].shape[0]
    cols = range(M) if cols is None else cols
    out = []
    for j in cols:
        Q, _ = v_column_system_strong(st, R, ybar, Delta, j)
        ev = np.linalg.eigvalsh(Q)
        out.append(ev[-1] / ev[0])
    return np.array(out)


# --------------------------------------------------------------------------
# fast_mvn: the dense branches and the dispatcher   (fast_mvn.py:49-60, :77-179)
# --------------------------------------------------------------------------

def sample_mvn_dense(Q, mu=None, mu_part=None, precision=False, chol_factor=False, z=None):
    """The `sparse=False` branches of sample_mvn_from_precision (fast_mvn.py:49-60) and
    sample_mvn_from_covariance (:126-142), reached through sample_mvn (:145-179; a scalar or vector Q means Q*I).
    precision: x = Lt^-1 z (+ Q^-1 mu_part | + mu), Lt = chol(Q)' (chol_factor: Q is the lower factor);
    covariance: x = L z (+ Q mu_part | + mu), L = chol(Q) (chol_factor: Q IS that factor)."""
    if not chol_factor and (np.isscalar(Q) or np.ndim(Q) == 1):
        dim = len(mu) if mu is not None else len(mu_part)
        Q = np.eye(dim) * Q
    Q = np.asarray(Q, float)
    n = Q.shape[0]
    if z is None:
        z = np.random.normal(size=n)
    if precision:
        Lt = np.linalg.cholesky(Q).T if not chol_factor else Q.T
        x = sla.solve_triangular(Lt, z, lower=False)
        if mu_part is not None:
            x = x + sla.cho_solve((Lt, False), mu_part)
        elif mu is not None:
            x = x + mu
        return x
    if chol_factor:
        L = Q
        Q = L @ L.T
    else:
        L = np.linalg.cholesky(Q)
    x = L @ z
    if mu_part is not None:
        x = x + Q @ mu_part
    elif mu is not None:
        x = x + mu
    return x


# --------------------------------------------------------------------------
# elliptical slice sampling          (elliptical_slice.py:59-124)
# --------------------------------------------------------------------------

def elliptical_slice(xx, prior, log_like_fn, cur_log_like=None, angle_range=0, ll_args=None, mu=None, info=None):
    """One elliptical-slice update.  prior: a prior sample (size D) or chol(Sigma, lower) (D x D: nu = L randn).
    Legacy global RNG in the reference's order: [randn] - rand (slice height) - rand (first angle; two when
    angle_range > 0) - one rand per shrink.  Returns (proposal on the slice, its log-likelihood)."""
    xx = np.array(xx, dtype=float)
    D = xx.size
    prior = np.asarray(prior, float)
    if prior.size == D:
        nu = prior.reshape(D)
    else:
        if prior.shape != (D, D):
            raise ValueError("prior must be a D-element sample or a D x D lower Cholesky factor")
        nu = (prior @ np.random.randn(D, 1)).T.reshape(xx.shape)
    mu = np.zeros(D) if mu is None else np.asarray(mu, float)
    if cur_log_like is None:
        cur_log_like = log_like_fn(xx, ll_args)
    hh = np.log(np.random.rand()) + cur_log_like
    if angle_range <= 0:
        phi = np.random.rand() * 2 * np.pi
        phi_min, phi_max = phi - 2 * np.pi, phi
    else:
        phi_min = -angle_range * np.random.rand()
        phi_max = phi_min + angle_range
        phi = np.random.rand() * (phi_max - phi_min) + phi_min
    nev = 0
    while True:
        prop = (xx - mu) * np.cos(phi) + nu * np.sin(phi) + mu
        cur_log_like = log_like_fn(prop, ll_args)
        nev += 1
        if cur_log_like >= hh:
            break
        if phi > 0:
            phi_max = phi
        elif phi < 0:
            phi_min = phi
        else:
            break                       # shrunk to the current position and still rejected (:112-116)
        phi = np.random.rand() * (phi_max - phi_min) + phi_min
    if info is not None:
        info["evaluations"] = nev
        info["phi"] = phi
    return prop, cur_log_like


# --------------------------------------------------------------------------
# generalized analytic slice sampling   (gass.py:13-130)
# --------------------------------------------------------------------------

def gass_valid_grid(x0, v, A, c, ngrid=100):
    """The candidate angles of one GASS update before any subsampling (gass.py:38-83): for every constraint row
    a = A x0, b = A v the valid angles are an interval or the complement of one; the reference intersects them
    numerically on linspace(-pi, pi, 10000).  Returns (grid, restricted); no constraint restricts the ellipse:
    linspace(-pi, pi, ngrid)."""
    a, b = A @ x0, A @ v
    sqrt_term = a ** 2 + b ** 2 - c ** 2
    eps = 1e-6
    concerning = (sqrt_term >= 0) & (a != -c)
    if not np.any(concerning):
        return np.linspace(-np.pi, np.pi, ngrid), False
    denom = a + c
    rt = np.sqrt(sqrt_term[concerning])
    theta1 = 2 * np.arctan((b[concerning] + rt) / denom[concerning])
    theta2 = 2 * np.arctan((b[concerning] - rt) / denom[concerning])
    comp = a[concerning] ** 2 < c[concerning] ** 2
    grid = np.linspace(-np.pi, np.pi, 10000)
    for t1, t2 in zip(theta1[comp], theta2[comp]):          # convex: outside [min, max]
        grid = grid[(grid <= min(t1, t2)) | (grid >= max(t1, t2))]
    if np.any(~comp):                                       # concave: inside every [min, max]
        t1i, t2i = theta1[~comp], theta2[~comp]
        lo = np.minimum(t1i, t2i).max() + eps
        hi = np.maximum(t1i, t2i).min() - eps
        grid = grid[(grid >= lo) & (grid <= hi)]
    return grid, True


def gass(x, v, loglikelihood, Constraints, cur_ll=None, mu=None, ll_args=None, ngrid=100, info=None, rng=None):
    """One GASS update of x under the linear constraints  Constraints[:, :-1] x >= Constraints[:, -1].
    v: the proposal the reference draws with sample_mvn (gass.py:24) - passed in, because which fast_mvn branch
    draws it is the caller's choice (sample_mvn_dense / mvn_from_precision above).  RNG order after v:
    rand (slice height, drawn BEFORE v in the reference: the caller draws v after calling np.random.random
    once - see tests), [choice(grid, ngrid) if the valid grid is longer than ngrid], choice(#accepted).
    rng: a RandomState to draw them from (default: the global legacy generator, as the reference)."""
    rng = np.random if rng is None else rng
    x = np.asarray(x, float)
    if cur_ll is None:
        cur_ll = loglikelihood(x, ll_args)
    ll = cur_ll + np.log(rng.random_sample())
    v = v() if callable(v) else np.asarray(v, float)
    mu = np.zeros_like(x) if mu is None else np.asarray(mu, float)
    A, cvec = Constraints[:, :-1], Constraints[:, -1]
    if not np.all(A @ x >= cvec):
        raise ValueError("invalid starting point")
    x0 = x - mu
    grid, _ = gass_valid_grid(x0, v, A, cvec - A @ mu, ngrid)
    if len(grid) == 0:
        options, opt_ll = [], []
    else:
        if len(grid) > ngrid:
            grid = rng.choice(grid, size=ngrid, replace=False)
        options = x0[None] * np.cos(grid[:, None]) + v[None] * np.sin(grid[:, None]) + mu[None]
        opt_ll = loglikelihood(options, ll_args)
        keep = opt_ll >= ll
        options, opt_ll = options[keep], opt_ll[keep]
    if info is not None:
        info["grid"] = len(grid)
        info["accepted"] = len(options)
        info["slice"] = ll
    if len(options) > 0:
        sel = rng.choice(len(options))
        return options[sel], opt_ll[sel]
    return x, cur_ll


# --------------------------------------------------------------------------
# constrained non-conjugate model: GASS per row / per column   (factor.py:665-855)
# --------------------------------------------------------------------------

def constrained_w_constraints(V, Cons, ndims, Row_constraints=None):
    """Constraint matrix of one row of W given V (factor.py:713-727): for every column j and constraint q the row
    (Cons_q V_j)[:ndims] with bound Cons[q, -1] - ordered column-major over (j, q) - then the fixed row constraints."""
    A, C = Cons[:, :-1], Cons[:, -1:]
    M = V.shape[0]
    AV = np.einsum("qt,jtk->jqk", A, V)[..., :ndims].reshape(-1, ndims)
    out = np.concatenate([AV, np.tile(C, (M, 1))], axis=1)
    if Row_constraints is not None:
        out = np.concatenate([out, np.concatenate([Row_constraints[:, :ndims], Row_constraints[:, -1:]], axis=1)], axis=0)
    return out


def constrained_v_constraints(W, Cons, T):
    """Constraint matrix of one column of V (k-major vector) given W (factor.py:848-855): rows (i, q) ->
    coefficient W[i,k] Cons[q,t] at position k*T + t."""
    A, C = Cons[:, :-1], Cons[:, -1:]
    N, K = W.shape
    J = A.shape[0]
    Am = (A[None, :, None, :] * W[:, None, :, None]).reshape(N * J, K * T)
    return np.concatenate([Am, np.tile(C, (N, 1))], axis=1)


def poisson_curves_loglik(Y, tau, link):
    """nansum of log Poisson(y | lambda(tau)) over a block of curves (the row / column likelihood of
    examples/poisson_tensor_filtering.py:26-37); the y-only term -lgamma(y+1) is dropped (it cancels in every
    comparison GASS makes).  tau: (..., T); Y: (..., T) or (..., T, R)."""
    Y4 = Y[..., None] if Y.ndim == tau.ndim else Y
    obs = ~np.isnan(Y4)
    y = np.where(obs, Y4, 0.0)
    t = tau[..., None]
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        if link == "log":
            term = y * t - np.exp(t)
        else:
            term = np.where(t > 0, y * np.log(np.where(t > 0, t, 1.0)) - t, -np.inf)
    return float(np.where(obs, term, 0.0).sum())


def constrained_w_step(st, Y, Cons, link="identity", ngrid=100, z=None, rngs=None, Row_constraints=None, info=None):
    """_resample_W_i for every row (factor.py:665-711, no EP approximation): prior precision I/sigma2 on the free
    entries (z[i]: the normals of row i's proposal; sqrt(sigma2) z), constraints from the current V, Poisson row
    likelihood.  All rows see the same V (the reference maps them over a worker pool).  rngs[i]: RandomState of row i
    (slice height, subsample, selection)."""
    W, V = st["W"], st["V"]
    N, K = W.shape
    out = W.copy()
    for i in range(N):
        d = min(K, i + 1)
        Ci = constrained_w_constraints(V, Cons, d, Row_constraints)
        Vi = V[:, :, :d]

        def ll(w, _):
            w = np.atleast_2d(w)
            r = np.array([poisson_curves_loglik(Y[i], np.einsum("jtk,k->jt", Vi, wk), link) for wk in w])
            return r if r.size > 1 else r[0]
        rng = None if rngs is None else rngs[i]
        if z is None:       # drawn AFTER the slice height, from the row's own stream (gass.py:21-24)
            prop = lambda d=d, rng=rng: np.sqrt(st["sigma2"]) * (np.random if rng is None else rng).normal(size=d)
        else:
            prop = np.sqrt(st["sigma2"]) * np.asarray(z[i], float)[:d]
        inf = {}
        new, _ = gass(W[i, :d], prop, ll, Ci, ngrid=ngrid, info=inf, rng=rng)
        out[i, :d] = new
        if info is not None:
            info.setdefault("grid", []).append(inf["grid"])
            info.setdefault("accepted", []).append(inf["accepted"])
    st["W"] = out
    return out


def constrained_v_step(st, Y, Delta, Cons, link="identity", ngrid=100, perm="depth", z=None, rngs=None, info=None):
    """_resample_V_j for every column (factor.py:759-846, no EP approximation): prior precision
    I_K (x) Delta' Lambda_j Delta (k-major), proposal drawn in the declared order `perm`, constraints from the
    current W, Poisson column likelihood."""
    W, V = st["W"], st["V"]
    M, T, K = V.shape
    n = K * T
    if isinstance(perm, str):
        p = twisted_perm(K, T, band_halfwidth(Delta.T @ Delta) - 1) if perm == "twist" else \
            (depth_major_perm(K, T) if perm == "depth" else np.arange(n))
    else:
        p = np.asarray(perm)
    Cv = constrained_v_constraints(W, Cons, T)
    out = V.copy()
    for j in range(M):
        Q = np.kron(np.eye(K), prior_precision_1d(Delta, st["lam2"], st["Tau2"][j]))
        rng = None if rngs is None else rngs[j]
        if z is None:
            v = lambda Q=Q, rng=rng: mvn_from_precision(Q, perm=p, z=(np.random if rng is None else rng).normal(size=n))
        else:
            v = mvn_from_precision(Q, perm=p, z=np.asarray(z[j], float))

        def ll(vec, _):
            vec = np.atleast_2d(vec)
            r = np.array([poisson_curves_loglik(Y[:, j], np.einsum("nk,kt->nt", W, vk.reshape(K, T)), link) for vk in vec])
            return r if r.size > 1 else r[0]
        inf = {}
        new, _ = gass(V[j].T.reshape(-1), v, ll, Cv, ngrid=ngrid, info=inf, rng=rng)
        out[j] = new.reshape(K, T).T
        if info is not None:
            info.setdefault("grid", []).append(inf["grid"])
            info.setdefault("accepted", []).append(inf["accepted"])
    st["V"] = out
    return out


# --------------------------------------------------------------------------
# non-conjugate model: joint elliptical-slice updates of W and V   (factor.py:155-228, :567-590)
# --------------------------------------------------------------------------

def pack_W(W):
    """Free entries of W: lower triangle of the leading square block, then the rows below (factor.py:155-174)."""
    N, K = W.shape
    h = min(K, N)
    return np.concatenate([W[np.tril_indices(h)], W[h:].reshape(-1)])


def unpack_W(vec, W):
    N, K = W.shape
    h = min(K, N)
    nt = h * (h + 1) // 2
    W[np.tril_indices(h)] = vec[:nt]
    W[h:] = vec[nt:].reshape(N - h, K)
    return W


def poisson_loglik(W, V, Y, link="log"):
    """sum over observed cells / replicates of log Poisson(y | lambda), lambda = exp(w.v) or w.v
    (the likelihood of examples/poisson_tensor_filtering.py:26-37, scipy.stats.poisson.logpmf + nansum)."""
    from scipy.special import gammaln
    eta = np.einsum("nk,mtk->nmt", W, V)
    Y4 = Y[..., None] if Y.ndim == 3 else Y
    obs = ~np.isnan(Y4)
    y = np.where(obs, Y4, 0.0)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        if link == "log":
            term = y * eta[..., None] - np.exp(eta)[..., None]
        else:
            term = np.where(eta[..., None] > 0, y * np.log(np.where(eta > 0, eta, 1.0))[..., None] - eta[..., None], -np.inf)
    term = np.where(obs, term - gammaln(y + 1.0), 0.0)
    return float(term.sum())


def family_loglik(W, V, Y, family, param=None):
    """Log-likelihoods a user of the reference would pass to NonconjugateBayesianTensorFiltering as callbacks
    (factor.py:567-570 takes any `loglikelihood(W, V, data)`), for the families the build evaluates on the device,
    eta = w.v per cell, summed over the observed replicates (NaN = missing), normalising terms included:
      "bernoulli_logit"  y in {0, 1}:  y eta - log(1 + e^eta)                      (scipy.stats.bernoulli.logpmf(y, ilogit(eta)))
      "gaussian"         norm.logpdf(y, eta, sqrt(param)), param = the variance
      "negbin_logit"     nbinom.logpmf(y, param, 1 - ilogit(eta)), param = the rate r"""
    from scipy.special import gammaln
    eta = np.einsum("nk,mtk->nmt", W, V)[..., None]
    Y4 = Y[..., None] if Y.ndim == 3 else Y
    obs = ~np.isnan(Y4)
    y = np.where(obs, Y4, 0.0)
    sp = np.logaddexp(0.0, eta)                             # softplus
    if family == "bernoulli_logit":
        term = y * eta - sp
    elif family == "gaussian":
        term = -0.5 * (y - eta) ** 2 / param - 0.5 * np.log(2 * np.pi * param)
    elif family == "negbin_logit":
        term = gammaln(y + param) - gammaln(param) - gammaln(y + 1.0) + y * eta - (y + param) * sp
    else:
        raise ValueError(family)
    return float(np.where(obs, term, 0.0).sum())


def nc_loglik(W, V, Y, link, param=None):
    """dispatcher: a callable (the reference's own interface, factor.py:567-570: loglikelihood(W, V, data)), the Poisson
    links of poisson_loglik, or a family of family_loglik"""
    if callable(link):
        return link(W, V, Y)
    return poisson_loglik(W, V, Y, link) if link in ("log", "identity") else family_loglik(W, V, Y, link, param)


def nonconjugate_w_step(st, Y, link="log", z=None, info=None, param=None):
    """factor.py:572-581: prior sample of the packed W (precision I/sigma2: nu = sqrt(sigma2) z), one elliptical
    slice over all of it with the full-tensor likelihood."""
    W, V = st["W"], st["V"]
    cur = pack_W(W)
    if z is None:
        z = np.random.normal(size=cur.size)
    nu = np.sqrt(st["sigma2"]) * np.asarray(z)

    def ll(vec, _):
        return nc_loglik(unpack_W(vec, np.zeros_like(W)), V, Y, link, param)
    new, _ = elliptical_slice(cur, nu, ll, info=info)
    unpack_W(new, W)
    return W


def nonconjugate_v_step(st, Y, Delta, link="log", perm="depth", z=None, info=None, param=None):
    """factor.py:583-590: prior sample of the packed V (block-diagonal precision I_K (x) Delta' Lambda_j Delta per
    column, k-major, factor.py:176-195; drawn column by column in the declared order `perm`), one elliptical
    slice over all of V."""
    W, V = st["W"], st["V"]
    M, T, K = V.shape
    n = K * T
    if isinstance(perm, str):
        p = twisted_perm(K, T, band_halfwidth(Delta.T @ Delta) - 1) if perm == "twist" else \
            (depth_major_perm(K, T) if perm == "depth" else np.arange(n))
    else:
        p = np.asarray(perm)
    if z is None:
        z = np.random.normal(size=M * n)
    z = np.asarray(z).reshape(M, n)
    nu = np.empty(M * n)
    for j in range(M):
        Q = np.kron(np.eye(K), prior_precision_1d(Delta, st["lam2"], st["Tau2"][j]))
        nu[j * n:(j + 1) * n] = mvn_from_precision(Q, perm=p, z=z[j])
    cur = np.concatenate([V[j].T.reshape(-1) for j in range(M)])

    def ll(vec, _):
        Vp = np.stack([vec[j * n:(j + 1) * n].reshape(K, T).T for j in range(M)])
        return nc_loglik(W, Vp, Y, link, param)
    new, _ = elliptical_slice(cur, nu, ll, info=info)
    for j in range(M):
        V[j] = new[j * n:(j + 1) * n].reshape(K, T).T
    return V
