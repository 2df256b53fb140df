"""Parity of the HIP path (through the C ABI and the reference-shaped Python API)
against the golden fixtures captured from the real reference, and against the
oracle on seeded inputs.  Needs an MI355X: run with -m gpu.

Tolerances (north_star: 1e-5 relative per half-sweep, max|diff| / max|ref|):
  W step : 1e-10   (K x K systems, cond ~1e2)
  V step : 1e-6    (cond(Q) up to ~1e9 after lam2 has collapsed; two correct fp64
                    factorisations only agree to ~cond * eps)
"""
import numpy as np
import pytest
from conftest import state_from, relerr

pytestmark = pytest.mark.gpu

W_TOL = 1e-10
V_TOL = 1e-6


def gaussian_model(g, prefix, **kw):
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    N, M, T, R, K, tf = [int(x) for x in g["dims"]]
    st = state_from(g, prefix)
    model = GaussianBayesianTensorFiltering(
        N, M, T, nembeds=K, tf_order=tf, sigma2_init=st["sigma2"], lam2_init=st["lam2"],
        nu2_init=st["nu2"], W_init=st["W"], V_init=st["V"], Tau2_init=st["Tau2"], **kw)
    for k in ("Tau2_a", "Tau2_b", "Tau2_c"):
        setattr(model, k, st[k].copy())
    model.lam2_a = st["lam2_a"]
    return model, st


GAUSS = [("g1_c1_heldout.npz", 100), ("g2_c2_complete.npz", 300), ("g3_partial_reps.npz", 400)]


@pytest.mark.parametrize("name,seed", GAUSS)
def test_w_step_vs_reference(golden, name, seed):
    g = golden(name)
    model, _ = gaussian_model(g, "s0_")
    np.random.seed(seed)                      # same legacy stream the reference consumed
    model._resample_W(g["Y"])
    assert relerr(model.W, g["W_after"]) < W_TOL
    # G1: held-out whole curves -> complete-data kernels plus corrections; G2 complete; G3 (replicates missing at
    # single depths) the per-cell weighted form
    assert model.likelihood_form() == {"g1": "curve_counts", "g2": "complete", "g3": "weighted"}[name[:2]]


@pytest.mark.parametrize("kernel", ["twist", "fast", "generic"])
@pytest.mark.parametrize("name,seed", GAUSS)
def test_v_step_vs_reference(golden, name, seed, kernel):
    """The banded samplers: the twisted two-chain kernel (default), the single-chain LDS
    LDL' kernel and the any-size generic one."""
    g = golden(name)
    model, _ = gaussian_model(g, "s0_", sampler={"twist": "banded", "fast": "chain", "generic": "generic"}[kernel])
    model.W = g["W_after"]
    np.random.seed(seed + 1)
    model._resample_V(g["Y"])
    T, K = model.ndepth, model.nembeds
    from oracle import btf_oracle as orc
    if kernel == "twist":        # the declared elimination order of the default kernel
        assert model.likelihood_form() == {"g1": "curve_counts", "g2": "complete", "g3": "weighted"}[name[:2]]
        assert np.array_equal(model.v_order(), orc.twisted_order(K, T, 2))
        assert relerr(model.V, g["V_after_twist"]) < V_TOL
    else:
        assert np.array_equal(model.v_order(), np.arange(K * T))
        assert relerr(model.V, g["V_after_depth"]) < V_TOL


def test_illconditioned_state(golden):
    """lam2 at its floor, Tau2 over 13 decades: cond(Q) up to 1e13, so only the control
    flow and ~cond*eps agreement can be asked for."""
    g = golden("g5_illcond.npz")
    model, _ = gaussian_model(g, "s0_")
    np.random.seed(600)
    model._resample_W(g["Y"])
    assert relerr(model.W, g["W_after"]) < W_TOL
    np.random.seed(601)
    model._resample_V(g["Y"])
    assert relerr(model.V, g["V_after_twist"]) < 2e-3


def test_jitter_retry_matches_reference_schedule(golden):
    import ctypes as C
    g = golden("g5_illcond.npz")
    model, _ = gaussian_model(g, "retry_s0_")
    np.random.seed(601)
    model._resample_V(g["Y"])
    V = model.V.copy()
    tries = np.zeros(model.ncols, dtype=np.int32)
    model._ctx.call("btf_get_V_attempts", tries.ctypes.data_as(C.POINTER(C.c_int32)))
    assert np.array_equal(tries, g["retry_tries_twist"])
    assert relerr(V, g["retry_V_after_twist"]) < 1e-5
    model, _ = gaussian_model(g, "retry_s0_", sampler="chain")     # single-chain kernel: depth-major fixture
    np.random.seed(601)
    model._resample_V(g["Y"])
    model._ctx.call("btf_get_V_attempts", tries.ctypes.data_as(C.POINTER(C.c_int32)))
    assert np.array_equal(tries, g["retry_tries"])
    assert relerr(model.V, g["retry_V_after"]) < 1e-5


def test_not_positive_definite_is_reported(golden):
    """Where the reference spins forever (fast_mvn.py:69-72) the C ABI returns BTF_ENOTPD."""
    from functionalmf_amd._native import NotPositiveDefiniteError
    g = golden("g1_c1_heldout.npz")
    model, _ = gaussian_model(g, "s0_")
    model.Tau2[4, 7] = -1e-3                  # grossly indefinite prior term
    np.random.seed(0)
    model._resample_V(g["Y"])
    with pytest.raises(NotPositiveDefiniteError) as e:
        model.sync()
    assert e.value.index == 4


def test_sse_and_nu2(golden):
    g = golden("g1_c1_heldout.npz")
    model, st = gaussian_model(g, "h0_")
    np.random.seed(200)
    model._resample_nu2(g["Y"])
    assert abs(model.nu2 - g["h_nu2"]) / g["h_nu2"] < 1e-12
    np.random.seed(201)
    model._resample_sigma2()
    assert abs(model.sigma2 - g["h_sigma2"]) / g["h_sigma2"] < 1e-12
    np.random.seed(202)
    model._resample_Tau2()
    for k in ("Tau2", "Tau2_a", "Tau2_b", "Tau2_c"):
        assert relerr(getattr(model, k), g["h_tau_" + k]) < 1e-11, k
    np.random.seed(203)
    model._resample_lam2()
    assert abs(model.lam2 - g["h_lam2"]) / g["h_lam2"] < 1e-11
    assert abs(model.lam2_a - g["h_lam2_a"]) / g["h_lam2_a"] < 1e-11


def test_construction_matches_reference(golden):
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    g = golden("g1_c1_heldout.npz")
    N, M, T, R, K, tf = [int(x) for x in g["dims"]]
    np.random.seed(11)
    model = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=tf, sigma2_init=0.5,
                                            lam2_init=0.1, nu2_init=1.0, nthreads=1)
    assert relerr(model.W, g["init_W"]) < 1e-13
    assert relerr(model.Tau2, g["init_Tau2"]) < 1e-13
    assert relerr(model.V, g["init_V"]) < 1e-6
    assert model.Delta.shape == (3 * T - 1, T)


def test_run_gibbs_chain_vs_reference(golden):
    """3 burn-in + 2*4 kept sweeps from the reference's seed: the whole chain (all six
    result arrays) reproduces the reference run to 1e-5."""
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    g = golden("g6_c1_chain.npz")
    Y = g["Y"]
    N, M, T, R = Y.shape
    K = g["init_W"].shape[1]
    np.random.seed(21)
    model = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, sigma2_init=0.5,
                                            lam2_init=0.1, nu2_init=1.0, nthreads=1)
    np.random.seed(22)
    res = model.run_gibbs(Y, nburn=3, nthin=2, nsamples=4, verbose=False)
    assert set(res) == {"W", "V", "sigma2", "lam2", "Tau2", "nu2"}
    for k in res:
        assert res[k].shape == g["rest_" + k].shape, k
        assert relerr(res[k], g["rest_" + k]) < 1e-5, k        # reference chain under the declared (twisted) order


def test_banded_sampler_vs_oracle():
    """The stand-alone fast_mvn entry point on random banded SPD systems."""
    from functionalmf_amd.fast_mvn import sample_mvn_from_precision
    from oracle import btf_oracle as orc
    rs = np.random.RandomState(5)
    for n, bw in ((40, 3), (64, 9), (150, 15)):
        L = np.tril(rs.normal(size=(n, n))) * (np.abs(np.subtract.outer(np.arange(n), np.arange(n))) <= bw)
        Q = L @ L.T + 0.5 * np.eye(n)
        Q[np.abs(np.subtract.outer(np.arange(n), np.arange(n))) > bw] = 0
        Q = Q + 2 * bw * np.eye(n)
        mu, z = rs.normal(size=n), rs.normal(size=n)
        x = sample_mvn_from_precision(Q, mu_part=mu, z=z)
        ref = orc.mvn_from_precision(Q, mu_part=mu, z=z)
        assert relerr(x, ref) < 1e-11


def test_device_rng_moments():
    """rng='device': Philox normals; the W draw must have the analytic conditional mean
    and covariance (homoskedastic complete data => every row >= K shares one Q)."""
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    N, M, T, R, K = 4096, 6, 8, 2, 3
    rs = np.random.RandomState(0)
    V = rs.normal(size=(M, T, K))
    W0 = rs.normal(size=(N, K))
    Y = np.einsum("nk,mtk->nmt", np.ones((N, 1)) * W0[:1], V)[..., None] + rs.normal(0, 1.0, size=(N, M, T, R)) * 0
    np.random.seed(0)
    model = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=1, sigma2_init=2.0, lam2_init=0.1,
                                            nu2_init=3.0, V_init=V, W_init=W0, rng="device", device_seed=9)
    model._resample_W(Y)
    W = model.W[K:].copy()
    Vf = V.reshape(-1, K)
    Q = R / 3.0 * Vf.T @ Vf + np.eye(K) / 2.0
    mean = np.linalg.solve(Q, R / 3.0 * Vf.T @ Y[0].mean(-1).reshape(-1))
    cov = np.linalg.inv(Q)
    assert np.abs(W.mean(0) - mean).max() < 5 * np.sqrt(cov.diagonal().max() / W.shape[0])
    assert relerr(np.cov(W.T), cov) < 0.1
    # and a second call draws different numbers
    model._resample_W(Y)
    assert np.abs(model.W[K:] - W).max() > 1e-3


# ------------------------------------------------------------------ Binomial / PG
def binomial_model(g, **kw):
    from functionalmf_amd.factor import BinomialBayesianTensorFiltering
    N, M, T, R, K, tf = [int(x) for x in g["dims"]]
    st = state_from(g, "s0_")
    model = BinomialBayesianTensorFiltering(
        N, M, T, nembeds=K, tf_order=tf, sigma2_init=st["sigma2"], lam2_init=st["lam2"],
        W_init=st["W"], V_init=st["V"], Tau2_init=st["Tau2"], **kw)
    model.nu2 = st["nu2"]            # = 1/omega, as the reference holds it (factor.py:460)
    return model


@pytest.mark.parametrize("tag", ["nan", "full"])
def test_binomial_half_sweeps_given_omega(golden, tag):
    """Q1 (tag=full: no NaN => rows >= K reuse row K-1's omega) and Q2 (every column
    reuses the omega of the last pattern change) are reproduced."""
    g = golden("g4_binomial_%s.npz" % tag)
    data = (g["Ysucc"], g["Ntrials"])
    model = binomial_model(g)
    np.random.seed(500)
    model._resample_W(data)
    assert relerr(model.W, g["W_after"]) < W_TOL
    np.random.seed(501)
    model._resample_V(data)
    assert relerr(model.V, g["V_after_twist"]) < V_TOL


def test_binomial_exact_mode_differs_from_reference_quirk(golden):
    g = golden("g4_binomial_full.npz")
    data = (g["Ysucc"], g["Ntrials"])
    model = binomial_model(g, compat="exact")
    np.random.seed(500)
    model._resample_W(data)
    from oracle import btf_oracle as orc
    st = state_from(g, "s0_")
    # textbook conditional: every row uses its own omega = force the refresh branch
    Yk = orc.binomial_kappa(g["Ysucc"], g["Ntrials"], st["nu2"])
    W = st["W"].copy()
    N, K = W.shape
    Vf = st["V"].reshape(-1, K)
    zpos = 0
    for i in range(N):
        d = min(i + 1, K)
        c = 1.0 / st["nu2"][i].reshape(-1)
        Q = (Vf[:, :d] * c[:, None]).T @ Vf[:, :d] + np.eye(d) / st["sigma2"]
        m = (Vf[:, :d] * c[:, None]).T @ Yk[i].reshape(-1)
        L = np.linalg.cholesky(Q)
        W[i, :d] = np.linalg.solve(Q, m) + np.linalg.solve(L.T, g["z_W"][zpos:zpos + d])
        zpos += d
    assert relerr(model.W, W) < 1e-10
    assert relerr(model.W, g["W_after"]) > 1e-3


PG_BATCH_MODES = {"auto": 0, "exact": 1, "series": 2, "ref64": 3, "exact_allf64": 4}


def pg_batch(b, psi, seed, exact=False, mode=None):
    """mode (include/btf.h, btf_pg_batch_mode): "auto" (default) / "exact" / "series" / "ref64" (the f64 Devroye
    kernel of round 2) / "exact_allf64" (the flat exact sampler with every trip repeated in f64)."""
    import ctypes as C
    from functionalmf_amd import _native
    lib = _native.load()
    b = _native.as_f64(b)
    psi = _native.as_f64(psi)
    out = np.empty_like(b)
    m = PG_BATCH_MODES[mode] if mode is not None else (1 if exact else 0)
    rc = lib.btf_pg_batch_mode(0, b.size, _native.dptr(b), _native.dptr(psi), C.c_uint64(seed), m, _native.dptr(out))
    assert rc == 0, lib.btf_last_error(None)
    return out


def _cumulants3(x):
    m = x.mean()
    d = x - m
    return m, (d * d).mean(), (d * d * d).mean()


@pytest.mark.parametrize("b,c", [(b, c) for b in [3, 4, 8, 4.3] for c in [0.0, 1.0, 5.0, 20.0]] +
                         [(4, 3.0), (4, 3.3), (4, 8.0), (3, 6.4), (13, 0.5)])
def test_pg_series_sampler_against_exact_sampler(b, c):
    """A/B of the two device samplers (BTF_OPT_PG_EXACT): the opt-in sum-of-gammas series (2 drawn Gamma(b)
    terms + 2|psi|/2pi - the extra pairs straddle the |psi| = pi, 2 pi steps of that count -, f32-transcendental
    variates, the rest through a moment-matched Wilson-Hilferty gamma) against Devroye's exact sampler summed
    floor(b) times (+ a 128-term f64 series for a fractional part) - the algorithm pypolyagamma runs for
    integer b (factor.py:459).  b = 4 is config C4's trial count.  2e6 draws each: two-sample KS and
    Anderson-Darling (tail-weighted), and the first three cumulants - the exact sampler's against the closed
    forms, the series sampler's (and its fourth) against the exact sampler's within Monte-Carlo error."""
    from scipy.stats import ks_2samp, anderson_ksamp
    from oracle import btf_oracle as orc
    n = 2000000 if b == int(b) else 400000
    x = pg_batch(np.full(n, float(b)), np.full(n, c), seed=9000 + int(10 * b) + int(c), mode="series")
    y = pg_batch(np.full(n, float(b)), np.full(n, c), seed=19000 + int(10 * b) + int(c), exact=True)
    assert np.all(x > 0) and np.all(y > 0)
    assert ks_2samp(x, y).pvalue > 1e-3
    ad = anderson_ksamp([x[:400000], y[:400000]])
    assert ad.statistic < ad.critical_values[-1], (ad.statistic, ad.critical_values)      # 0.1 % level
    m, v = float(orc.pg_mean(b, c)), float(orc.pg_var(b, c))
    mx, vx, tx = _cumulants3(x)
    my, vy, ty = _cumulants3(y)
    se = np.sqrt(v / n)
    assert abs(my - m) < 5 * se and abs(mx - m) < 5 * se, (mx, my, m)
    k4 = ((y - my) ** 4).mean()
    sev = np.sqrt((k4 - v * v) / n)
    assert abs(vy - v) < 6 * sev and abs(vx - v) < 6 * sev + 2e-3 * v, (vx, vy, v)
    se3 = np.sqrt(((y - my) ** 6).mean() / n) * 1.5
    assert abs(tx - ty) < 6 * se3 + 0.02 * abs(ty), (tx, ty)
    # fourth cumulant (the tails the moment-matched remainder could lose)
    k4x, k4y = ((x - mx) ** 4).mean() - 3 * vx * vx, k4 - 3 * vy * vy
    se4 = np.sqrt(((y - my) ** 8).mean() / n) * 1.5
    assert abs(k4x - k4y) < 6 * se4 + 0.03 * abs(k4y), (k4x, k4y)


@pytest.mark.parametrize("b,c", [(1, 0.0), (1, 2.0), (2, 3.0), (4, 0.0), (4, 1.0), (4, 3.0), (4, 3.125), (4, 3.3), (4, 8.0),
                                 (4, 40.0), (8, 0.5), (31, 1.5), (1, 120.0)])
def test_pg_flat_exact_sampler_against_f64_devroye(b, c):
    """The default sampler of integer counts - the flat per-lane work-queue kernel (csrc/btf_pg_exact.h: f32 squeeze,
    f64 decisions inside the guard bands) - against the f64 Devroye kernel it replaces (mode "ref64": libm
    throughout; the algorithm of pypolyagamma, factor.py:459).  Same distribution: 2e6 draws each (own seeds),
    two-sample KS and Anderson-Darling, first four cumulants against the closed forms / each other.  |psi| = 3.125
    sits on the switch between the two truncated inverse-Gaussian samplers (z = 1/t), 3.3 and up take
    Michael-Schucany-Haas, 40 and 120 have a vanishing exponential piece."""
    from scipy.stats import ks_2samp, anderson_ksamp
    from oracle import btf_oracle as orc
    n = 2000000
    x = pg_batch(np.full(n, float(b)), np.full(n, c), seed=700 + int(10 * b) + int(c), mode="exact")
    y = pg_batch(np.full(n, float(b)), np.full(n, c), seed=1700 + int(10 * b) + int(c), mode="ref64")
    assert np.all(x > 0) and np.all(np.isfinite(x))
    assert ks_2samp(x, y).pvalue > 1e-3
    ad = anderson_ksamp([x[:400000], y[:400000]])
    assert ad.statistic < ad.critical_values[-1], (ad.statistic, ad.critical_values)
    m, v = float(orc.pg_mean(b, c)), float(orc.pg_var(b, c))
    mx, vx, tx = _cumulants3(x)
    my, vy, ty = _cumulants3(y)
    assert abs(mx - m) < 5 * np.sqrt(v / n), (mx, m)
    k4 = ((y - my) ** 4).mean()
    assert abs(vx - v) < 6 * np.sqrt((k4 - v * v) / n), (vx, v)
    se3 = np.sqrt(((y - my) ** 6).mean() / n) * 1.5
    assert abs(tx - ty) < 6 * se3, (tx, ty)
    k4x, k4y = ((x - mx) ** 4).mean() - 3 * vx * vx, k4 - 3 * vy * vy
    assert abs(k4x - k4y) < 6 * np.sqrt(((y - my) ** 8).mean() / n) * 1.5, (k4x, k4y)


def test_pg_flat_exact_sampler_f64_fallback_makes_the_same_decisions():
    """Every trip of the flat sampler evaluated twice: squeezed (f32, f64 only inside the guard bands) and wholly in
    f64 from the same words (mode "exact_allf64").  The accept / reject decisions must agree trip by trip - a single
    different decision changes a draw by O(1) - so the two runs return the same omega for every element up to the
    f32 rounding of the accepted variates.  Mixed counts and |psi| from 0 to 60 in one batch (lists of four cells per
    lane with different counts; both inverse-Gaussian samplers; empty cells)."""
    rs = np.random.RandomState(5)
    n = 1 << 20
    b = rs.randint(0, 9, size=n).astype(float)
    b[rs.rand(n) < 0.02] = 31.0
    psi = rs.normal(size=n) * rs.choice([0.3, 1.5, 3.2, 8.0, 30.0], size=n)
    psi[:1000] = 0.0
    psi[1000:2000] = 3.125                      # z = 1/t exactly
    x = pg_batch(b, psi, seed=99, mode="exact")
    y = pg_batch(b, psi, seed=99, mode="exact_allf64")
    assert np.all(x[b == 0] == 0) and np.all(y[b == 0] == 0)
    pos = b > 0
    assert np.all(x[pos] > 0)
    rel = np.abs(x[pos] - y[pos]) / y[pos]
    # (f32 rounding of the variates: ~1e-7 each; rare draws of the z > 1/t branch lose up to ~1e-4 to v_cos_f32)
    assert rel.max() < 1e-3 and (rel > 2e-5).mean() < 1e-5, (rel.max(), int((rel > 2e-5).sum()))
    # ... and the f64 trips sample PG: z-scores of this batch against the closed-form moments
    from oracle import btf_oracle as orc
    zs = (y[pos] - orc.pg_mean(b[pos], psi[pos])) / np.sqrt(orc.pg_var(b[pos], psi[pos]))
    assert abs(zs.mean()) < 5 / np.sqrt(pos.sum()) and abs(zs.var() - 1) < 0.01, (zs.mean(), zs.var())


def test_pg_batch_classes_and_layout_independence():
    """Which sampler an element goes through (pg_class_of): in the default mode integer counts up to 32 are the flat
    exact sampler's - bit-identical to the exact mode's draws, whatever else is in the batch and wherever the element
    sits in its lane's list - larger and non-integer counts the series', counts >= 200 the normal's; no observation: 0."""
    rs = np.random.RandomState(8)
    n = 50000
    b = rs.choice([0.0, 1.0, 2.0, 4.0, 7.0, 32.0, 33.0, 4.5, 150.0, 250.0], size=n)
    psi = rs.normal(size=n) * 2
    a = pg_batch(b, psi, seed=3)
    e = pg_batch(b, psi, seed=3, mode="exact")
    s = pg_batch(b, psi, seed=3, mode="series")
    small = (b >= 1) & (b <= 32) & (b == np.floor(b))
    assert np.array_equal(a[small], e[small])
    rest = (b > 32) | (b != np.floor(b))
    assert np.array_equal(a[rest & (b < 200)], s[rest & (b < 200)])
    assert np.array_equal(a[b >= 200], e[b >= 200])
    assert np.all(a[b == 0] == 0) and np.all(a[b > 0] > 0)
    # the stream of an element is keyed by (seed, index) alone: a shorter batch gives the same leading draws
    assert np.array_equal(pg_batch(b[:1234], psi[:1234], seed=3), a[:1234])
    assert not np.array_equal(pg_batch(b, psi, seed=4)[small], a[small])


@pytest.mark.parametrize("b,c", [(1, 0.0), (2, 3.0), (4, 1.0), (7, 0.5), (2.5, 2.0), (0.4, 1.0)])
def test_pg_exact_sampler_ks_against_definition(b, c):
    """The exact device sampler against the oracle's 256-term definition-based sampler."""
    from scipy.stats import ks_2samp
    from oracle import btf_oracle as orc
    n = 100000
    x = pg_batch(np.full(n, float(b)), np.full(n, c), seed=31, exact=True)
    y = orc.pg_draw_series(b, c, n, np.random.default_rng(5))
    assert ks_2samp(x, y).pvalue > 1e-3


@pytest.mark.parametrize("b", [1, 2, 4, 10, 1.3, 7.5, 13, 40, 33.7, 100.5, 300])
@pytest.mark.parametrize("c", [0.0, 0.5, 2.0, 8.0, 60.0])
def test_pg_moments(b, c):
    from oracle import btf_oracle as orc
    n = 400000
    x = pg_batch(np.full(n, float(b)), np.full(n, c), seed=1234 + int(10 * c) + int(b))
    m, v = float(orc.pg_mean(b, c)), float(orc.pg_var(b, c))
    assert np.all(x > 0)
    assert abs(x.mean() - m) < 5 * np.sqrt(v / n), (x.mean(), m)
    assert abs(x.var() - v) / v < 0.03, (x.var(), v)


@pytest.mark.parametrize("b,c", [(1, 0.0), (1, 3.0), (4, 1.0), (2.5, 2.0), (1.2, 0.0), (16, 0.7), (27.3, 4.0)])
def test_pg_distribution_ks(b, c):
    """Two-sample KS against the definition-based series sampler of the oracle."""
    from scipy.stats import ks_2samp
    from oracle import btf_oracle as orc
    x = pg_batch(np.full(20000, float(b)), np.full(20000, c), seed=77)
    y = orc.pg_draw_series(b, c, 20000, np.random.default_rng(3))
    assert ks_2samp(x, y).pvalue > 1e-3


@pytest.mark.parametrize("b,c", [(3, 0.0), (4, 1.0), (1.2, 0.0), (8, 0.3), (27.3, 4.0), (60, 12.0)])
def test_pg_series_sampler_ks_large_sample(b, c):
    """The sum-of-gammas path (every b but the integers 1, 2): 2-16 Gamma(b) terms from f32-transcendental
    variates + a moment-matched remainder.  200 k draws against the oracle's 256-term f64 series sampler:
    a two-sample KS at this size sees CDF differences of ~0.4 %."""
    from scipy.stats import ks_2samp
    from oracle import btf_oracle as orc
    n = 200000
    x = pg_batch(np.full(n, float(b)), np.full(n, c), seed=4242 + int(7 * b), mode="series")
    y = orc.pg_draw_series(b, c, n, np.random.default_rng(11))
    assert ks_2samp(x, y).pvalue > 1e-3
    # third central moment (the skewness the normal remainder could lose): within 5 % + noise
    m3x, m3y = ((x - x.mean()) ** 3).mean(), ((y - y.mean()) ** 3).mean()
    assert abs(m3x - m3y) < 0.05 * abs(m3y) + 8 * x.std() ** 3 / np.sqrt(n / 15), (m3x, m3y)


def test_pg_draw_fills_both_layouts_identically_and_masks_missing(golden):
    g = golden("g4_binomial_nan.npz")
    data = (g["Ysucc"], g["Ntrials"])
    model = binomial_model(g)
    model._resample_nu2(data)
    nu2 = model.nu2
    miss = np.isnan(g["Ysucc"])
    assert np.all(np.isinf(nu2[miss])) and np.all(np.isfinite(nu2[~miss])) and np.all(nu2[~miss] > 0)
    # mean of omega against E[PG(6, psi)]
    from oracle import btf_oracle as orc
    psi = np.einsum("nk,mtk->nmt", model.W, model.V)
    om = 1 / nu2[~miss]
    mref = orc.pg_mean(6.0, psi[~miss])
    vref = orc.pg_var(6.0, psi[~miss])
    zscore = (om - mref).sum() / np.sqrt(vref.sum())
    assert abs(zscore) < 5
    # the V-layout copy holds the same draws: a V step in exact mode must match the oracle given omega
    st = dict(W=model.W.copy(), V=model.V.copy(), Tau2=model.Tau2.copy(), lam2=float(model.lam2),
              sigma2=float(model.sigma2), nu2=nu2.copy())
    model2_V0 = model.V.copy()
    np.random.seed(9)
    z = np.random.normal(size=(model.ncols, model.nembeds * model.ndepth))
    np.random.seed(9)
    model.compat = "exact"
    model._resample_V(data)
    Delta = orc.trend_penalty(model.ndepth, 2)
    with np.errstate(invalid="ignore"):
        Vref = orc.binomial_v_step(st, g["Ysucc"], g["Ntrials"], Delta, z=z, compat="exact",
                                   perm=orc.perm_from_order(model.v_order(), model.nembeds, model.ndepth))
    assert relerr(model.V, Vref) < V_TOL


def test_binomial_chain_recovers_probabilities():
    """End-to-end statistical check of the PG path (device PG + weighted half-sweeps)."""
    from functionalmf_amd.factor import BinomialBayesianTensorFiltering
    rs = np.random.RandomState(3)
    N, M, T, K = 24, 10, 12, 2
    Wt = rs.normal(size=(N, K))
    Vt = 0.4 * np.cumsum(rs.normal(size=(M, T, K)), axis=1)
    P = 1 / (1 + np.exp(-np.einsum("nk,mtk->nmt", Wt, Vt)))
    Nt = np.full((N, M, T), 30.0)
    Ys = rs.binomial(30, P).astype(float)
    np.random.seed(4)
    model = BinomialBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=1, sigma2_init=1.0, lam2_init=0.1,
                                            compat="exact")
    res = model.run_gibbs((Ys, Nt), nburn=150, nthin=2, nsamples=100, verbose=False)
    Mu = np.einsum("znk,zmtk->znmt", res["W"], res["V"])
    Phat = (1 / (1 + np.exp(-Mu))).mean(0)
    assert np.corrcoef(Phat.reshape(-1), P.reshape(-1))[0, 1] > 0.97
    assert np.abs(Phat - P).mean() < 0.06


def test_device_buffers_are_visible_to_torch_distributed(tmp_path):
    """A caller that prefers its own collectives wraps the ctx's W / V device buffers (btf_dev_W / btf_dev_V) as torch
    tensors and gathers in place (parallel.Exchange.device_views; the library's own exchange is btf_allgather_W / _V).
    On a 1-GPU box: 1-rank nccl group, check aliasing both ways and the collective."""
    import os
    import torch
    import torch.distributed as dist
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        N, M, T, K = 12, 5, 8, 3
        np.random.seed(0)
        model = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=1, sigma2_init=1.0, lam2_init=0.1,
                                                nu2_init=1.0, shard=(0, 1),
                                                stream=torch.cuda.current_stream().cuda_stream)
        W0 = model.W.copy()
        model._push_state()
        Wt, Vt = model._exchange.device_views()
        assert Wt.is_cuda and Wt.dtype == torch.float64
        assert np.array_equal(Wt[:N * K].cpu().numpy().reshape(N, K), W0)          # same memory, read
        Wt[:N * K] += 1.0                                                            # ... and write
        torch.cuda.synchronize()
        model._W_dev_new = True
        assert np.array_equal(model.W, W0 + 1.0)
        n = model._plan.row_chunk * K
        dist.all_gather_into_tensor(Wt[:n], Wt[0:n])                                 # the in-place form btf_allgather_W issues
        torch.cuda.synchronize()
        model._W_dev_new = True
        assert np.array_equal(model.W, W0 + 1.0)
        s, c = model._exchange.sum_scalars(1.5, 2.0)
        assert (s, c) == (1.5, 2.0)
    finally:
        if created:
            dist.destroy_process_group()


def test_fused_gram_path_complete_data(golden):
    """Complete data: the K x K Gram of the fixed factor is produced by the preceding solve
    kernel (W'W by w_solve, V'V by the banded sampler) instead of a separate launch.  Run
    W -> V -> W back to back and check each against the reference / oracle."""
    from oracle import btf_oracle as orc
    g = golden("g2_c2_complete.npz")
    model, st = gaussian_model(g, "s0_")
    np.random.seed(300)
    model._resample_W(g["Y"])                 # V'V from the stand-alone kernel (first call)
    np.random.seed(301)
    model._resample_V(g["Y"])                 # W'W fused into w_solve
    assert relerr(model._W if not model._W_dev_new else model.W, g["W_after"]) < W_TOL
    assert relerr(model.V, g["V_after_twist"]) < V_TOL
    ost = dict(st, W=g["W_after"].copy(), V=model.V.copy())
    model._V_host_new = False                 # keep the device copy (and its fused V'V) authoritative
    np.random.seed(302)
    z = np.random.normal(size=g["z_W"].size)
    np.random.seed(302)
    model._resample_W(g["Y"])                 # V'V fused into the banded sampler
    orc.w_step(ost, g["Y"], z=z)
    assert relerr(model.W, ost["W"]) < W_TOL


def test_device_tau2_update_has_the_right_conditionals():
    """rng='device': the horseshoe+ chain is drawn on the GPU (Philox).  Check every level
    against its analytic conditional mean given the values it was drawn from."""
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    rs = np.random.RandomState(0)
    N, M, T, K = 8, 400, 40, 3
    V = 0.3 * np.cumsum(rs.normal(size=(M, T, K)), axis=1)
    np.random.seed(1)
    model = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, sigma2_init=1.0, lam2_init=0.4,
                                            nu2_init=1.0, V_init=V, rng="device", device_seed=3, compat="exact")
    a0, b0, c0 = model.Tau2_a.copy(), model.Tau2_b.copy(), model.Tau2_c.copy()
    lam2_a0 = float(np.ravel(model.lam2_a)[0])
    lo, hi = 1e-6, 1e6
    D = model.Delta.toarray()
    dsq = (np.einsum("rt,mtk->mrk", D, V) ** 2).sum(-1)
    model._resample_Tau2()
    tau, a, b, c = model.Tau2.copy(), model.Tau2_a.copy(), model.Tau2_b.copy(), model.Tau2_c.copy()
    n = tau.size

    def zscore(x, mean, var):
        return (x - mean).sum() / np.sqrt(var.sum())
    shape = (K + 1) / 2
    rate = np.clip(dsq / (2 * 0.4) + 1 / np.clip(c0, lo, hi), lo, hi)
    assert abs(zscore(1 / tau * rate, shape, np.full(n, shape))) < 5           # 1/tau ~ Gamma(shape, 1/rate)
    x = np.clip(1 / tau + 1 / b0, lo, hi)
    assert abs(zscore(x / c, 1.0, np.ones(n))) < 5                              # x/c ~ Exp(1)
    x = np.clip(1 / c + 1 / a0, lo, hi)
    assert abs(zscore(x / b, 1.0, np.ones(n))) < 5
    x = np.clip(1 / b + 1, lo, hi)
    assert abs(zscore(x / a, 1.0, np.ones(n))) < 5
    # lam2 | rest from the per-column sums the Tau2 kernel left on the device (compat="exact":
    # rate = 1/lam2_a + sum_j sum_r dsq/Tau2 / 2; shape/2 = 71 k, so ONE draw pins the rate to ~0.4 %)
    model._resample_lam2()
    shape = D.shape[0] * M * K + 1
    rate = 1 / lam2_a0 + (dsq / tau).sum() / 2
    assert abs(rate / model.lam2 / (shape / 2) - 1) < 0.02
    assert model.lam2_a > 0
    # the V step that follows must see the new Tau2 (prior band refreshed)
    Y = rs.normal(size=(N, M, T))
    model._resample_V(Y)
    model.sync()
    assert np.all(np.isfinite(model.V))


def test_device_mode_chain_matches_host_mode_statistically():
    """End to end in rng='device' mode (device normals + device horseshoe+ chain) against
    rng='host' on the same problem.  lam2 and sigma2 are held fixed: with both free the model's
    own scale feedback (lam2 collapsing to its floor, SURVEY hard part 2) makes single chains too
    erratic to compare.  Bounds are ~3x the chain-to-chain spread seen over seeds."""
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    rs = np.random.RandomState(5)
    N, M, T, R, K = 30, 12, 16, 3, 3
    Wt = rs.normal(size=(N, K))
    Vt = 0.4 * np.cumsum(rs.normal(size=(M, T, K)), axis=1)
    Mu = np.einsum("nk,mtk->nmt", Wt, Vt)
    Y = Mu[..., None] + rs.normal(0, 0.5, size=(N, M, T, R))
    out = {}
    for mode in ("device", "host"):
        np.random.seed(6)
        model = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, sigma2_true=1.0, lam2_true=0.05,
                                                nu2_init=1.0, rng=mode, compat="exact", device_seed=6)
        res = model.run_gibbs(Y, nburn=300, nthin=2, nsamples=200, verbose=False)
        Mu_hat = np.einsum("znk,zmtk->znmt", res["W"], res["V"]).mean(0)
        out[mode] = (res["nu2"].mean(), np.log(res["Tau2"]).mean(), np.sqrt(((Mu_hat - Mu) ** 2).mean()),
                     np.corrcoef(Mu_hat.reshape(-1), Mu.reshape(-1))[0, 1])
        assert out[mode][3] > 0.985, (mode, out[mode])
        assert 0.24 < out[mode][0] < 0.36, (mode, out[mode])
    d, h = out["device"], out["host"]
    assert abs(d[0] - h[0]) < 0.05 and abs(d[1] - h[1]) < 1.2 and abs(d[2] - h[2]) / h[2] < 0.3, out


def test_examples_run_end_to_end():
    """Config C1 plumbing (BASELINE.json configs[0]) and the Binomial example, as a user runs them."""
    import importlib.util
    import os
    from conftest import ROOT
    outs = {}
    for name in ("gaussian_tensor_filtering", "binomial_tensor_filtering", "negbinom_tensor_filtering"):
        spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "examples", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        outs[name] = mod.main(seed=1)
    # (default compat="reference" keeps the reference's lam2 collapse, quirk Q3, so the fits are as
    # smooth as the reference's own; the numbers below are what its sampler gives on this data)
    g = outs["gaussian_tensor_filtering"]
    assert g["rmse_observed"] < 0.9 and g["rmse_heldout"] < 1.2 and 0.25 < g["nu2"] < 1.0
    b = outs["binomial_tensor_filtering"]
    assert b["corr"] > 0.7 and b["mae_observed"] < 0.13
    nb = outs["negbinom_tensor_filtering"]
    assert nb["corr_observed"] > 0.7 and nb["rel_mae_observed"] < 0.4


def test_poisson_example_runs_end_to_end():
    """examples/poisson_tensor_filtering.py: the non-conjugate model with per-row / per-column slices."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("poisson_tensor_filtering", os.path.join(ROOT, "examples", "poisson_tensor_filtering.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rmse_in, rmse_out, cover = mod.main(seed=1, nburn=800, nsamples=300, nthin=2)
    # (short chain, and the default compat="reference" keeps the lam2 collapse of quirk Q3: the bands are too narrow,
    #  as the reference's own are)
    assert rmse_in < 0.35 and rmse_out < 0.8 and cover > 0.25


def test_bitwise_reproducible(golden):
    """No floating-point atomics, fixed-order reductions, counter-based RNG: two runs from the
    same state and seeds agree bit for bit (host- and device-RNG modes)."""
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    g = golden("g2_c2_complete.npz")
    N, M, T, R, K, tf = [int(x) for x in g["dims"]]
    runs = []
    for rep in range(2):
        st = state_from(g, "s0_")
        np.random.seed(3)
        model = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=tf, sigma2_init=st["sigma2"],
                                                lam2_init=st["lam2"], nu2_init=st["nu2"], W_init=st["W"],
                                                V_init=st["V"], Tau2_init=st["Tau2"], rng="device", device_seed=42)
        for _ in range(3):
            model._resample_W(g["Y"])
            model._resample_V(g["Y"])
        runs.append((model.W.copy(), model.V.copy()))
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])


def test_binomial_device_rng_mode_runs():
    """rng='device' with the Binomial model: device normals + device Tau2 chain + device PG."""
    from functionalmf_amd.factor import BinomialBayesianTensorFiltering
    rs = np.random.RandomState(2)
    N, M, T, K = 16, 6, 10, 2
    P = 1 / (1 + np.exp(-rs.normal(size=(N, M, T))))
    Nt = np.full((N, M, T), 8.0)
    Ys = rs.binomial(8, P).astype(float)
    np.random.seed(3)
    model = BinomialBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=1, sigma2_init=1.0, lam2_init=0.1,
                                            rng="device", compat="exact")
    res = model.run_gibbs((Ys, Nt), nburn=20, nthin=1, nsamples=10, verbose=False)
    assert np.all(np.isfinite(res["W"])) and np.all(np.isfinite(res["V"])) and res["nu2"].shape == (10, N, M, T)


@pytest.mark.timeout(400)
def test_two_ranks_share_one_gpu():
    """The sharded compute path (nonzero row / column offsets, local slabs, partial SSE) on real
    hardware: two processes on cuda:0, gloo process group, host-staged exchange."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29581", os.path.join(ROOT, "tests", "dist_gpu_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=380)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("SHARD_GPU_OK") == 2, out.stdout[-2000:]


@pytest.mark.timeout(400)
def test_three_ranks_overlapped_exchange_split_accumulation():
    """BTF_OPT_SPLIT_ACCUM on hardware with three ranks on cuda:0 (gloo control plane, exchange staged through the host):
    rank 0's own block opens the chunk range, rank 1's sits in the middle (a chunk map with a hole), rank 2's closes it.
    Every accumulation after the first runs as two launches and the half-sweeps equal the unsharded oracle; whole
    rng="device" sweeps equal the one-launch path bit for bit (tests/dist_gpu_worker.py: split_section)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT, BTF_DIST_SECTION="split")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3",
           "--master-addr", "127.0.0.1", "--master-port", "29585", os.path.join(ROOT, "tests", "dist_gpu_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=380)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("SHARD_GPU_OK") == 3, out.stdout[-2000:]


@pytest.mark.timeout(400)
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_reference_compat_reads_its_stale_weights_from_the_halo(world):
    """compat="reference" sharded: ranks on cuda:0 (gloo control plane) whose stale-weight source row / column lies in
    another rank's block carry it as one more row / column of their slabs (btf_set_shard_halo) and must reproduce the
    unsharded compat="reference" chains - Gaussian with missing curves, Binomial (Polya-Gamma weights of the halo redrawn
    every sweep), Binomial with missing cells, host-fed weights, Negative-Binomial (tests/dist_gpu_worker.py:
    reference_section)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT, BTF_DIST_SECTION="reference")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
           "--master-addr", "127.0.0.1", "--master-port", "29589", os.path.join(ROOT, "tests", "dist_gpu_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=380)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("SHARD_GPU_OK") == world, out.stdout[-2000:]


@pytest.mark.timeout(500)
@pytest.mark.parametrize("world,sections,overlap", [(2, "base", "0"), (3, "base,split,reference", "1")])
def test_peer_windows_between_processes_on_one_gpu(world, sections, overlap):
    """The N > 1 device exchange of the C ABI on hardware: `world` processes on cuda:0 (RCCL refuses that; gloo only
    carries the descriptors) with BTF_EXCHANGE_TRANSPORT=peer - btf_peer_export / btf_peer_init map every rank's W / V
    and mailbox into every other rank (hipIpc), and btf_allgather_W / _V / btf_allreduce_sse / _sum are one kernel each
    that stores this rank's block into the peers' buffers and waits for theirs (csrc/btf_comm.h).  The whole worker:
    host-RNG half-sweeps against the oracle, rng="device" chains (Gaussian complete / held-out, Binomial,
    Negative-Binomial) against the unsharded chains, the split accumulation with the gathers on the communication stream,
    compat="reference" with halo sources."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT, BTF_EXCHANGE_TRANSPORT="peer", HSA_ENABLE_IPC_MODE_LEGACY="0",
               BTF_DIST_SECTION=sections, BTF_DIST_OVERLAP=overlap)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
           "--master-addr", "127.0.0.1", "--master-port", "29591", os.path.join(ROOT, "tests", "dist_gpu_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=460)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("SHARD_GPU_OK") == world, out.stdout[-2000:]
    assert out.stdout.count("peer windows") == world, out.stdout[-2000:]


@pytest.mark.timeout(400)
def test_rccl_exchange_with_one_rank_reproduces_the_plain_chain():
    """The DEVICE collective path of sharded runs (btf_allgather_W / btf_allgather_V on the context's own W / V
    buffers and btf_allreduce_sse, the 8-byte all-reduce of the residual sum of squares: the library's own RCCL calls on
    the context's communicator and stream, bootstrapped by functionalmf_amd/parallel.py over the process group; the nu2
    draw split around the all-reduce, btf_draw_scalars which | 8 then | 16) on real hardware: a one-rank RCCL ("nccl")
    group with BTF_EXERCISE_EXCHANGE=1 issues every collective of an N-rank run.  Host-RNG half-sweeps against the oracle, then whole rng="device" sweeps (Gaussian complete,
    Gaussian with held-out cells, Binomial): the chain with the exchanges in must equal the plain one - a collective
    or a draw kernel running out of order on the stream would change it."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    for overlap in ("0", "1"):      # collectives in line on the ctx's stream / on the communication stream (btf_comm_fork / _join)
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT, BTF_DIST_BACKEND="nccl", BTF_EXERCISE_EXCHANGE="1",
                   HSA_ENABLE_IPC_MODE_LEGACY="0", BTF_DIST_OVERLAP=overlap, BTF_DIST_SECTION="base")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
               "--master-addr", "127.0.0.1", "--master-port", "29583", os.path.join(ROOT, "tests", "dist_gpu_worker.py")]
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=180)
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
        assert out.stdout.count("SHARD_GPU_OK") == 1, out.stdout[-2000:]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])
def test_rccl_ranks_on_their_own_gpus(world):
    """A real N-rank RCCL run, one GPU per rank (skipped on a one-GPU box): tests/dist_gpu_worker.py with the "nccl" backend
    and BTF_DIST_GPU_PER_RANK=1 - host-RNG half-sweeps against the oracle, then whole rng="device" chains (Gaussian complete,
    Gaussian with held-out cells, Binomial, Negative-Binomial) that must reproduce the unsharded chains, and the split
    accumulation with the overlapped exchange; every rank reports the backend and the world size it ran in."""
    import os
    import subprocess
    import sys
    import torch
    from conftest import ROOT
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs (this box has %d)" % (world, torch.cuda.device_count()))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT, BTF_DIST_BACKEND="nccl", BTF_DIST_GPU_PER_RANK="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0", BTF_DIST_SECTION="base,split,reference")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
           "--master-addr", "127.0.0.1", "--master-port", "29587", os.path.join(ROOT, "tests", "dist_gpu_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=560)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("SHARD_GPU_OK") == world, out.stdout[-2000:]
    assert out.stdout.count("backend nccl world %d" % world) == world, out.stdout[-2000:]


def test_device_scalar_draws_have_the_right_conditionals(golden):
    """rng='device': nu2, sigma2 (btf_draw_scalars) and lam2, lam2_a (btf_draw_lam2) are drawn on
    the GPU.  Repeat each draw from a fixed state and compare with the analytic conditionals
    (factor.py:411-416 / :130-132 / :143-153); the statistics they reduce (SSE, sum W^2) must
    equal the oracle's to rounding."""
    import ctypes
    from oracle import btf_oracle as orc
    from functionalmf_amd import _native
    g = golden("g1_c1_heldout.npz")
    # (sampler="banded": the host-scalar twin below uses it, and held-out curves now qualify for the spectral one)
    model, st = gaussian_model(g, "s0_", rng="device", device_seed=11, sampler="banded")
    Y = g["Y"]
    sse, nobs = orc.sse_and_count(st, Y)
    wfree = model._pack_W(st["W"])
    n = 3000
    nu2s, sig2s = np.zeros(n), np.zeros(n)
    for i in range(n):
        model._resample_nu2(Y)
        model._resample_sigma2()
        nu2s[i], sig2s[i] = model.nu2, model.sigma2
    out = np.zeros(6)
    model._ctx.call("btf_get_scalars", _native.dptr(out))
    assert abs(out[4] - sse) < 1e-9 * sse and abs(out[5] - wfree @ wfree) < 1e-12 * (wfree @ wfree)

    def check_gamma(x, shape):         # x ~ Gamma(shape, 1): mean and variance both = shape
        z = (x.mean() - shape) / np.sqrt(shape / x.size)
        assert abs(z) < 5, (x.mean(), shape, z)
        assert abs(x.var() / shape - 1) < 0.15, (x.var(), shape)
    check_gamma((0.1 + sse / 2) / nu2s, 0.1 + nobs / 2)
    check_gamma((0.1 + wfree @ wfree / 2) / sig2s, 0.1 + wfree.size / 2)
    # the half-sweeps must consume the device values: W step with nu2/sigma2 set through the device
    # path equals the W step of a host-scalar model with the same numbers and normals
    model.nu2, model.sigma2 = 0.7, 1.3
    model.rng = "host"                 # supply the normals; the scalars stay device-resident
    np.random.seed(5)
    model._resample_W(Y)
    ref, _ = gaussian_model(g, "s0_")
    ref.nu2, ref.sigma2 = 0.7, 1.3
    np.random.seed(5)
    ref._resample_W(Y)
    assert relerr(model.W, ref.W) < 1e-12
    np.random.seed(6)
    model._resample_V(Y)
    np.random.seed(6)
    ref._resample_V(Y)
    assert relerr(model.V, ref.V) < 1e-9


def test_full_device_sweep_needs_no_host_value_and_matches_host_scalars():
    """A chain whose scalars never visit the host between sweeps (rng='device') must sample the
    same posterior as the same chain with host-drawn scalars: compare nu2 and sigma2 posterior
    means (compat='exact'; lam2 fixed - see test_device_mode_chain_matches_host_mode_statistically)."""
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    rs = np.random.RandomState(7)
    N, M, T, R, K = 40, 16, 16, 3, 3
    Wt = rs.normal(size=(N, K))
    Vt = 0.4 * np.cumsum(rs.normal(size=(M, T, K)), axis=1)
    Y = np.einsum("nk,mtk->nmt", Wt, Vt)[..., None] + rs.normal(0, 0.5, size=(N, M, T, R))
    out = {}
    for mode in ("device", "host"):
        np.random.seed(8)
        model = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, lam2_true=0.05, sigma2_init=1.0,
                                                nu2_init=1.0, rng=mode, compat="exact", device_seed=9)
        assert model._dev_scalars == (mode == "device")
        res = model.run_gibbs(Y, nburn=1500, nthin=1, nsamples=1500, verbose=False)
        out[mode] = (res["nu2"].mean(), np.log(res["sigma2"]).mean(), res["nu2"].std())
        assert np.isfinite(res["W"]).all() and np.isfinite(res["V"]).all()
    # (sigma2 itself keeps drifting upwards in this model - W's scale is only weakly identified, the
    # reference does the same - so it is compared loosely; nu2 is stationary after ~1000 sweeps:
    # 0.2499 / 0.2488 in every mode over seeds, scripts/cmp_scalars.py)
    d, h = out["device"], out["host"]
    assert abs(d[0] - h[0]) < 0.004, out
    assert 0.24 < d[0] < 0.26, out
    assert abs(d[1] - h[1]) < 1.5, out


# ---- Negative-Binomial rate update ------------------------------------
NB_TAGS = ["scalar", "rows", "cells", "cols_depth"]


def negbinom_model(g, **kw):
    from functionalmf_amd.factor import NegativeBinomialBayesianTensorFiltering
    N, M, T, R, K, tf = [int(x) for x in g["dims"]]
    st = state_from(g, "s0_")
    model = NegativeBinomialBayesianTensorFiltering(
        N, M, T, nembeds=K, tf_order=tf, sigma2_init=st["sigma2"], lam2_init=st["lam2"], W_init=st["W"],
        V_init=st["V"], Tau2_init=st["Tau2"], R_init=g["R_before"].copy(),
        rdims=tuple(int(d) for d in g["rdims"]), **kw)
    for k in ("Tau2_a", "Tau2_b", "Tau2_c"):
        setattr(model, k, st[k].copy())
    model.lam2_a = st["lam2_a"]
    return model, st


@pytest.mark.parametrize("hist", [True, False])
@pytest.mark.parametrize("tag", NB_TAGS)
def test_negbinom_loglik_ratio_vs_oracle(golden, tag, hist):
    """The data-sized part of one MH step (btf_nb_loglik) against the oracle's gammaln form, for
    every sharing pattern of R; counts include NaNs, zeros and values beyond the product fast path."""
    from oracle import btf_oracle as orc
    from functionalmf_amd import _native
    g = golden("g7_negbinom_%s.npz" % tag)
    model, st = negbinom_model(g)
    if not hist:                       # full-tensor kernel even where the count histograms apply
        model._ctx.call("btf_set_option", _native.OPT_NB_HISTOGRAMS, 0)
    model._bind_data(g["data"])
    model._push_state()
    rs = np.random.RandomState(3)
    rdims = tuple(int(d) for d in g["rdims"])
    R = 1 + rs.gamma(2.0, 1.5, size=g["R_before"].shape)
    cand = R * np.exp(rs.normal(0, 0.3, size=R.shape))
    ll = np.zeros(R.shape)
    model._ctx.call("btf_nb_loglik", _native.dptr(R), _native.dptr(cand),
                    model._shared_flags().ctypes.data_as(_native._c_ip), _native.dptr(ll))
    ref = orc.nb_loglik_ratio(g["data"], R, cand, orc.nb_log1m_p(st["W"], st["V"]), rdims)
    assert relerr(ll.reshape(ref.shape), ref) < 1e-11


@pytest.mark.parametrize("tag", NB_TAGS)
def test_negbinom_rate_update_vs_reference(golden, tag):
    """30 random-walk MH steps from the reference's seed: same accept/reject path, same R, same
    Binomial trial counts (factor.py:513-554)."""
    g = golden("g7_negbinom_%s.npz" % tag)
    model, _ = negbinom_model(g)
    np.random.seed(int(g["seed_R"]))
    model._resample_R(g["data"])
    assert relerr(model.R, g["R_after"]) < 1e-10
    assert relerr(model.N, g["N_after"]) < 1e-10


@pytest.mark.parametrize("tag", ["scalar", "rows"])
def test_negbinom_full_sweep_vs_reference(golden, tag):
    """One whole sweep of the reference's NegativeBinomial model (R, then the Binomial sweep on the
    rebuilt pseudo-data) given the injected Polya-Gamma draws."""
    g = golden("g7_negbinom_%s.npz" % tag)
    model, _ = negbinom_model(g)
    model.sample_nu2 = False
    with np.errstate(divide="ignore"):
        model.nu2 = 1 / g["omega"]
    np.random.seed(int(g["seed_full"]))
    model.resample(g["data"])
    assert relerr(model.R, g["full_R"]) < 1e-10 and relerr(model.N, g["full_N"]) < 1e-10
    for k, tol in (("sigma2", 1e-10), ("lam2", 1e-10), ("Tau2", 1e-9), ("W", 1e-9), ("V", V_TOL)):
        assert relerr(getattr(model, k), g["full_" + k]) < tol, k


@pytest.mark.parametrize("rng", ["host", "device"])
def test_negbinom_chain_recovers_rate_and_mean(rng):
    """End to end (device PG draws): counts simulated from NB(R=5, p) - the chain must put R near 5
    and the fitted mean R p/(1-p) near the truth.  rng="device": the MH loop itself runs on the GPU."""
    from functionalmf_amd.factor import NegativeBinomialBayesianTensorFiltering
    rs = np.random.RandomState(12)
    N, M, T, Rr, K = 24, 10, 12, 4, 2
    Wt = 0.7 * rs.normal(size=(N, K))
    Vt = 0.3 * np.cumsum(rs.normal(size=(M, T, K)), axis=1)
    P = 1 / (1 + np.exp(-np.einsum("nk,mtk->nmt", Wt, Vt)))
    data = rs.negative_binomial(5.0, 1 - P[..., None].repeat(Rr, -1)).astype(float)
    data[:2, :2] = np.nan
    np.random.seed(13)
    model = NegativeBinomialBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, sigma2_init=0.5, lam2_init=0.1,
                                                    nmetropolis=10, rng=rng, device_seed=3)
    res = model.run_gibbs(data, nburn=300, nthin=2, nsamples=150, verbose=False)
    assert res["R"].shape == (150, 1, 1, 1)
    assert getattr(model, "_mh_on_device", True)
    R_hat = res["R"].mean()
    assert 3.5 < R_hat < 7.0, R_hat
    Ps = 1 / (1 + np.exp(-np.einsum("znk,zmtk->znmt", res["W"], res["V"]).clip(-10, 10)))
    Mu_hat = (res["R"] * Ps / (1 - Ps)).mean(0)
    Mu = 5.0 * P / (1 - P)
    obs = ~np.isnan(data[..., 0])
    assert np.corrcoef(Mu_hat[obs], Mu[obs])[0, 1] > 0.9


@pytest.mark.parametrize("tag", ["scalar", "rows"])
def test_negbinom_histogram_path_with_outliers(golden, tag):
    """Counts beyond the LDS table (>= 1024) and fractional pseudo-counts ride along the histogram
    form as per-row outlier lists, summed in data order."""
    from oracle import btf_oracle as orc
    from functionalmf_amd import _native
    g = golden("g7_negbinom_%s.npz" % tag)
    data = g["data"].copy()
    data[3, 2, 1, 0], data[3, 2, 1, 1], data[3, 4, 0, 2] = 2500.0, 1024.0, 1023.0
    data[6, 5, 7, 2], data[6, 0, 0, 0], data[2, 3, 3, 1] = 3.5, 40000.0, 0.25
    model, st = negbinom_model(g)
    model._bind_data(data)
    model._push_state()
    rs = np.random.RandomState(4)
    rdims = tuple(int(d) for d in g["rdims"])
    ll = np.zeros(g["R_before"].shape)
    flags = model._shared_flags().ctypes.data_as(_native._c_ip)
    for _ in range(2):          # second call: cached sum cnt*log(1-p)
        R = 1 + rs.gamma(2.0, 1.5, size=ll.shape)
        cand = R * np.exp(rs.normal(0, 0.3, size=R.shape))
        model._ctx.call("btf_nb_loglik", _native.dptr(R), _native.dptr(cand), flags, _native.dptr(ll))
        ref = orc.nb_loglik_ratio(data, R, cand, orc.nb_log1m_p(st["W"], st["V"]), rdims)
        assert relerr(ll.reshape(ref.shape), ref) < 1e-11
    # after W moves the cached row sums must be rebuilt
    model.W = st["W"] * 0.9
    model._push_state()
    model._ctx.call("btf_nb_loglik", _native.dptr(R), _native.dptr(cand), flags, _native.dptr(ll))
    ref = orc.nb_loglik_ratio(data, R, cand, orc.nb_log1m_p(st["W"] * 0.9, st["V"]), rdims)
    assert relerr(ll.reshape(ref.shape), ref) < 1e-11


@pytest.mark.parametrize("rdims", [(0, 1, 2), (1, 2)])
def test_negbinom_device_mh_loop_samples_the_same_posterior(rdims):
    """R | W, V sampled by the host-driven loop (legacy RNG, reference path) and by the device loop
    (btf_nb_mh, Philox) with W and V held fixed: same posterior mean and spread of log R."""
    from functionalmf_amd.factor import NegativeBinomialBayesianTensorFiltering
    rs = np.random.RandomState(21)
    N, M, T, Rr, K = 6, 8, 10, 2, 2
    Wt = 0.7 * rs.normal(size=(N, K))
    Vt = 0.3 * np.cumsum(rs.normal(size=(M, T, K)), axis=1)
    P = 1 / (1 + np.exp(-np.einsum("nk,mtk->nmt", Wt, Vt)))
    data = rs.negative_binomial(3.0, 1 - P[..., None].repeat(Rr, -1)).astype(float)
    data[0, 0, 0, 0] = 1500.0               # an outlier beyond the count table
    out = {}
    for rng in ("host", "device"):
        np.random.seed(22)
        model = NegativeBinomialBayesianTensorFiltering(N, M, T, nembeds=K, W_true=Wt, V_true=Vt, sigma2_true=1.0,
                                                        lam2_true=0.1, Tau2_true=np.ones((M, 3 * T - 1)),
                                                        rdims=rdims, nmetropolis=5, rng=rng, device_seed=5)
        model._bind_data(data)
        draws = []
        for sweep in range(1500):
            model._resample_R(data)
            if sweep >= 300:
                draws.append(np.log(np.array(model.R)).reshape(-1).copy())
        d = np.array(draws)
        out[rng] = (d.mean(0), d.std(0))
        if rng == "device":
            assert getattr(model, "_mh_on_device", True) and model._R_dev_new is not None
    (mh, sh), (md, sd) = out["host"], out["device"]
    assert np.all(np.abs(mh - md) < 0.25 * np.maximum(sh, sd) + 0.02), (mh, md, sh, sd)
    assert np.all(np.abs(sh - sd) < 0.35 * np.maximum(sh, sd) + 0.01), (sh, sd)


@pytest.mark.parametrize("noutliers", [1, 300])
def test_negbinom_single_rate_mh_loop_in_one_launch_equals_the_stepwise_loop(monkeypatch, noutliers):
    """rdims = (0,1,2) (one rate, the reference's default): btf_nb_mh runs its whole loop in one launch from the
    histogram of all counts (or, with many counts beyond the table, one partial-sum launch per step);
    BTF_NB_MH_STEPWISE=1 keeps the per-row launches every other sharing pattern uses.
    Same Philox streams -> the same chain of rates (the sums are formed in a different order: rounding only)."""
    from functionalmf_amd.factor import NegativeBinomialBayesianTensorFiltering
    rs = np.random.RandomState(33)
    N, M, T, Rr, K = 12, 10, 12, 3, 2
    Wt = 0.7 * rs.normal(size=(N, K))
    Vt = 0.3 * np.cumsum(rs.normal(size=(M, T, K)), axis=1)
    P = 1 / (1 + np.exp(-np.einsum("nk,mtk->nmt", Wt, Vt)))
    data = rs.negative_binomial(4.0, 1 - P[..., None].repeat(Rr, -1)).astype(float)
    # counts beyond the 1024-entry table: one -> the one-launch loop; 300 (> 256) -> one partial-sum launch per step
    flat = data.reshape(-1)
    flat[rs.choice(flat.size, noutliers, replace=False)] += 1500.0 + rs.randint(0, 700, size=noutliers)
    data[0, :2] = np.nan
    chains = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("BTF_NB_MH_STEPWISE", mode)
        np.random.seed(4)
        model = NegativeBinomialBayesianTensorFiltering(N, M, T, nembeds=K, W_true=Wt, V_true=Vt, sigma2_true=1.0,
                                                        lam2_true=0.1, Tau2_true=np.ones((M, 3 * T - 1)),
                                                        rdims=(0, 1, 2), nmetropolis=7, rng="device", device_seed=11)
        model._bind_data(data)
        out = []
        for _ in range(25):
            model._resample_R(data)
            out.append(float(np.asarray(model.R).reshape(-1)[0]))
        assert getattr(model, "_mh_on_device", True)
        chains[mode] = np.array(out)
    assert len(set(np.round(chains["0"], 9))) > 5          # the chain moves
    np.testing.assert_allclose(chains["0"], chains["1"], rtol=1e-9)


def test_run_gibbs_device_collection_equals_per_sample_copies():
    """rng='device': run_gibbs collects the kept states on the GPU (btf_collect*).  Same seeds -> the same
    Philox streams, so the result dict must equal, bit for bit, what per-sample host copies
    (genlasso.py:51-65, forced through a no-op callback) return; and the summary computed from the
    device-resident samples must equal numpy on the downloaded ones."""
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    rs = np.random.RandomState(31)
    N, M, T, R, K = 14, 9, 10, 2, 3
    Y = np.einsum("nk,mtk->nmt", rs.normal(size=(N, K)), 0.3 * np.cumsum(rs.normal(size=(M, T, K)), axis=1))[..., None] \
        + rs.normal(0, 0.5, size=(N, M, T, R))
    Y[:2, :2] = np.nan
    res = {}
    for how in ("device", "host"):
        np.random.seed(32)
        model = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, sigma2_init=0.5, lam2_init=0.1,
                                                nu2_init=1.0, rng="device", device_seed=33)
        cb = None if how == "device" else (lambda m, d, step: None)
        res[how] = model.run_gibbs(Y, nburn=7, nthin=3, nsamples=12, verbose=False, callback=cb)
        if how == "device":
            mean, quant = model.posterior_summary(q=(5, 95))
    assert sorted(res["device"]) == sorted(res["host"])
    for k in res["host"]:
        assert res["device"][k].shape == res["host"][k].shape, k
        assert np.array_equal(res["device"][k], res["host"][k]), k
    Mu = np.einsum("znk,zmtk->znmt", res["device"]["W"], res["device"]["V"])
    assert np.max(np.abs(mean - Mu.mean(0))) < 1e-12 * np.abs(Mu).max()
    assert np.max(np.abs(quant - np.percentile(Mu, (5, 95), axis=0))) < 1e-12 * np.abs(Mu).max()


def test_negbinom_fixed_rate_and_edge_arguments():
    """R_true: the reference never defines self.N and fails in resample (factor.py:476-478 vs :508); here the
    Binomial pseudo-data is built from the fixed rate.  nmetropolis=0 leaves R alone; a (Y, N) tuple is refused."""
    from functionalmf_amd.factor import NegativeBinomialBayesianTensorFiltering
    rs = np.random.RandomState(41)
    N, M, T, K = 8, 5, 9, 2
    data = rs.poisson(3.0, size=(N, M, T)).astype(float)          # 3-D input: one replicate implied
    data[0, 0] = np.nan
    np.random.seed(42)
    m = NegativeBinomialBayesianTensorFiltering(N, M, T, nembeds=K, R_true=np.full((1, 1, 1), 4.0))
    assert m.sample_R is False
    m.resample(data)
    m.sync()
    Ntr = m.N
    assert np.allclose(Ntr[1:], data[1:] + 4.0) and np.all(Ntr[0, 0] == 0.0)
    assert np.isfinite(m.W).all() and np.isfinite(m.V).all()
    m2 = NegativeBinomialBayesianTensorFiltering(N, M, T, nembeds=K, nmetropolis=0, rdims=(1, 2))
    R0 = np.array(m2.R).copy()
    m2._resample_R(data)
    assert np.array_equal(np.array(m2.R), R0) and m2.R.shape == (N, 1, 1)
    with pytest.raises(ValueError):
        m2.resample((data, data))


def test_posterior_summary_rejects_bad_input_and_model_without_samples():
    from functionalmf_amd.utils import posterior_summary
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    from functionalmf_amd._native import BTFError
    Ws, Vs = np.zeros((4, 3, 2)), np.zeros((4, 5, 6, 2))
    with pytest.raises(ValueError):
        posterior_summary(Ws, Vs[:3])
    with pytest.raises(BTFError):
        posterior_summary(Ws, Vs, q=(101,))
    with pytest.raises(KeyError):
        posterior_summary(Ws, Vs, transform="log")
    np.random.seed(0)
    m = GaussianBayesianTensorFiltering(3, 5, 6, nembeds=2, nu2_init=1.0, rng="device")
    with pytest.raises(RuntimeError):
        m.posterior_summary()


@pytest.mark.parametrize("name,compat", [("g2_c2_complete.npz", "reference"), ("g1_c1_heldout.npz", "exact"),
                                         ("g3_partial_reps.npz", "exact"), ("g1_c1_heldout.npz", "reference")])
def test_in_sweep_sse_comes_from_the_w_partials(golden, name, compat):
    """Inside a device-mode sweep nu2 | rest takes its residual sum of squares from the W half-sweep's
    accumulation partials (btf_w_accum + btf_draw_scalars(which|4)); it must equal the direct reduction
    (the stale-weight mode of compat='reference' with missing data falls back to the direct kernel), and the
    W step that follows must be the same as without the early accumulation."""
    from oracle import btf_oracle as orc
    from functionalmf_amd import _native
    g = golden(name)
    Y = g["Y"]
    model, st = gaussian_model(g, "s0_", rng="device", device_seed=17, compat=compat)
    sse, nobs = orc.sse_and_count(st, Y)
    model._in_sweep = True
    model._resample_nu2(Y)
    model._in_sweep = False
    out = np.zeros(6)
    model._ctx.call("btf_get_scalars", _native.dptr(out))
    assert abs(out[4] - sse) < 1e-9 * sse, (out[4], sse)
    # the W step after the early accumulation vs a fresh model that never called btf_w_accum
    model.nu2, model.sigma2 = 0.7, 1.3
    model.rng = "host"
    np.random.seed(5)
    model._resample_W(Y)
    ref, _ = gaussian_model(g, "s0_", compat=compat)
    ref.nu2, ref.sigma2 = 0.7, 1.3
    np.random.seed(5)
    ref._resample_W(Y)
    assert relerr(model.W, ref.W) < 1e-12


# ---- spectral V sampler (complete data; default with rng="device") -------------------------------------
def _spectral_model(golden, tag, **kw):
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    from test_oracle_golden import _spectral_case
    Y, st, (N, M, T, R, K, tf), z, Vref = _spectral_case(golden, tag)
    model = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=tf, sigma2_init=st["sigma2"], lam2_init=st["lam2"],
                                            nu2_init=st["nu2"], W_init=st["W"], V_init=st["V"], Tau2_init=st["Tau2"],
                                            sampler="spectral", **kw)
    return model, Y, st, (N, M, T, R, K, tf), z, Vref


@pytest.mark.parametrize("tag", ["g2", "k5", "tf0", "tf1", "tf3", "short", "held"])
def test_spectral_v_step_vs_reference(golden, tag, monkeypatch):
    """The spectral kernel against the fixture the reference's own _resample_V produced under the declared
    spectral square root (tests/golden/make_golden_spectral.py), from the same state and normals."""
    model, Y, st, dims, z, Vref = _spectral_model(golden, tag)
    monkeypatch.setattr(model, "_v_normals", lambda: z)
    model._resample_V(Y)
    assert model.v_sampler() == "spectral"
    # held: whole curves missing (the reference examples' pattern) - complete-data kernels plus corrections
    assert model.likelihood_form() == ("curve_counts" if tag == "held" else "complete")
    assert relerr(model.V, Vref) < V_TOL


@pytest.mark.parametrize("tag", ["g2", "k5", "tf0", "tf1", "tf3", "short", "held"])
@pytest.mark.parametrize("fused", [0, 1])
def test_spectral_mean_vs_dense_lapack_on_the_references_system(golden, tag, fused, monkeypatch):
    """The benchmarked square root pinned to dense LAPACK: with z = 0 the spectral kernel's draw is the conditional mean,
    and G8 holds np.linalg.solve(Q_j, mu_part_j) on the Q and mu_part the reference's own _resample_V assembled
    (fast_mvn.py:47; tests/golden/make_golden_spectral.py), with cond(Q_j).  Column by column to 50 cond(Q_j) eps; the
    stand-alone kernel and (where the shape admits it) the fused tail of the V accumulation launch."""
    from functionalmf_amd import _native
    g8 = golden("g8_spectral.npz")
    model, Y, st, (N, M, T, R, K, tf), z, _ = _spectral_model(golden, tag)
    model._ctx.call("btf_set_option", _native.OPT_FUSED_STEP, fused)
    monkeypatch.setattr(model, "_v_normals", lambda: 0 * z)
    model._resample_V(Y)
    assert model.v_sampler() == "spectral"
    dense, cond = g8[tag + "_V_mean_dense"], g8[tag + "_cond"]
    V = model.V
    for j in range(M):
        err = np.abs(V[j] - dense[j]).max() / np.abs(dense[j]).max()
        assert err <= 50 * cond[j] * np.finfo(float).eps, (tag, j, err, cond[j])


def test_spectral_jitter_retry_matches_reference_schedule(golden):
    import ctypes as C
    g5, g8 = golden("g5_illcond.npz"), golden("g8_spectral.npz")
    model, _ = gaussian_model(g5, "retry_s0_", sampler="spectral")
    np.random.seed(601)
    model._resample_V(g5["Y"])
    V = model.V.copy()
    tries = np.zeros(model.ncols, dtype=np.int32)
    model._ctx.call("btf_get_V_attempts", tries.ctypes.data_as(C.POINTER(C.c_int32)))
    assert np.array_equal(tries, g8["g5_retry_tries_spectral"])
    assert relerr(V, g8["g5_retry_V_after_spectral"]) < 1e-5


def test_spectral_square_root_covariance_identity(golden, monkeypatch):
    """Feed unit normals: the kernel's noise map S (column by column) must satisfy S S' Q = I for the
    precision the reference assembles (oracle v_step_system) - the draw has exactly the conditional
    covariance, whatever the normals."""
    from oracle import btf_oracle as orc
    model, Y, st, (N, M, T, R, K, tf), z, _ = _spectral_model(golden, "k5")
    n = K * T
    V0 = st["V"].copy()
    monkeypatch.setattr(model, "_v_normals", lambda: np.zeros((M, n)))
    model._resample_V(Y)
    mean = model.V.copy()
    S = np.zeros((M, n, n))                         # depth-major rows, pivot columns
    for i in range(n):
        e = np.zeros((M, n)); e[:, i] = 1.0
        monkeypatch.setattr(model, "_v_normals", lambda e=e: e)
        model.V = V0
        model._resample_V(Y)
        S[:, :, i] = (model.V - mean).reshape(M, n)
    Delta = orc.trend_penalty(T, tf)
    st["_cnt"], st["_ybar"] = orc.replicate_stats(Y)
    pm = orc.depth_major_perm(K, T)
    for j in range(M):
        Q, _ = orc.v_step_system(st, Y, Delta, j, j)
        Qd = Q[np.ix_(pm, pm)]
        assert np.abs(S[j] @ S[j].T @ Qd - np.eye(n)).max() < 1e-8


def test_device_rng_v_draw_is_white_under_the_conditional_precision(golden):
    """rng="device" (the benchmarked mode; spectral sampler on complete data): for every column,
    C' (x - Q^-1 mu) with C C' = Q must be standard normal.  Pooled over columns and repeats: mean, variance,
    fourth moment, lag correlations and a KS test of the whitened residuals."""
    from scipy import stats
    from oracle import btf_oracle as orc
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    for sampler in ("spectral", "banded"):
        _, Y, st, (N, M, T, R, K, tf), _, _ = _spectral_model(golden, "k5")
        model = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=tf, sigma2_init=st["sigma2"], lam2_init=st["lam2"],
                                                nu2_init=st["nu2"], W_init=st["W"], V_init=st["V"], Tau2_init=st["Tau2"],
                                                rng="device", sampler=sampler, device_seed=5)
        Delta = orc.trend_penalty(T, tf)
        n = K * T
        pm = orc.depth_major_perm(K, T)
        st["_cnt"], st["_ybar"] = orc.replicate_stats(Y)
        Cs, means = [], []
        for j in range(M):
            Q, mu = orc.v_step_system(st, Y, Delta, j, j)
            Qd = Q[np.ix_(pm, pm)]
            Cs.append(np.linalg.cholesky(Qd))
            means.append(np.linalg.solve(Qd, mu[pm]))
        reps = 400
        res = np.empty((reps, M, n))
        for r in range(reps):
            model._resample_V(Y)
            Vd = model.V.reshape(M, n)
            for j in range(M):
                res[r, j] = Cs[j].T @ (Vd[j] - means[j])
        assert model.v_sampler() == sampler
        x = res.reshape(-1)
        assert abs(x.mean()) < 5 / np.sqrt(x.size)
        assert abs(x.var() - 1) < 5 * np.sqrt(2 / x.size)
        assert abs((x ** 4).mean() - 3) < 5 * np.sqrt(96 / x.size)
        assert stats.kstest(x[::7], "norm").pvalue > 1e-3
        # no correlation between unknowns of a column, nor between columns, nor between consecutive draws
        c1 = np.mean(res[:, :, :-1] * res[:, :, 1:])
        c2 = np.mean(res[:, :-1, :] * res[:, 1:, :])
        c3 = np.mean(res[:-1] * res[1:])
        assert max(abs(c1), abs(c2), abs(c3)) < 5 / np.sqrt(x.size)
        # per-coordinate variances (a wrong factor entry shows up in a few coordinates only)
        v = res.var(axis=0)
        assert np.abs(v - 1).max() < 6 * np.sqrt(2 / reps)


@pytest.mark.parametrize("K", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10])
def test_sym_eig_matches_lapack(K):
    """btf_sym_eig (the one-wave Jacobi the spectral sampler runs beside the V accumulation) against numpy
    eigh under the declared conventions: ascending eigenvalues, largest-magnitude entry of each vector positive.
    Matrices: random Grams (well separated), a nearly diagonal one, and one with a tiny eigenvalue."""
    from functionalmf_amd import _native
    from oracle import btf_oracle as orc
    lib = _native.load()
    rs = np.random.RandomState(K)
    KK = K * (K + 1) // 2
    tril = np.tril_indices(K)
    mats = []
    X = rs.normal(size=(300, K)) * (1 + np.arange(K))
    parts = np.stack([(X[b::4].T @ X[b::4])[tril] for b in range(4)])       # four partial Grams
    mats.append(parts)
    D = np.diag(1.0 + np.arange(K)) + 1e-9 * rs.normal(size=(K, K))
    mats.append(((D + D.T) / 2)[tril][None, :])
    Y = rs.normal(size=(K + 3, K))
    Y[:, -1] *= 1e-7
    mats.append((Y.T @ Y)[tril][None, :])
    for parts in mats:
        parts = np.ascontiguousarray(parts)
        out = np.zeros(K + K * K + 1)
        rc = lib.btf_sym_eig(0, K, parts.shape[0], _native.dptr(parts), _native.dptr(out), None)
        assert rc == 0
        G = np.zeros((K, K))
        G[tril] = parts.sum(axis=0)
        G = G + np.tril(G, -1).T
        g, U = orc.gram_eigensystem(G)
        lam, Ud = out[:K], out[K:K + K * K].reshape(K, K)
        assert np.abs(lam - g).max() <= 1e-13 * np.abs(g).max()
        assert np.abs(Ud.T @ Ud - np.eye(K)).max() < 1e-14
        assert np.abs(Ud.T @ G @ Ud - np.diag(lam)).max() <= 1e-13 * np.abs(g).max()
        gap = np.min(np.diff(g)) / np.abs(g).max() if K > 1 else 1.0
        assert np.abs(Ud - U).max() < 1e-13 / max(gap, 1e-12)
        assert out[-1] <= 12
        # warm path: the eigen-system of a nearby matrix (what one Gibbs sweep does to W'W), refined
        if K > 1:
            E = 1e-3 * np.abs(G).max() * rs.normal(size=(K, K))
            G2 = G + (E + E.T) / 2
            p2 = np.ascontiguousarray(G2[tril][None, :])
            out2 = np.zeros(K + K * K + 1)
            assert lib.btf_sym_eig(0, K, 1, _native.dptr(p2), _native.dptr(out2), _native.dptr(out)) == 0
            g2, U2 = orc.gram_eigensystem(G2)
            lam2, Ud2 = out2[:K], out2[K:K + K * K].reshape(K, K)
            assert np.abs(lam2 - g2).max() <= 1e-13 * np.abs(g2).max()
            assert np.abs(Ud2.T @ Ud2 - np.eye(K)).max() < 1e-14
            assert np.abs(Ud2.T @ G2 @ Ud2 - np.diag(lam2)).max() <= 1e-13 * np.abs(g2).max()


# ---- elliptical slice sampling: NonconjugateBayesianTensorFiltering ---------------------------
def _nc_model(golden, link, **kw):
    from functionalmf_amd.factor import NonconjugateBayesianTensorFiltering
    from test_oracle_golden import _nc_case
    g, tag, st, (N, M, T, R, K, tf) = _nc_case(golden, link)
    model = NonconjugateBayesianTensorFiltering(
        N, M, T, "poisson_log" if link == "log" else "poisson_identity", nembeds=K, tf_order=tf, sigma2_init=st["sigma2"],
        lam2_init=st["lam2"], W_init=st["W"], V_init=st["V"], Tau2_init=st["Tau2"], **kw)
    return model, g, tag, st


@pytest.mark.parametrize("link", ["log", "identity"])
def test_nonconjugate_joint_slice_walks_the_reference_path(golden, link):
    """rng="host": prior draw, likelihood passes and proposals on the GPU, normals and uniforms from the legacy numpy
    stream - the chain must land where the reference's NonconjugateBayesianTensorFiltering landed (fixture g9, made
    by the reference itself), after the same number of likelihood evaluations."""
    from oracle import btf_oracle as orc
    model, g, tag, st = _nc_model(golden, link)
    Y = g[tag + "Y"]
    ll0 = model.log_likelihood(Y)
    assert abs(ll0 - orc.poisson_loglik(st["W"], st["V"], Y, link)) < 1e-9 * abs(ll0)
    np.random.seed(1100)
    model._resample_W(Y)
    assert model.ess_evaluations == int(g[tag + "W_nev"])
    assert relerr(model.W, g[tag + "W_after"]) < 1e-10
    np.random.seed(1200)
    model._resample_V(Y)
    assert model.ess_evaluations == int(g[tag + "V_nev"])
    assert relerr(model.V, g[tag + "V_after"]) < 1e-6


@pytest.mark.gpu
def test_nonconjugate_python_callback_walks_the_reference_path(golden):
    """`loglikelihood` as a Python function, the reference's own interface (factor.py:567-612; the callers pass functions:
    examples/poisson_tensor_filtering.py:60-70): prior draws and proposals on the GPU, every proposal read back and handed
    to the function.  rng="host": the chain lands where the reference landed with the same (arbitrary: Student-t around a
    tanh link) function - fixture g9 case (e), three W / V slices in a row, evaluation counts included."""
    from functionalmf_amd.factor import NonconjugateBayesianTensorFiltering
    from test_oracle_golden import _cb_case, student_t_tanh_loglik
    g, st, (N, M, T, R, K, tf) = _cb_case(golden)
    Y = g["cb_Y"]
    calls = [0]

    def counted(W, V, data):
        calls[0] += 1
        assert data is Y and W.shape == (N, K) and V.shape == (M, T, K)
        return student_t_tanh_loglik(W, V, data)
    model = NonconjugateBayesianTensorFiltering(N, M, T, counted, nembeds=K, tf_order=tf, sigma2_init=st["sigma2"], lam2_init=st["lam2"],
                                                W_init=st["W"], V_init=st["V"], Tau2_init=st["Tau2"])
    assert abs(model.log_likelihood(Y) - student_t_tanh_loglik(st["W"], st["V"], Y)) < 1e-9
    for sweep in range(3):
        calls[0] = 0
        np.random.seed(1300 + 2 * sweep)
        model._resample_W(Y)
        assert model.ess_evaluations == int(g["cb_nev"][2 * sweep]) == calls[0] - 1
        assert relerr(model.W, g["cb_W_chain"][sweep]) < 1e-10
        np.random.seed(1301 + 2 * sweep)
        model._resample_V(Y)
        assert model.ess_evaluations == int(g["cb_nev"][2 * sweep + 1])
        assert relerr(model.V, g["cb_V_chain"][sweep]) < 1e-6
    # run_gibbs with a callback and device normals: runs, stays finite, keeps the result layout
    model2 = NonconjugateBayesianTensorFiltering(N, M, T, student_t_tanh_loglik, nembeds=K, tf_order=tf, sigma2_init=st["sigma2"],
                                                 lam2_init=st["lam2"], W_init=st["W"], V_init=st["V"], rng="device", device_seed=3)
    np.random.seed(5)
    out = model2.run_gibbs(Y, nburn=2, nthin=1, nsamples=3, verbose=False)
    assert out["W"].shape == (3, N, K) and out["V"].shape == (3, M, T, K) and np.isfinite(out["V"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["bernoulli_logit", "gaussian", "negbin_logit"])
def test_nonconjugate_other_likelihoods_walk_the_reference_path(golden, family):
    """The device likelihoods beyond Poisson (include/btf.h, links 2..4): rng="host", the chain must land where the
    reference's NonconjugateBayesianTensorFiltering landed with the matching scipy.stats callback (fixture g11, made by
    the reference itself: tests/golden/make_golden_lik.py), after the same number of likelihood evaluations; then whole
    device-RNG slices (joint and per row / per column) leave a finite log-likelihood."""
    from oracle import btf_oracle as orc
    from functionalmf_amd.factor import NonconjugateBayesianTensorFiltering
    g = golden("g11_likelihoods.npz")
    tag = "lk_%s_" % family
    N, M, T, R, K, tf = [int(x) for x in g[tag + "dims"]]
    par = None if np.isnan(g[tag + "param"]) else float(g[tag + "param"])
    st = {k: (float(g[tag + "s0_" + k]) if k in ("lam2", "sigma2") else g[tag + "s0_" + k].copy()) for k in ("W", "V", "Tau2", "lam2", "sigma2")}
    Y = g[tag + "Y"]
    model = NonconjugateBayesianTensorFiltering(N, M, T, family, likelihood_param=par, nembeds=K, tf_order=tf, sigma2_init=st["sigma2"],
                                                lam2_init=st["lam2"], W_init=st["W"], V_init=st["V"], Tau2_init=st["Tau2"])
    ll0 = model.log_likelihood(Y)
    assert abs(ll0 - float(g[tag + "ll0"])) < 1e-9 * abs(ll0), (ll0, float(g[tag + "ll0"]))
    np.random.seed(int(g[tag + "seeds"][0]))
    model._resample_W(Y)
    assert model.ess_evaluations == int(g[tag + "W_nev"])
    assert relerr(model.W, g[tag + "W_after"]) < 1e-10
    np.random.seed(int(g[tag + "seeds"][1]))
    model._resample_V(Y)
    assert model.ess_evaluations == int(g[tag + "V_nev"])
    assert relerr(model.V, g[tag + "V_after"]) < 1e-6
    assert abs(model.log_likelihood(Y) - orc.family_loglik(model.W, model.V, Y, family, par)) < 1e-9 * abs(ll0)
    for ess in ("joint", "rows"):
        np.random.seed(4)            # (no Tau2_init: with it the reference - and the build - have no horseshoe+ levels to update, quirk Q6)
        m2 = NonconjugateBayesianTensorFiltering(N, M, T, family, likelihood_param=par, nembeds=K, tf_order=tf, sigma2_init=st["sigma2"],
                                                 lam2_init=st["lam2"], W_init=st["W"], V_init=st["V"], rng="device", ess=ess)
        l0 = m2.log_likelihood(Y)
        for _ in range(30):
            m2.resample(Y)
        l1 = m2.log_likelihood(Y)
        assert np.isfinite(l1) and l1 > l0 - 0.5 * abs(l0), (ess, l0, l1)


@pytest.mark.parametrize("ess", ["joint", "rows"])
def test_nonconjugate_device_slices_stay_on_the_slice_and_recover_rates(ess):
    """rng="device": every update must leave the log-likelihood finite; the per-row / per-column slices must find the
    Poisson rates of a small synthetic tensor (the joint slice - the reference's scheme - moves all of W at once and
    mixes slowly at this dimension: for it only the climb of the likelihood is checked here, its stationary
    distribution in the next test)."""
    from functionalmf_amd.factor import NonconjugateBayesianTensorFiltering
    rs = np.random.RandomState(4)
    N, M, T, R, K = 24, 10, 12, 3, 2
    Wt = 0.6 * rs.normal(size=(N, K))
    Vt = 0.3 * np.cumsum(rs.normal(size=(M, T, K)), axis=1)
    Wt[:, 0] = 1.0 + 0.2 * rs.normal(size=N)
    Vt[:, :, 0] += 1.0                                      # a level the first embedding carries
    Mu = np.einsum("nk,mtk->nmt", Wt, Vt)
    Y = rs.poisson(np.repeat(np.exp(Mu)[..., None], R, axis=-1)).astype(float)
    Y[rs.rand(N, M, T, R) < 0.05] = np.nan
    np.random.seed(5)
    model = NonconjugateBayesianTensorFiltering(N, M, T, "poisson_log", nembeds=K + 1, tf_order=1, sigma2_init=1.0, lam2_init=0.5,
                                                rng="device", ess=ess, device_seed=3)
    ll_start = model.log_likelihood(Y)
    nsweeps = 1500 if ess == "rows" else 300
    keep = []
    for s in range(nsweeps):
        model.resample(Y)
        if s >= nsweeps // 2 and s % 5 == 0:
            keep.append(np.einsum("nk,mtk->nmt", model.W, model.V))
    assert model.ess_unfinished() == 0
    ll_end = model.log_likelihood(Y)
    assert np.isfinite(ll_end) and ll_end > ll_start
    est = np.mean(keep, axis=0)
    corr = np.corrcoef(est.reshape(-1), Mu.reshape(-1))[0, 1]
    if ess == "rows":
        assert corr > 0.85, corr


@pytest.mark.parametrize("ess", ["rows", "joint"])
def test_nonconjugate_device_slices_sample_the_conditional(ess):
    """With V fixed the rows of W are independent: repeated device-driven updates (one slice per row, or one joint
    slice over the six rows) must reproduce the conditional posterior mean and standard deviation of every row,
    computed by quadrature (K = 1)."""
    from functionalmf_amd.factor import NonconjugateBayesianTensorFiltering
    rs = np.random.RandomState(8)
    N, M, T = 6, 3, 5
    v = 0.5 + 0.1 * rs.normal(size=(M, T, 1))
    Y = rs.poisson(np.exp(0.7 * v[None, :, :, 0].repeat(N, 0))).astype(float)
    sigma2 = 0.8
    np.random.seed(1)
    model = NonconjugateBayesianTensorFiltering(N, M, T, "poisson_log", nembeds=1, tf_order=0, sigma2_true=sigma2, lam2_true=0.3,
                                                V_true=v, Tau2_true=np.ones((M, T)), rng="device", ess=ess, device_seed=11)
    draws = []
    for s in range(6000 if ess == "rows" else 20000):
        model._resample_W(Y)
        if s >= 200:
            draws.append(model.W[:, 0].copy())
    draws = np.array(draws)
    grid = np.linspace(-4, 4, 8001)
    for i in range(N):
        lp = -0.5 * grid ** 2 / sigma2
        for j in range(M):
            for t in range(T):
                lp += Y[i, j, t] * grid * v[j, t, 0] - np.exp(grid * v[j, t, 0])
        p = np.exp(lp - lp.max())
        p /= p.sum()
        m = (grid * p).sum()
        sd = np.sqrt(((grid - m) ** 2 * p).sum())
        # slice-sampler draws are autocorrelated: a generous effective sample size
        assert abs(draws[:, i].mean() - m) < 6 * sd / np.sqrt(len(draws) / 10), (i, draws[:, i].mean(), m)
        assert abs(draws[:, i].std() - sd) < 0.1 * sd, (i, draws[:, i].std(), sd)


# ---- curve-structured replicate counts (whole curves missing / thinned): complete-data kernels + corrections ----
def _curve_data(seed=3, dims=(14, 9, 13, 3, 3, 2)):
    N, M, T, R, K, tf = dims
    rs = np.random.RandomState(seed)
    Wt = rs.normal(size=(N, K))
    Wt[np.triu_indices(K, 1)] = 0
    Vt = 0.3 * np.cumsum(rs.normal(size=(M, T, K)), axis=1)
    Y = np.einsum("nk,mtk->nmt", Wt, Vt)[..., None] + rs.normal(0, 0.5, size=(N, M, T, R))
    Y[:3, :3] = np.nan                      # held-out block, as examples/gaussian_tensor_filtering.py:16-18
    Y[7, 5] = np.nan                        # one more whole curve
    Y[4, 2, :, 1] = np.nan                  # thinned curves: one / two replicates gone at every depth
    Y[9, 6, :, :2] = np.nan
    Y[12, 0, :, 2] = np.nan
    st = dict(W=Wt + 0.1 * rs.normal(size=Wt.shape), V=Vt + 0.1 * rs.normal(size=Vt.shape), lam2=0.2, sigma2=0.6, nu2=0.4)
    st["W"][np.triu_indices(K, 1)] = 0
    from oracle import btf_oracle as orc
    st["Tau2"] = rs.gamma(2.0, 0.5, size=(M, orc.trend_penalty(T, tf).shape[0]))
    return Y, st, dims


def _curve_model(Y, st, dims, chain=False, **kw):
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    N, M, T, R, K, tf = dims
    if not chain:
        kw["Tau2_init"] = st["Tau2"]        # (whole sweeps draw Tau2: the model must build its own horseshoe+ chain, quirk Q6)
    return GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=tf, sigma2_init=st["sigma2"], lam2_init=st["lam2"],
                                           nu2_init=st["nu2"], W_init=st["W"], V_init=st["V"], compat="exact", **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("sampler", ["spectral", "banded"])
def test_curve_counts_half_sweeps_vs_oracle_and_weighted_form(sampler, monkeypatch):
    """Counts constant along the depth axis: the W and V half-sweeps run the complete-data stream plus per-row /
    per-column corrections.  They must equal (i) the oracle's weighted steps (factor.py:343-346, :388-391 restated)
    and (ii) the same model with BTF_OPT_CURVE_COUNTS off, from the same state and normals; the residual sum of
    squares taken from the W partials must equal the oracle's."""
    from oracle import btf_oracle as orc
    from functionalmf_amd import _native
    Y, st, dims = _curve_data()
    N, M, T, R, K, tf = dims
    Delta = orc.trend_penalty(T, tf)
    rs = np.random.RandomState(11)
    zw = rs.normal(size=K * (K + 1) // 2 + (N - K) * K)
    zv = rs.normal(size=(M, K * T))
    out = {}
    for curve in (1, 0):
        model = _curve_model(Y, st, dims, sampler=sampler)
        model.set_data(Y)
        model._ctx.call("btf_set_option", _native.OPT_CURVE_COUNTS, curve)
        assert model.likelihood_form() == ("curve_counts" if curve else "weighted")
        assert model.v_sampler() == (sampler if curve else "banded")          # the weighted form has no spectral sampler
        monkeypatch.setattr(model, "_w_normals", lambda: zw)
        monkeypatch.setattr(model, "_v_normals", lambda: zv)
        model._resample_W(Y)
        Wd = model.W.copy()
        model._resample_V(Y)
        out[curve] = (Wd, model.V.copy(), model.v_order())
    o = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in st.items()}
    orc.w_step(o, Y, z=zw)
    assert relerr(out[1][0], o["W"]) < 1e-10 and relerr(out[0][0], o["W"]) < 1e-10
    perm = "spectral" if sampler == "spectral" else orc.perm_from_order(out[1][2], K, T)
    orc.v_step(o, Y, Delta, perm=perm, z=zv, compat="exact")
    assert relerr(out[1][1], o["V"]) < 1e-7
    if sampler == "banded":
        assert relerr(out[1][1], out[0][1]) < 1e-7


@pytest.mark.gpu
def test_curve_counts_device_chain_matches_weighted_form():
    """rng="device", whole sweeps: with the banded sampler the curve-counts form and the weighted form consume the same
    Philox streams, so the two chains must coincide (up to rounding) - scalars drawn from the partials' residual sum of
    squares included."""
    from functionalmf_amd import _native
    Y, st, dims = _curve_data(seed=5, dims=(40, 12, 16, 3, 3, 2))
    chains = []
    for curve in (1, 0):
        np.random.seed(7)
        m = _curve_model(Y, st, dims, chain=True, sampler="banded", rng="device", device_seed=13)
        m.set_data(Y)
        m._ctx.call("btf_set_option", _native.OPT_CURVE_COUNTS, curve)
        assert m.likelihood_form() == ("curve_counts" if curve else "weighted")
        for _ in range(4):
            m.resample(Y)
        chains.append((m.W.copy(), m.V.copy(), float(m.nu2), float(m.sigma2), np.array(m.Tau2).copy()))
    a, b = chains
    assert abs(a[2] - b[2]) / b[2] < 1e-9 and abs(a[3] - b[3]) / b[3] < 1e-9
    assert relerr(a[0], b[0]) < 1e-7 and relerr(a[1], b[1]) < 1e-5 and relerr(a[4], b[4]) < 1e-5


@pytest.mark.gpu
def test_counts_varying_with_depth_keep_the_weighted_form():
    Y, st, dims = _curve_data()
    Y[5, 5, 3, 0] = np.nan                  # a single replicate at a single depth
    model = _curve_model(Y, st, dims, sampler="spectral")
    model.set_data(Y)
    assert model.likelihood_form() == "weighted" and model.v_sampler() == "banded"


@pytest.mark.gpu
def test_curve_counts_spectral_chain_samples_the_same_posterior():
    """The per-column eigen-systems are warm-started from sweep to sweep: a longer rng="device" chain with the spectral
    sampler must agree with the banded-sampler chain in its posterior means (two square roots, one distribution)."""
    Y, st, dims = _curve_data(seed=9, dims=(30, 10, 12, 3, 3, 2))
    means = {}
    for sampler in ("spectral", "banded"):
        np.random.seed(3)
        m = _curve_model(Y, st, dims, chain=True, sampler=sampler, rng="device", device_seed=21, lam2_true=0.2)
        acc, n = 0.0, 0
        for it in range(700):
            m.resample(Y)
            if it >= 100:
                acc = acc + np.einsum("nk,mtk->nmt", m.W, m.V)
                n += 1
        assert m.likelihood_form() == "curve_counts" and m.v_sampler() == sampler
        means[sampler] = acc / n
    d = np.abs(means["spectral"] - means["banded"])
    assert d.mean() < 0.05 and np.corrcoef(means["spectral"].ravel(), means["banded"].ravel())[0, 1] > 0.995


# ---- dense fast_mvn branches on the device (btf_mvn_dense; fast_mvn.py:49-60, :126-142, :145-179) ---------------
@pytest.mark.gpu
def test_dense_mvn_branches_vs_reference_fixture(golden):
    """The ten argument combinations the reference's own sample_mvn produced (tests/golden/make_golden_ess.py (b)),
    through the product's dispatcher and the dense kernel, from the same legacy-RNG normals."""
    from functionalmf_amd.fast_mvn import sample_mvn
    g = golden("g9_ess.npz")
    S, mu = g["mvn_S"], g["mvn_mu"]
    kws = (dict(precision=True), dict(precision=True, mu_part=mu), dict(precision=True, mu=mu), dict(precision=False),
           dict(precision=False, mu=mu), dict(precision=False, mu_part=mu), dict(precision=False, chol_factor=True),
           dict(precision=True, chol_factor=True, mu_part=mu))
    for i, kw in enumerate(kws):
        Q = np.linalg.cholesky(S) if kw.get("chol_factor") else S
        np.random.seed(900 + i)
        x = sample_mvn(Q, sparse=False, Q_shape=Q.shape if kw.get("chol_factor") else None, **kw)
        assert relerr(x, g["mvn_out"][i]) < 1e-12, kw
    np.random.seed(950)
    assert relerr(sample_mvn(0.7, mu=mu, sparse=False), g["mvn_out"][8]) < 1e-12
    np.random.seed(951)
    assert relerr(sample_mvn(np.full(mu.size, 2.5), mu_part=mu, sparse=False, precision=True), g["mvn_out"][9]) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 33, 130, 300])
def test_dense_mvn_batches_vs_oracle(n):
    from oracle import btf_oracle as orc
    from functionalmf_amd.fast_mvn import sample_dense_batch
    rs = np.random.RandomState(n)
    B = 5
    A = rs.normal(size=(B, n, n))
    S = A @ A.transpose(0, 2, 1) + n * np.eye(n)
    m = rs.normal(size=(B, n))
    z = rs.normal(size=(B, n))
    for prec in (True, False):
        for fac in (False, True):
            Q = np.linalg.cholesky(S) if fac else S
            for kw in (dict(), dict(mu=m), dict(mu_part=m)):
                x, tries = sample_dense_batch(Q, precision=prec, chol_factor=fac, z=z, **kw)
                ref = np.stack([orc.sample_mvn_dense(Q[b], precision=prec, chol_factor=fac, z=z[b],
                                                     **{k: v[b] for k, v in kw.items()}) for b in range(B)])
                assert relerr(x, ref) < 1e-9 and not tries.any(), (n, prec, fac, list(kw))


@pytest.mark.gpu
def test_dense_mvn_jitter_schedule_and_failure():
    """fast_mvn.py:62-68 / :133-139: eps, then 10 eps more, ... added to the diagonal until the factorisation goes
    through; without force_psd the failure is reported (np.linalg.cholesky raises LinAlgError there)."""
    from oracle import btf_oracle as orc
    from functionalmf_amd import _native
    from functionalmf_amd.fast_mvn import sample_dense_batch
    rs = np.random.RandomState(4)
    n = 12
    A = rs.normal(size=(n, 3))
    S = A @ A.T                                        # rank 3: not positive definite
    S[np.diag_indices(n)] -= 5e-6                      # needs 1e-6 + 1e-5 (two shifts)
    z = rs.normal(size=(1, n))
    with pytest.raises(_native.NotPositiveDefiniteError):
        sample_dense_batch(S[None], precision=True, z=z)
    x, tries = sample_dense_batch(S[None], precision=True, z=z, force_psd=True, force_psd_eps=1e-6, force_psd_attempts=4)
    assert tries[0] == 2
    ref = orc.sample_mvn_dense(S + 1.1e-5 * np.eye(n), precision=True, z=z[0])
    assert relerr(x[0], ref) < 1e-6


@pytest.mark.gpu
def test_dense_mvn_device_rng_has_the_right_covariance():
    from functionalmf_amd.fast_mvn import sample_dense_batch
    rs = np.random.RandomState(8)
    n, B = 6, 40000
    A = rs.normal(size=(n, n))
    S = A @ A.T + np.eye(n)
    mu = rs.normal(size=n)
    x, _ = sample_dense_batch(np.broadcast_to(S, (B, n, n)).copy(), precision=False, mu=np.broadcast_to(mu, (B, n)).copy(), seed=5)
    assert np.abs(x.mean(0) - mu).max() < 5 * np.sqrt(np.diag(S).max() / B)
    assert np.abs(np.cov(x.T) - S).max() < 0.06 * np.abs(S).max()
    xp, _ = sample_dense_batch(np.broadcast_to(S, (B, n, n)).copy(), precision=True, seed=6)
    assert np.abs(np.cov(xp.T) - np.linalg.inv(S)).max() < 0.06 * np.abs(np.linalg.inv(S)).max()


# ---- constrained non-conjugate model: GASS on the device (btf_gass_*; gass.py:13-130, factor.py:665-855) ---------
def _gass_model(golden, **kw):
    from functionalmf_amd.factor import ConstrainedNonconjugateBayesianTensorFiltering
    from test_oracle_golden import _gass_case
    g, st, (N, M, T, R, K, tf) = _gass_case(golden)
    model = ConstrainedNonconjugateBayesianTensorFiltering(
        N, M, T, "poisson_identity", g["Cons"], Row_constraints=g["Row_constraints"], gass_ngrid=int(g["ngrid"]),
        nembeds=K, tf_order=tf, sigma2_init=st["sigma2"], lam2_init=st["lam2"], W_init=st["W"], V_init=st["V"],
        Tau2_init=st["Tau2"], **kw)
    return model, g, st, (N, M, T, R, K, tf)


@pytest.mark.gpu
def test_gass_row_and_column_updates_vs_reference_fixture(golden):
    """The reference's own _resample_W_i / _resample_V_j (tests/golden/make_golden_gass.py) against the device path
    driven with the same per-chain streams: constraint analysis, candidate likelihoods and the commit run on the GPU,
    the reference's np.random.choice calls on the host."""
    model, g, st, (N, M, T, R, K, tf) = _gass_model(golden, sampler="banded")
    model.chain_rngs = lambda what: [np.random.RandomState((2000 if what == 0 else 3000) + c) for c in range(N if what == 0 else M)]
    model._resample_W(g["Y"])
    assert relerr(model.W, g["W_after"]) < 1e-10
    assert model.gass_info["accepted"].max() > 0
    model.W = st["W"]
    model._resample_V(g["Y"])
    assert relerr(model.V, g["V_after"]) < 1e-7


def test_identity_link_logarithm_is_accurate_to_double_precision():
    """The identity-link Poisson term takes its log from a 128-entry table + a degree-8 log1p (csrc/btf_ess.h: log_tab)
    instead of libm.  Rates that depend on the row alone make a row's log-likelihood M T (y log w_i - w_i) with the
    logarithm of ONE number in it: the per-row values the GASS set-up reports (btf_gass_grid: cur_ll) against numpy, for
    rates across 30 binary orders of magnitude, next to 1 and at interval edges of the table."""
    import ctypes
    from functionalmf_amd.factor import ConstrainedNonconjugateBayesianTensorFiltering
    from functionalmf_amd import _native
    N, M, T, K, y = 96, 3, 8, 2, 3.0
    rs = np.random.RandomState(8)
    w = np.concatenate([2.0 ** rs.uniform(-15, 15, size=N - 24), 1 + rs.uniform(-1e-3, 1e-3, size=8),
                        1 + np.arange(8) / 128.0, (1 + np.arange(8) / 128.0) * (1 - 2.0 ** -52)])
    W = np.zeros((N, K)); W[:, 0] = w
    V = np.zeros((M, T, K)); V[..., 0] = 1.0; V[..., 1] = 0.5
    Cons = np.concatenate([np.eye(T), np.zeros((T, 1))], axis=1)
    model = ConstrainedNonconjugateBayesianTensorFiltering(N, M, T, "poisson_identity", Cons, gass_ngrid=16, nembeds=K, tf_order=0,
                                                           sigma2_init=1.0, lam2_init=0.1, W_init=W, V_init=V,
                                                           Tau2_init=np.ones((M, T)), sampler="banded")
    Y = np.full((N, M, T, 1), y)
    model._bind_data(Y)
    model._push_state()
    model._ctx.call("btf_gass_set_constraints", _native.dptr(model._cons), int(model._cons.shape[0]), None, 0)
    z = rs.normal(size=K * (K + 1) // 2 + (N - K) * K)
    u = rs.rand(N)
    model._ctx.call("btf_gass_begin", 0, 1, _native.dptr(z), _native.dptr(u), 1, 1e-6, 0, 0)
    info = np.zeros((N, 2), dtype=np.int32)
    cur = np.empty(N)
    model._ctx.call("btf_gass_grid", 0, info.ctypes.data_as(_native._c_ip), None, None, _native.dptr(cur))
    ref = M * T * (y * np.log(w) - w)
    # error budget: the logarithm's (a few 1e-16 of max(|log w|, 1)) times y M T, plus the summation's
    tol = 4e-16 * M * T * (y * np.maximum(np.abs(np.log(w)), 1.0) + w) + 1e-15 * np.abs(ref)
    assert np.all(np.abs(cur - ref) <= tol), (np.abs(cur - ref) / tol).max()


@pytest.mark.gpu
def test_gass_valid_grid_equals_the_oracles(golden):
    """The validity of each of the 10000 grid angles (difference array + scan on the device) against the oracle's
    sequential pruning (gass.py:66-80), for every row and every column, from random proposals."""
    import ctypes
    from oracle import btf_oracle as orc
    from functionalmf_amd import _native
    model, g, st, (N, M, T, R, K, tf) = _gass_model(golden, sampler="banded")
    Y = g["Y"]
    model._bind_data(Y)
    model._push_state()
    rc = model.Row_constraints
    model._ctx.call("btf_gass_set_constraints", _native.dptr(model._cons), int(model._cons.shape[0]), _native.dptr(rc), int(rc.shape[0]))
    rs = np.random.RandomState(5)
    full = np.linspace(-np.pi, np.pi, 10000)
    # rows
    z = rs.normal(size=K * (K + 1) // 2 + (N - K) * K)
    u = rs.rand(N)
    model._ctx.call("btf_gass_begin", 0, 1, _native.dptr(z), _native.dptr(u), 1, 1e-6, 0, 0)
    info = np.zeros((N, 2), dtype=np.int32)
    mask = np.zeros((N, 10000), dtype=np.uint8)
    hh, cur = np.empty(N), np.empty(N)
    model._ctx.call("btf_gass_grid", 0, info.ctypes.data_as(_native._c_ip), mask.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                    _native.dptr(hh), _native.dptr(cur))
    off = 0
    for i in range(N):
        d = min(K, i + 1)
        v = np.sqrt(st["sigma2"]) * z[off:off + d]
        off += d
        Ci = orc.constrained_w_constraints(st["V"], g["Cons"], d, g["Row_constraints"])
        grid, restricted = orc.gass_valid_grid(st["W"][i, :d], v, Ci[:, :-1], Ci[:, -1])
        assert bool(info[i, 1]) == (not restricted)
        if restricted:
            assert np.array_equal(full[mask[i] != 0], grid), i
        ll = orc.poisson_curves_loglik(Y[i], np.einsum("jtk,k->jt", st["V"][:, :, :d], st["W"][i, :d]), "identity")
        assert abs(cur[i] - ll) < 1e-9 * max(1.0, abs(ll)) and abs(hh[i] - (ll + np.log(u[i]))) < 1e-9 * max(1.0, abs(ll))
    # columns
    Delta = orc.trend_penalty(T, tf)
    p = orc.perm_from_order(model.v_order(), K, T)
    zv = rs.normal(size=(M, K * T))
    uv = rs.rand(M)
    model._ctx.call("btf_gass_begin", 1, 1, _native.dptr(zv), _native.dptr(uv), 2, 1e-6, 0, 0)
    info = np.zeros((M, 2), dtype=np.int32)
    mask = np.zeros((M, 10000), dtype=np.uint8)
    model._ctx.call("btf_gass_grid", 1, info.ctypes.data_as(_native._c_ip), mask.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), None, None)
    Cv = orc.constrained_v_constraints(st["W"], g["Cons"], T)
    for j in range(M):
        Q = np.kron(np.eye(K), orc.prior_precision_1d(Delta, st["lam2"], st["Tau2"][j]))
        v = orc.mvn_from_precision(Q, perm=p, z=zv[j])
        grid, restricted = orc.gass_valid_grid(st["V"][j].T.reshape(-1), v, Cv[:, :-1], Cv[:, -1])
        assert bool(info[j, 1]) == (not restricted)
        if restricted:
            dev = full[mask[j] != 0]
            # (a proposal from an ill-conditioned prior: arc ends agree to ~1e-9, one boundary angle may differ)
            assert abs(len(dev) - len(grid)) <= 2 and len(np.setxor1d(dev, grid)) <= 2, j


@pytest.mark.gpu
def test_gass_device_chain_stays_feasible_and_fits(golden):
    """rng="device": whole sweeps on the GPU.  Every state satisfies every constraint (positivity and monotone curves),
    and the chain moves to a good fit of the Poisson rates it was simulated from."""
    from functionalmf_amd.factor import ConstrainedNonconjugateBayesianTensorFiltering
    rs = np.random.RandomState(12)
    N, M, T, R, K = 24, 10, 12, 3, 3
    Wt = rs.gamma(2.0, 0.5, size=(N, K))
    Wt[np.triu_indices(K, 1)] = 0
    Vt = np.zeros((M, T, K))
    for j in range(M):
        Vt[j, -1] = rs.gamma(2.0, 0.5, size=K)
        for t in range(T - 2, -1, -1):
            Vt[j, t] = Vt[j, t + 1] + (rs.gamma(1.0, 0.6, size=K) if rs.rand() < 0.4 else 0.0)
    rate = np.einsum("nk,mtk->nmt", Wt, Vt)
    Y = rs.poisson(np.repeat(rate[..., None], R, axis=-1)).astype(float)
    Y[:2, :2] = np.nan
    Cons = np.concatenate([np.eye(T), np.zeros((T, 1))], axis=1)
    mono = np.array([np.concatenate([np.zeros(t), [1, -1], np.zeros(T - t - 2), [-1e-2]]) for t in range(T - 1)])
    Cons = np.concatenate([Cons, mono], axis=0)
    np.random.seed(4)
    W0 = np.abs(Wt + 0.3 * rs.normal(size=Wt.shape)) + 0.05
    W0[np.triu_indices(K, 1)] = 0
    V0 = np.maximum.accumulate((np.abs(Vt + 0.2 * rs.normal(size=Vt.shape)) + 0.05)[:, ::-1], axis=1)[:, ::-1]   # decreasing in t
    model = ConstrainedNonconjugateBayesianTensorFiltering(N, M, T, "poisson_identity", Cons, gass_ngrid=64, nembeds=K, tf_order=0,
                                                           sigma2_init=1.0, lam2_init=0.5, W_init=W0, V_init=V0, rng="device",
                                                           device_seed=3)
    ll0 = model.log_likelihood(Y)
    acc, n = 0.0, 0
    for it in range(300):
        model.resample(Y)
        if it % 25 == 24 or it >= 200:
            tau = np.einsum("nk,mtk->nmt", model.W, model.V)
            assert (np.einsum("qt,nmt->nmq", Cons[:, :-1], tau) >= Cons[:, -1] - 1e-9).all(), it
        if it >= 150:
            acc = acc + np.einsum("nk,mtk->nmt", model.W, model.V)
            n += 1
    assert model.log_likelihood(Y) > ll0
    fit = acc / n
    assert np.corrcoef(fit.ravel(), rate.ravel())[0, 1] > 0.9


@pytest.mark.gpu
def test_gass_with_a_gaussian_likelihood_keeps_monotone_curves_and_fits():
    """The constrained model with another device likelihood (link 3, Gaussian with known variance): monotone curves
    observed with noise.  Every state keeps every curve monotone, and the posterior mean recovers the curves."""
    from functionalmf_amd.factor import ConstrainedNonconjugateBayesianTensorFiltering
    rs = np.random.RandomState(13)
    N, M, T, R, K = 20, 8, 12, 3, 3
    Wt = rs.gamma(2.0, 0.4, size=(N, K))
    Vt = np.cumsum(rs.gamma(1.0, 0.15, size=(M, T, K)), axis=1)
    curve = np.einsum("nk,mtk->nmt", Wt, Vt)
    Y = curve[..., None] + rs.normal(0, 0.3, size=(N, M, T, R))
    Y[:2, :2] = np.nan
    Cons = np.zeros((T - 1, T + 1))                                        # monotone (a Gaussian mean may be negative)
    Cons[np.arange(T - 1), np.arange(T - 1)] = -1.0
    Cons[np.arange(T - 1), np.arange(1, T)] = 1.0
    np.random.seed(5)
    model = ConstrainedNonconjugateBayesianTensorFiltering(N, M, T, "gaussian", Cons, likelihood_param=0.09, nembeds=K, tf_order=0,
                                                           sigma2_init=1.0, lam2_init=0.1, W_init=Wt * 0 + 0.5,
                                                           V_init=np.repeat(np.linspace(0.1, 1.0, T)[None, :, None], M, axis=0).repeat(K, axis=2),
                                                           rng="device", device_seed=3)
    ll0 = model.log_likelihood(Y)
    acc, n = 0.0, 0
    for it in range(300):
        model.resample(Y)
        if it % 25 == 24 or it >= 200:
            tau = np.einsum("nk,mtk->nmt", model.W, model.V)
            assert (np.einsum("qt,nmt->nmq", Cons[:, :-1], tau) >= Cons[:, -1] - 1e-9).all(), it
        if it >= 150:
            acc = acc + np.einsum("nk,mtk->nmt", model.W, model.V)
            n += 1
    assert model.log_likelihood(Y) > ll0
    assert np.corrcoef((acc / n).ravel(), curve.ravel())[0, 1] > 0.9


@pytest.mark.gpu
def test_constrained_poisson_example_runs_end_to_end():
    """examples/poisson_constrained_tensor_filtering.py: positivity + monotone curves, GASS on the device."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("poisson_constrained", os.path.join(ROOT, "examples", "poisson_constrained_tensor_filtering.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rel_in, rel_out, corr, feasible = mod.main(seed=1, nburn=400, nsamples=200)
    assert feasible and corr > 0.8 and rel_in < 0.5


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["g2_c2_complete.npz", "g1_c1_heldout.npz", "g3_partial_reps.npz"])
def test_c_driven_sweeps_walk_the_python_driven_chain(golden, name, monkeypatch):
    """btf_gibbs_sweeps queues whole sweeps from the C side with the seed sequence the Python driver consumes: the two
    chains - and two run_gibbs result dicts - must be bit-identical."""
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    g = golden(name)
    N, M, T, R, K, tf = [int(x) for x in g["dims"]]
    st = state_from(g, "s0_")

    def make():
        np.random.seed(11)
        return GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=tf, sigma2_init=st["sigma2"], lam2_init=st["lam2"],
                                               nu2_init=st["nu2"], W_init=st["W"], V_init=st["V"], rng="device", device_seed=5)
    a, b = make(), make()
    assert b._sweeps_on_device()
    for _ in range(7):
        a.resample(g["Y"])
    b.resample_sweeps(g["Y"], 3)
    b.resample_sweeps(g["Y"], 4)
    for x, y in ((a.W, b.W), (a.V, b.V), (a.Tau2, b.Tau2)):
        assert np.array_equal(np.asarray(x), np.asarray(y))
    assert (a.nu2, a.sigma2, a.lam2) == (b.nu2, b.sigma2, b.lam2)
    a.resample(g["Y"])
    b.resample(g["Y"])                       # the drivers can be mixed
    assert np.array_equal(a.V, b.V)
    # run_gibbs: the C-driven loop against the step-by-step one
    c, d = make(), make()
    monkeypatch.setattr(d, "_sweeps_on_device", lambda: False)
    rc = c.run_gibbs(g["Y"], nburn=3, nthin=2, nsamples=4, verbose=False)
    rd = d.run_gibbs(g["Y"], nburn=3, nthin=2, nsamples=4, verbose=False)
    assert set(rc) == set(rd)
    for k in rc:
        assert np.array_equal(rc[k], rd[k]), k
    # the same run queued in blocks of two sweeps (verbose: a block per progress line), no burn-in, every sweep kept:
    # the collection schedule of the C side survives the calls in between
    e = make()
    re = e.run_gibbs(g["Y"], nburn=3, nthin=2, nsamples=4, verbose=True, print_freq=2)
    for k in rc:
        assert np.array_equal(rc[k], re[k]), k
    f, h = make(), make()
    monkeypatch.setattr(h, "_sweeps_on_device", lambda: False)
    rf = f.run_gibbs(g["Y"], nburn=0, nthin=1, nsamples=5, verbose=False)
    rh = h.run_gibbs(g["Y"], nburn=0, nthin=1, nsamples=5, verbose=False)
    for k in rf:
        assert np.array_equal(rf[k], rh[k]), k


@pytest.mark.gpu
def test_in_place_edit_of_a_few_cells_is_noticed(golden):
    """The reference re-reads Y every half-sweep (factor.py:329-330); the build uploads it once.  A few cells set to NaN in
    place after the first sweep slip past the 64-point fingerprint; the periodic hash of the whole array
    (data_check_seconds) must catch them, warn and re-upload, after which the half-sweep equals one on a fresh model."""
    import warnings
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    g = golden("g2_c2_complete.npz")
    N, M, T, R, K, tf = [int(x) for x in g["dims"]]
    st = state_from(g, "s0_")
    Y = g["Y"].copy()

    def make():
        return GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=tf, sigma2_init=st["sigma2"], lam2_init=st["lam2"],
                                               nu2_init=st["nu2"], W_init=st["W"], V_init=st["V"], Tau2_init=st["Tau2"])
    a = make()
    a.data_check_seconds = 0.0
    np.random.seed(3)
    a._resample_W(Y)
    Y[5:8, 3:5] = np.nan                                   # 6 curves of 2048: the fingerprint does not see them
    a.W = st["W"]
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        np.random.seed(3)
        a._resample_W(Y)
    assert any("modified in place" in str(w.message) for w in wl)
    b = make()
    np.random.seed(3)
    b._resample_W(Y)
    assert np.array_equal(a.W, b.W)
    with warnings.catch_warnings(record=True) as wl:       # nothing changed: no warning, no upload
        warnings.simplefilter("always")
        a._resample_W(Y)
    assert not wl


@pytest.mark.gpu
@pytest.mark.parametrize("compat", ["reference", "exact"])
def test_four_launch_sweep_walks_the_six_launch_chain(golden, compat):
    """BTF_OPT_FUSED_SWEEP (include/btf.h): from the second sweep on a full device sweep on complete Gaussian data is four
    launches - nu2 | rest and sigma2 | rest ride in the W accumulation launch, fed by the per-column residual parts the
    spectral V sampler left behind; lam2 | rest rides in the V accumulation launch - instead of six.  Same conditionals,
    same Philox streams: the chains coincide up to the rounding of differently grouped sums (factor.py:306-311, :411-416,
    :130-153 on the device).  The residual parts themselves are checked against the oracle's residual sum of squares."""
    import ctypes
    from oracle import btf_oracle as orc
    from functionalmf_amd import _native
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    g = golden("g2_c2_complete.npz")
    N, M, T, R, K, tf = [int(x) for x in g["dims"]]
    st = state_from(g, "s0_")

    def make(fused):
        np.random.seed(11)
        m = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=tf, sigma2_init=st["sigma2"], lam2_init=st["lam2"],
                                            nu2_init=st["nu2"], W_init=st["W"], V_init=st["V"], rng="device", device_seed=5, compat=compat)
        m._ctx.call("btf_set_option", _native.OPT_FUSED_SWEEP, 1 if fused else 0)
        return m
    a, b = make(True), make(False)
    a.resample(g["Y"]); b.resample(g["Y"])                  # the first sweep has no residual parts yet: six launches both
    for m, want in ((a, 4), (b, 6)):
        m._ctx.kernel_times()
        m.resample_sweeps(g["Y"], 5)
        kt = m._ctx.kernel_times()
        assert sum(v[1] for v in kt.values()) == 5 * want, kt
    for x, y, tol in ((a.W, b.W, 1e-8), (a.V, b.V, 1e-6), (a.Tau2, b.Tau2, 1e-6)):
        assert relerr(np.asarray(x), np.asarray(y)) < tol
    for x, y in ((a.nu2, b.nu2), (a.sigma2, b.sigma2), (a.lam2, b.lam2)):
        assert abs(x - y) / abs(y) < 1e-8, (x, y)
    # the Python-driven sweep takes the same decisions: bit-identical to the C-driven one
    c, d = make(True), make(True)
    for _ in range(4):
        c.resample(g["Y"])
    d.resample_sweeps(g["Y"], 4)
    assert np.array_equal(c.W, d.W) and np.array_equal(c.V, d.V) and (c.nu2, c.sigma2, c.lam2) == (d.nu2, d.sigma2, d.lam2)
    # the residual sum of squares the next nu2 draw would use = the oracle's, for the state as it stands
    a.sync()
    sc = np.zeros(8)
    a._ctx.call("btf_get_scalars", _native.dptr(sc))          # HYP_SSE of the last draw: the state BEFORE the last W, V
    stn = dict(W=a.W.copy(), V=a.V.copy(), nu2=float(a.nu2))
    a.resample(g["Y"])
    a._ctx.call("btf_get_scalars", _native.dptr(sc))
    sse, n = orc.sse_and_count(stn, g["Y"])
    assert abs(sc[4] - sse) / sse < 1e-10, (sc[4], sse)


@pytest.mark.gpu
@pytest.mark.parametrize("ntrials", [4, 150, 2.5])
def test_binomial_pseudo_data_bytes_and_fallback(ntrials):
    """Integer counts up to 127 stream the pseudo-data kappa = Y - N/2 as one byte per cell (9 B/cell); larger or
    non-integer trial counts keep the f64 array (16 B/cell).  Either way the weighted half-sweeps given omega equal
    the oracle's (factor.py:437-460 with :343-346 / :388-391)."""
    import ctypes
    from oracle import btf_oracle as orc
    from functionalmf_amd.factor import BinomialBayesianTensorFiltering
    rs = np.random.RandomState(17)
    N, M, T, K, tf = 14, 6, 10, 3, 2
    Wt = rs.normal(size=(N, K))
    Vt = 0.3 * np.cumsum(rs.normal(size=(M, T, K)), axis=1)
    p = 1 / (1 + np.exp(-np.einsum("nk,mtk->nmt", Wt, Vt)))
    Ntr = np.full((N, M, T), float(ntrials))
    Ys = rs.binomial(int(np.ceil(ntrials)), p).astype(float) * (ntrials / np.ceil(ntrials))
    Ys[:2, :2] = np.nan
    Ntr[:2, :2] = np.nan
    st = dict(W=rs.normal(size=(N, K)), V=0.2 * rs.normal(size=(M, T, K)), Tau2=np.exp(rs.normal(size=(M, 3 * T - 1))), lam2=0.3,
              sigma2=0.8)
    st["W"][np.triu_indices(N, 1, K)] = 0
    model = BinomialBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=tf, sigma2_init=st["sigma2"], lam2_init=st["lam2"],
                                            W_init=st["W"], V_init=st["V"], Tau2_init=st["Tau2"], compat="exact")
    data = (Ys, Ntr)
    model._bind_data(data)
    b = ctypes.c_double()
    model._ctx.call("btf_get_accum_bytes_per_cell", ctypes.byref(b))
    assert b.value == (9.0 if ntrials == 4 else 16.0)
    omega = rs.gamma(2.0, 0.2, size=(N, M, T))
    with np.errstate(divide="ignore"):
        nu2 = np.where(np.isnan(Ys), np.inf, 1.0 / omega)
    model.nu2 = nu2                                   # 1 / omega, as factor.py:459-460 leaves it
    np.random.seed(9)
    zw = np.random.normal(size=K * (K + 1) // 2 + (N - K) * K)
    zv = np.random.normal(size=(M, K * T))
    np.random.seed(9)
    model._resample_W(data)
    model._resample_V(data)
    ost = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in dict(st, nu2=nu2).items()}
    Yk = orc.binomial_kappa(Ys, Ntr, nu2)
    # (compat="exact": every row / column with its own weights - the stale-weight quirks have their own fixtures)
    W = ost["W"].copy()
    Vf = ost["V"].reshape(-1, K)
    zpos = 0
    for i in range(N):
        d = min(i + 1, K)
        obs = ~np.isnan(Yk[i].reshape(-1))
        c = np.where(obs, 1.0 / nu2[i].reshape(-1), 0.0)
        y = np.where(obs, Yk[i].reshape(-1), 0.0)
        Q = (Vf[:, :d] * c[:, None]).T @ Vf[:, :d] + np.eye(d) / ost["sigma2"]
        m = (Vf[:, :d] * c[:, None]).T @ y
        L = np.linalg.cholesky(Q)
        W[i, :d] = np.linalg.solve(Q, m) + np.linalg.solve(L.T, zw[zpos:zpos + d])
        zpos += d
    assert relerr(model.W, W) < 1e-10
    ost["W"] = W
    orc.v_step(ost, Yk, orc.trend_penalty(T, tf), z=zv, compat="exact", perm=orc.perm_from_order(model.v_order(), K, T))
    assert relerr(model.V, ost["V"]) < 1e-8


# ---- checkpoint / resume ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("family,rng", [("gaussian", "device"), ("gaussian", "host"), ("binomial", "device"), ("negbinom", "device")])
def test_checkpoint_restore_continues_the_chain_bit_for_bit(tmp_path, family, rng):
    """model.checkpoint() -> np.savez -> a fresh model .restore(np.load(...)): the next sweeps equal the uninterrupted
    chain's exactly (device draw counter / legacy numpy state travel with the factors and the horseshoe+ levels)."""
    from functionalmf_amd.factor import (GaussianBayesianTensorFiltering, BinomialBayesianTensorFiltering,
                                         NegativeBinomialBayesianTensorFiltering)
    rs = np.random.RandomState(77)
    N, M, T, R, K = 13, 9, 12, 2, 3
    Wt = rs.normal(size=(N, K))
    Vt = 0.3 * np.cumsum(rs.normal(size=(M, T, K)), axis=1)
    Mu = np.einsum("nk,mtk->nmt", Wt, Vt)
    P = 1 / (1 + np.exp(-Mu))
    if family == "gaussian":
        data = Mu[..., None] + rs.normal(0, 0.5, size=(N, M, T, R))
        data[:2, :2] = np.nan
        make = lambda: GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=1, sigma2_init=0.5, lam2_init=0.1,
                                                       nu2_init=1.0, rng=rng, device_seed=3)
    elif family == "binomial":
        data = (rs.binomial(5, P).astype(float), np.full((N, M, T), 5.0))
        make = lambda: BinomialBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=1, sigma2_init=0.5, lam2_init=0.1,
                                                       rng=rng, device_seed=3)
    else:
        data = rs.negative_binomial(4.0, 1 - np.repeat(P[..., None], R, axis=-1)).astype(float)
        make = lambda: NegativeBinomialBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=1, sigma2_init=0.5, lam2_init=0.1,
                                                               nmetropolis=4, rng=rng, device_seed=3)
    np.random.seed(5)
    a = make()
    for _ in range(4):
        a.resample(data)
    np.savez(tmp_path / "ck.npz", **a.checkpoint())
    for _ in range(3):
        a.resample(data)
    a.sync()
    ref = dict(W=a.W.copy(), V=a.V.copy(), Tau2=np.array(a.Tau2).copy(), sigma2=float(a.sigma2), lam2=float(a.lam2))
    np.random.seed(999)                     # whatever happened to the global generator in between
    b = make()
    b.restore(dict(np.load(tmp_path / "ck.npz")))
    for _ in range(3):
        b.resample(data)
    b.sync()
    assert np.array_equal(b.W, ref["W"]) and np.array_equal(b.V, ref["V"])
    assert np.array_equal(np.array(b.Tau2), ref["Tau2"])
    assert float(b.sigma2) == ref["sigma2"] and float(b.lam2) == ref["lam2"]
    if family == "negbinom":
        assert np.array_equal(np.asarray(a.R), np.asarray(b.R))



def test_log_link_exponential_is_accurate_to_double_precision():
    """The log link's exp comes from a 128-entry table + a degree-5 polynomial (csrc/btf_ess.h: exp_tab).  With linear
    predictors that depend on the row alone, a row's log-likelihood is M T (y eta_i - exp(eta_i)): one exponential
    per row (btf_gass_begin with the log link, btf_gass_grid: cur_ll), against numpy."""
    from functionalmf_amd.factor import ConstrainedNonconjugateBayesianTensorFiltering
    from functionalmf_amd import _native
    N, M, T, K, y = 96, 3, 8, 2, 2.0
    rs = np.random.RandomState(9)
    eta = np.concatenate([rs.uniform(-30, 30, size=N - 16), rs.uniform(-1e-3, 1e-3, size=8), np.log(2) / 128 * (np.arange(8) + 0.5)])
    W = np.zeros((N, K)); W[:, 0] = eta
    V = np.zeros((M, T, K)); V[..., 0] = 1.0; V[..., 1] = 0.5
    Cons = np.concatenate([np.eye(T), np.full((T, 1), -1e6)], axis=1)
    model = ConstrainedNonconjugateBayesianTensorFiltering(N, M, T, "poisson_identity", Cons, gass_ngrid=16, nembeds=K, tf_order=0,
                                                           sigma2_init=1.0, lam2_init=0.1, W_init=W, V_init=V,
                                                           Tau2_init=np.ones((M, T)), sampler="banded")
    Y = np.full((N, M, T, 1), y)
    model._bind_data(Y)
    model._push_state()
    model._ctx.call("btf_gass_set_constraints", _native.dptr(model._cons), int(model._cons.shape[0]), None, 0)
    z = rs.normal(size=K * (K + 1) // 2 + (N - K) * K)
    u = rs.rand(N)
    model._ctx.call("btf_gass_begin", 0, 0, _native.dptr(z), _native.dptr(u), 1, 1e-6, 0, 0)        # link 0: log
    info = np.zeros((N, 2), dtype=np.int32)
    cur = np.empty(N)
    model._ctx.call("btf_gass_grid", 0, info.ctypes.data_as(_native._c_ip), None, None, _native.dptr(cur))
    ref = M * T * (y * eta - np.exp(eta))
    tol = 4e-16 * M * T * (np.exp(eta) + np.abs(y * eta)) + 1e-15 * np.abs(ref)
    assert np.all(np.abs(cur - ref) <= tol), (np.abs(cur - ref) / tol).max()


@pytest.mark.parametrize("noutliers", [0, 40])
def test_negbinom_per_row_rates_mh_loop_in_one_launch_equals_the_stepwise_loop(monkeypatch, noutliers):
    """rdims = (1,2) (one rate per row, examples/negbinom_tensor_filtering.py): the rows' chains do not interact, so
    btf_nb_mh runs the whole loop in one launch, a workgroup per row; BTF_NB_MH_STEPWISE=1 keeps the two launches per
    step.  Same Philox streams, same sums in the same order: the chains of rates are identical."""
    from functionalmf_amd.factor import NegativeBinomialBayesianTensorFiltering
    rs = np.random.RandomState(35)
    N, M, T, Rr, K = 37, 9, 11, 2, 3
    Wt = 0.7 * rs.normal(size=(N, K))
    Vt = 0.3 * np.cumsum(rs.normal(size=(M, T, K)), axis=1)
    P = 1 / (1 + np.exp(-np.einsum("nk,mtk->nmt", Wt, Vt)))
    data = rs.negative_binomial(4.0, 1 - P[..., None].repeat(Rr, -1)).astype(float)
    flat = data.reshape(-1)
    if noutliers:
        flat[rs.choice(flat.size, noutliers, replace=False)] += 1500.0 + rs.randint(0, 700, size=noutliers)
    data[0, :2] = np.nan
    chains = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("BTF_NB_MH_STEPWISE", mode)
        np.random.seed(4)
        model = NegativeBinomialBayesianTensorFiltering(N, M, T, nembeds=K, W_true=Wt, V_true=Vt, sigma2_true=1.0,
                                                        lam2_true=0.1, Tau2_true=np.ones((M, 3 * T - 1)),
                                                        rdims=(1, 2), nmetropolis=7, rng="device", device_seed=11)
        model._bind_data(data)
        out = []
        for _ in range(12):
            model._resample_R(data)
            out.append(np.asarray(model.R).reshape(-1).copy())
        assert getattr(model, "_mh_on_device", True)
        chains[mode] = np.array(out)
    assert len(set(np.round(chains["0"][:, 3], 9))) > 3          # the chains move
    assert np.array_equal(chains["0"], chains["1"])


def test_negbinom_single_rate_partial_sum_launch_with_wide_count_range(monkeypatch):
    """The per-step partial-sum launch of the single-rate MH loop with counts across the whole 1024-entry table (several
    workgroups take the suffix-sum form) and a few thousand counts beyond it (their chunks spread over the workgroups):
    the chain of rates equals the per-row loop's."""
    from functionalmf_amd.factor import NegativeBinomialBayesianTensorFiltering
    rs = np.random.RandomState(41)
    N, M, T, Rr, K = 64, 24, 16, 2, 3
    Wt = 0.9 * rs.normal(size=(N, K))
    Vt = 0.5 * np.cumsum(rs.normal(size=(M, T, K)), axis=1)
    eta = np.clip(np.einsum("nk,mtk->nmt", Wt, Vt), -6, 6.5)
    P = 1 / (1 + np.exp(-eta))
    data = rs.negative_binomial(3.0, 1 - P[..., None].repeat(Rr, -1)).astype(float)
    nout = int((data >= 1024).sum())
    assert 256 < nout < data.size // 8 and data[data < 1024].max() > 768, (nout, data[data < 1024].max())
    chains = {}
    for mode in ("0", "1", "2"):     # 0: one launch per step (decision merged into the next partial-sum launch); 2: two; 1: per row
        monkeypatch.setenv("BTF_NB_MH_STEPWISE", mode)
        np.random.seed(4)
        model = NegativeBinomialBayesianTensorFiltering(N, M, T, nembeds=K, W_true=Wt, V_true=Vt, sigma2_true=1.0,
                                                        lam2_true=0.1, Tau2_true=np.ones((M, 3 * T - 1)),
                                                        rdims=(0, 1, 2), nmetropolis=6, rng="device", device_seed=12)
        model._bind_data(data)
        out = []
        for _ in range(20):
            model._resample_R(data)
            out.append(float(np.asarray(model.R).reshape(-1)[0]))
        chains[mode] = np.array(out)
    assert len(set(np.round(chains["0"], 9))) > 3
    np.testing.assert_allclose(chains["0"], chains["1"], rtol=1e-9)
    assert np.array_equal(chains["0"], chains["2"])          # same sums in the same order
