"""The oracle (oracle/btf_oracle.py) against the fixtures captured from the real
reference code (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
from conftest import state_from, relerr
from oracle import btf_oracle as orc


def test_delta_matches_reference(golden):
    g = golden("g0_delta.npz")
    for key, ref in g.items():
        _, T, k = key.split("_")
        D = orc.trend_penalty(int(T[1:]), int(k[1:]))
        assert D.shape == ref.shape
        assert np.array_equal(D, ref), key


FIX = ["g1_c1_heldout.npz", "g2_c2_complete.npz", "g3_partial_reps.npz", "g5_illcond.npz"]


@pytest.mark.parametrize("name", FIX)
def test_w_step_gaussian(golden, name):
    g = golden(name)
    st = state_from(g, "s0_")
    np.random.seed(0)
    W = orc.w_step(st, g["Y"], z=g["z_W"])
    assert relerr(W, g["W_after"]) < 1e-11


@pytest.mark.parametrize("name", FIX)
def test_w_step_consumes_legacy_stream_in_row_order(golden, name):
    g = golden(name)
    st = state_from(g, "s0_")
    seed = {"g1": 100, "g2": 300, "g3": 400, "g5": 600}[name[:2]]
    np.random.seed(seed)
    W = orc.w_step(st, g["Y"])
    assert relerr(W, g["W_after"]) < 1e-11


@pytest.mark.parametrize("name", FIX)
@pytest.mark.parametrize("perm", ["depth", "ident", "twist"])
def test_v_step_gaussian(golden, name, perm):
    g = golden(name)
    N, M, T, R, K, tf = g["dims"]
    st = state_from(g, "s0_")
    st["W"] = g["W_after"].copy()
    Delta = orc.trend_penalty(int(T), int(tf))
    info = {}
    V = orc.v_step(st, g["Y"], Delta, perm={"depth": "depth", "ident": "identity", "twist": "twist"}[perm],
                   z=g["z_V"], info=info)
    # two fp64 factorisations of the same system agree to ~cond(Q)*eps (hard part 2)
    # G5 is deliberately extreme (cond(Q) up to 1.2e13): it pins control flow, not digits
    tol = 1e-4 if name.startswith("g5") else 1e-6
    assert relerr(V, g["V_after_" + perm]) < tol
    assert np.array_equal(info["attempts"], g["V_tries_" + perm])


@pytest.mark.parametrize("name", ["g1_c1_heldout.npz", "g3_partial_reps.npz", "g5_illcond.npz"])
def test_v_system_assembly(golden, name):
    """Precision matrix and mean-part exactly as the reference assembled them
    (captured on entry to the factorisation; no shim arithmetic involved)."""
    g = golden(name)
    N, M, T, R, K, tf = g["dims"]
    st = state_from(g, "s0_")
    st["W"] = g["W_after"].copy()
    Delta = orc.trend_penalty(int(T), int(tf))
    st["_cnt"], st["_ybar"] = orc.replicate_stats(g["Y"])
    src = orc.stale_column_sources(st["_ybar"])
    for j in range(int(M)):
        Q, mu = orc.v_step_system(st, g["Y"], Delta, j, int(src[j]))
        assert relerr(Q, g["V_Q"][j]) < 1e-13
        assert relerr(mu, g["V_mu"][j]) < 1e-12


def test_stale_column_quirk_matters(golden):
    """Q2: with partially missing replicates the correct weights give a different
    answer; the oracle must follow the reference, not the textbook."""
    g = golden("g3_partial_reps.npz")
    N, M, T, R, K, tf = g["dims"]
    st = state_from(g, "s0_")
    st["W"] = g["W_after"].copy()
    Delta = orc.trend_penalty(int(T), int(tf))
    V = orc.v_step(st, g["Y"], Delta, z=g["z_V"], compat="exact")
    assert relerr(V, g["V_after_depth"]) > 1e-4


@pytest.mark.parametrize("tag", ["nan", "full"])
def test_binomial_steps_given_omega(golden, tag):
    g = golden("g4_binomial_%s.npz" % tag)
    N, M, T, R, K, tf = g["dims"]
    st = state_from(g, "s0_")
    assert relerr(st["nu2"][~np.isnan(g["Ysucc"])], (1 / g["omega"])[~np.isnan(g["Ysucc"])]) < 1e-15
    W = orc.binomial_w_step(st, g["Ysucc"], g["Ntrials"], z=g["z_W"])
    assert relerr(W, g["W_after"]) < 1e-11
    Delta = orc.trend_penalty(int(T), int(tf))
    for perm, nm in (("depth", "depth"), ("identity", "ident"), ("twist", "twist")):
        st2 = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in st.items()}
        V = orc.binomial_v_step(st2, g["Ysucc"], g["Ntrials"], Delta, perm=perm, z=g["z_V"])
        assert relerr(V, g["V_after_" + nm]) < 1e-10


def test_jitter_retry(golden):
    g = golden("g5_illcond.npz")
    N, M, T, R, K, tf = g["dims"]
    st = state_from(g, "retry_s0_")
    Delta = orc.trend_penalty(int(T), int(tf))
    info = {}
    V = orc.v_step(st, g["Y"], Delta, z=g["retry_z_V"], info=info)
    assert np.array_equal(info["attempts"], g["retry_tries"])
    assert info["attempts"][2] == 2
    assert relerr(V, g["retry_V_after"]) < 1e-7     # min eigenvalue ~6e-6: cond ~1e9
    st = state_from(g, "retry_s0_")
    V = orc.v_step(st, g["Y"], Delta, z=g["retry_z_V"], info=info, perm="twist")
    assert np.array_equal(info["attempts"], g["retry_tries_twist"])
    assert relerr(V, g["retry_V_after_twist"]) < 1e-7


def test_hyper_steps(golden):
    g = golden("g1_c1_heldout.npz")
    N, M, T, R, K, tf = g["dims"]
    Delta = orc.trend_penalty(int(T), int(tf))
    st = state_from(g, "h0_")
    sse, n = orc.sse_and_count(st, g["Y"])
    assert abs(sse - g["h_sse"]) / g["h_sse"] < 1e-13 and n == int(g["h_nobs"])
    np.random.seed(200)
    assert abs(orc.nu2_step(st, g["Y"]) - g["h_nu2"]) / g["h_nu2"] < 1e-13
    np.random.seed(201)
    assert abs(orc.sigma2_step(st) - g["h_sigma2"]) / g["h_sigma2"] < 1e-13
    np.random.seed(202)
    orc.tau2_step(st, Delta)
    for k in ("Tau2", "Tau2_a", "Tau2_b", "Tau2_c"):
        assert relerr(st[k], g["h_tau_" + k]) < 1e-12, k
    np.random.seed(203)
    orc.lam2_step(st, Delta)
    assert abs(st["lam2"] - g["h_lam2"]) / g["h_lam2"] < 1e-12
    assert abs(st["lam2_a"] - g["h_lam2_a"]) / g["h_lam2_a"] < 1e-12


def test_construction_draws(golden):
    g = golden("g1_c1_heldout.npz")
    N, M, T, R, K, tf = [int(x) for x in g["dims"]]
    np.random.seed(11)
    st, Delta = orc.init_state(N, M, T, K=K, tf_order=tf, sigma2_init=0.5, lam2_init=0.1, nu2_init=1.0)
    for k in ("W", "Tau2", "Tau2_a", "Tau2_b", "Tau2_c"):
        assert relerr(st[k], g["init_" + k]) < 1e-13, k
    assert relerr(st["V"], g["init_V"]) < 1e-8          # prior precision is near-singular scale-wise
    assert relerr(st["lam2_a"], g["init_lam2_a"]) < 1e-13


def test_run_gibbs_chain(golden):
    """11 free-running sweeps from the same seed: chains are chaotic, so this only
    holds because every step above matches to ~1e-12."""
    g = golden("g6_c1_chain.npz")
    Y = g["Y"]
    N, M, T, R = Y.shape
    K = g["init_W"].shape[1]
    np.random.seed(21)
    st, Delta = orc.init_state(N, M, T, K=K, tf_order=2, sigma2_init=0.5, lam2_init=0.1, nu2_init=1.0)
    np.random.seed(22)
    st0 = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in st.items()}
    res = orc.run_gibbs(st, Y, Delta, nburn=3, nthin=2, nsamples=4)
    for k in ("W", "V", "sigma2", "lam2", "Tau2", "nu2"):
        assert res[k].shape == g["res_" + k].shape, k
        assert relerr(res[k], g["res_" + k]) < 1e-5, k
    np.random.seed(22)
    res = orc.run_gibbs(st0, Y, Delta, nburn=3, nthin=2, nsamples=4, perm="twist")
    for k in ("W", "V", "sigma2", "lam2", "Tau2", "nu2"):
        assert relerr(res[k], g["rest_" + k]) < 1e-5, k


def test_pg_series_sampler_moments():
    rng = np.random.default_rng(7)
    for b, c in ((1, 0.0), (2, 0.5), (4, 2.0), (10, 8.0)):
        x = orc.pg_draw_series(b, c, 20000, rng)
        m, v = orc.pg_mean(b, c), orc.pg_var(b, c)
        assert abs(x.mean() - m) < 5 * np.sqrt(v / x.size)
        assert abs(x.var() - v) / v < 0.08


def test_pg_series_sampler_for_many_cells_has_the_moments_of_each_cell():
    """pg_draw_series_cells (the CPU stand-in for pgdrawv that bench.py's Binomial cpu_baseline times): one tilt per cell,
    the truncated tail replaced by its mean in closed form - mean and variance per group of equal tilts against the
    Polya-Gamma moments (Polson, Scott & Windle 2013, sec. 2.3)."""
    rng = np.random.RandomState(3)
    tilts = np.array([0.0, 0.3, 1.5, 4.0, 9.0])
    n = 20000
    for b in (1.0, 4.0):
        x = orc.pg_draw_series_cells(b, np.repeat(tilts, n), rng).reshape(len(tilts), n)
        for row, c in zip(x, tilts):
            m, v = orc.pg_mean(b, c), orc.pg_var(b, c)
            assert abs(row.mean() - m) < 5 * np.sqrt(v / n), (b, c, row.mean(), m)
            assert abs(row.var() - v) / v < 0.08, (b, c)
    assert np.all(orc.pg_draw_series_cells(4.0, np.array([-2.0, 2.0]), np.random.RandomState(1)) > 0)     # (even in the tilt)


def test_strong_cpu_path_equals_reference_faithful_path(golden):
    """The vectorised / banded-LAPACK CPU baseline computes the same draws (depth-major order)."""
    g = golden("g2_c2_complete.npz")
    N, M, T, R, K, tf = [int(x) for x in g["dims"]]
    Delta = orc.trend_penalty(T, tf)
    Rr, ybar = orc.hoisted_stats(g["Y"])
    st = state_from(g, "s0_")
    W = orc.w_step_strong(st, Rr, ybar, z=g["z_W"])
    assert relerr(W, g["W_after"]) < 1e-11
    V = orc.v_step_strong(st, Rr, ybar, Delta, z=g["z_V"])
    assert relerr(V, g["V_after_depth"]) < 1e-6


NB_TAGS = ["scalar", "rows", "cells", "cols_depth"]


@pytest.mark.parametrize("tag", NB_TAGS)
def test_negbinom_rate_update(golden, tag):
    """the random-walk MH update of the Negative-Binomial rate R
    (factor.py:513-554) replayed from the same legacy-RNG seed must walk the same path."""
    g = golden("g7_negbinom_%s.npz" % tag)
    rdims = tuple(int(d) for d in g["rdims"])
    st = state_from(g, "s0_")
    st["R"] = g["R_before"].copy()
    assert st["R"].shape == orc.nb_rate_shape(g["data"].shape[:3], rdims)
    np.random.seed(int(g["seed_R"]))
    Y, Ntr = orc.nb_resample_rate(st, g["data"], rdims=rdims)
    assert relerr(st["R"], g["R_after"]) < 1e-12
    assert relerr(Ntr, g["N_after"]) < 1e-12
    assert np.array_equal(np.isnan(Y), np.all(np.isnan(g["data"]), axis=-1))


@pytest.mark.parametrize("tag", ["scalar", "rows"])
def test_negbinom_full_sweep_given_omega(golden, tag):
    """One whole NB sweep of the reference (R update, then the Binomial sweep on the pseudo-data
    Y = sum of counts, N = sum of counts + R) given the Polya-Gamma draws."""
    g = golden("g7_negbinom_%s.npz" % tag)
    N, M, T, R, K, tf = [int(x) for x in g["dims"]]
    rdims = tuple(int(d) for d in g["rdims"])
    st = state_from(g, "s0_")
    st["R"] = g["R_before"].copy()
    Delta = orc.trend_penalty(T, tf)
    np.random.seed(int(g["seed_full"]))
    Y, Ntr = orc.nb_resample_rate(st, g["data"], rdims=rdims)
    assert relerr(st["R"], g["full_R"]) < 1e-12 and relerr(Ntr, g["full_N"]) < 1e-12
    with np.errstate(divide="ignore"):
        st["nu2"] = 1 / g["omega"]                       # factor.py:455-460 through the PG stand-in
    orc.sigma2_step(st)
    orc.tau2_step(st, Delta)
    orc.lam2_step(st, Delta)
    orc.binomial_w_step(st, Y, Ntr)
    orc.binomial_v_step(st, Y, Ntr, Delta, perm="twist")
    for k, tol in (("sigma2", 1e-12), ("lam2", 1e-12), ("Tau2", 1e-10), ("W", 1e-10), ("V", 1e-8)):
        assert relerr(st[k], g["full_" + k]) < tol, k


# ---- spectral V sampler: reference code under the spectral square-root shim (make_golden_spectral.py) ----
def _spectral_case(golden, tag):
    g8 = golden("g8_spectral.npz")
    if tag == "g2":
        g2 = golden("g2_c2_complete.npz")
        st = state_from(g2, "s0_")
        st["W"] = g2["W_after"].copy()
        return g2["Y"], st, [int(x) for x in g2["dims"]], g2["z_V"], g8["g2_V_after_spectral"]
    st = {k: (float(g8["%s_s0_%s" % (tag, k)]) if k in ("lam2", "sigma2", "nu2") else g8["%s_s0_%s" % (tag, k)].copy())
          for k in ("W", "V", "Tau2", "lam2", "sigma2", "nu2")}
    return g8[tag + "_Y"], st, [int(x) for x in g8[tag + "_dims"]], g8[tag + "_z_V"], g8[tag + "_V_after_spectral"]


@pytest.mark.parametrize("tag", ["g2", "k5", "tf0", "tf1", "tf3", "short", "held"])
def test_v_step_spectral_square_root_vs_reference(golden, tag):
    """factor.py:364-409 + fast_mvn.py:35-47 run by the reference itself with the spectral shim: the
    oracle's restatement (faithful per-column assembly and the vectorised strong path) must reproduce it,
    and its mean term must be the order-invariant Q^-1 mu of the depth-major draw."""
    Y, st, (N, M, T, R, K, tf), z, Vref = _spectral_case(golden, tag)
    Delta = orc.trend_penalty(T, tf)
    a = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in st.items()}
    tol = 1e-6 if tag == "g2" else 1e-9        # g2: lam2 on its floor after two sweeps, cond(Q) ~ 1e9 (cond * eps)
    orc.v_step(a, Y, Delta, perm="spectral", z=z)
    assert relerr(a["V"], Vref) < tol
    if tag != "held":           # (held: whole curves missing - every column its own K x K block; the strong path is complete-data only)
        b = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in st.items()}
        Rr, ybar = orc.hoisted_stats(Y)
        orc.v_step_strong(b, Rr, ybar, Delta, z=z, order="spectral")
        assert relerr(b["V"], Vref) < tol
    c = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in st.items()}
    d = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in st.items()}
    orc.v_step(c, Y, Delta, perm="spectral", z=0 * z)
    orc.v_step(d, Y, Delta, perm="depth", z=0 * z)
    assert relerr(c["V"], d["V"]) < tol


@pytest.mark.parametrize("tag", ["g2", "k5", "tf0", "tf1", "tf3", "short", "held"])
def test_spectral_mean_is_the_dense_lapack_mean_of_the_references_system(golden, tag):
    """G8's *_V_mean_dense is np.linalg.solve(Q_j, mu_part_j) on the precision and mu_part the REFERENCE's _resample_V
    assembled (make_golden_spectral.py records them at fast_mvn.py:47); the generator already asserts its spectral shim
    against that solve.  Here: the oracle's spectral square root at z = 0 hits the same means, column by column, to
    50 cond(Q_j) eps."""
    g8 = golden("g8_spectral.npz")
    Y, st, (N, M, T, R, K, tf), z, _ = _spectral_case(golden, tag)
    dense, cond = g8[tag + "_V_mean_dense"], g8[tag + "_cond"]
    assert dense.shape == (M, T, K) and cond.shape == (M,)
    a = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in st.items()}
    orc.v_step(a, Y, orc.trend_penalty(T, tf), perm="spectral", z=0 * z)
    for j in range(M):
        err = np.abs(a["V"][j] - dense[j]).max() / np.abs(dense[j]).max()
        assert err <= 50 * cond[j] * np.finfo(float).eps, (tag, j, err, cond[j])


def test_spectral_square_root_has_the_right_covariance(golden):
    """S S' = Q^-1 for the spectral square root (unit normals pick out the columns of S)."""
    Y, st, (N, M, T, R, K, tf), z, _ = _spectral_case(golden, "k5")
    Delta = orc.trend_penalty(T, tf)
    st["_cnt"], st["_ybar"] = orc.replicate_stats(Y)
    Q, mu = orc.v_step_system(st, Y, Delta, 1, 1)
    S = np.stack([orc.mvn_from_precision_spectral(Q, K, T, z=e) for e in np.eye(K * T)], axis=1)
    assert np.abs(S @ S.T @ Q - np.eye(K * T)).max() < 1e-10


def test_spectral_jitter_retry_matches_reference_schedule(golden):
    g5, g8 = golden("g5_illcond.npz"), golden("g8_spectral.npz")
    N, M, T, R, K, tf = [int(x) for x in g5["dims"]]
    st = state_from(g5, "retry_s0_")
    info = {}
    orc.v_step(st, g5["Y"], orc.trend_penalty(T, tf), perm="spectral", z=g5["retry_z_V"], info=info)
    assert np.array_equal(info["attempts"], g8["g5_retry_tries_spectral"])
    assert relerr(st["V"], g8["g5_retry_V_after_spectral"]) < 1e-5


# ---- slice samplers and dense fast_mvn branches: fixtures from the reference itself (make_golden_ess.py) ----
def test_elliptical_slice_vs_reference(golden):
    g = golden("g9_ess.npz")
    Sigma, target = g["ess_Sigma"], g["ess_target"]
    L = np.linalg.cholesky(Sigma)
    D = target.size

    def bump(x, args):
        return -0.5 * np.sum((x - args) ** 2) / 0.3
    x = g["ess_x0"].copy()
    for rep in range(6):
        np.random.seed(700 + rep)
        prior = L if rep % 2 == 0 else L @ np.random.normal(size=D)
        np.random.seed(800 + rep)
        info = {}
        x, ll = orc.elliptical_slice(x, prior, bump, ll_args=target, mu=None if rep < 4 else 0.2 * target,
                                     angle_range=0 if rep != 3 else 1.5, info=info)
        assert relerr(x, g["ess_out"][rep]) < 1e-12 and abs(ll - g["ess_ll"][rep]) < 1e-10
        assert info["evaluations"] == g["ess_nev"][rep]


def test_dense_mvn_branches_vs_reference(golden):
    g = golden("g9_ess.npz")
    S, mu = g["mvn_S"], g["mvn_mu"]
    kws = (dict(precision=True), dict(precision=True, mu_part=mu), dict(precision=True, mu=mu), dict(precision=False),
           dict(precision=False, mu=mu), dict(precision=False, mu_part=mu), dict(precision=False, chol_factor=True),
           dict(precision=True, chol_factor=True, mu_part=mu))
    for i, kw in enumerate(kws):
        Q = np.linalg.cholesky(S) if kw.get("chol_factor") else S
        np.random.seed(900 + i)
        assert relerr(orc.sample_mvn_dense(Q, **kw), g["mvn_out"][i]) < 1e-12, kw
    np.random.seed(950)
    assert relerr(orc.sample_mvn_dense(0.7, mu=mu), g["mvn_out"][8]) < 1e-12
    np.random.seed(951)
    assert relerr(orc.sample_mvn_dense(np.full(mu.size, 2.5), mu_part=mu, precision=True), g["mvn_out"][9]) < 1e-12


def test_gass_vs_reference(golden):
    g = golden("g9_ess.npz")
    Sig, C, tgt = g["gass_Sigma"], g["gass_C"], g["gass_target"]

    def gll(x, args):
        x = np.atleast_2d(x)
        r = -0.5 * np.sum((x - args) ** 2, axis=1) / 0.5
        return r if r.shape[0] > 1 else r[0]
    x = g["gass_x"][0].copy()
    for rep in range(6):
        np.random.seed(1000 + rep)
        if rep % 2 == 0:
            x, ll = orc.gass(x, lambda: orc.sample_mvn_dense(Sig, mu=np.zeros_like(x)), gll, C, ll_args=tgt, ngrid=50)
        else:
            x, ll = orc.gass(x, lambda: orc.sample_mvn_dense(np.linalg.inv(Sig), mu=np.zeros_like(x), precision=True), gll, C,
                             ll_args=tgt, mu=0.1 * tgt, ngrid=50)
        assert relerr(x, g["gass_x"][rep + 1]) < 1e-10 and abs(ll - g["gass_ll"][rep]) < 1e-9
        assert np.all(C[:, :-1] @ x >= C[:, -1] - 1e-12)


def _nc_case(golden, link):
    g = golden("g9_ess.npz")
    tag = "nc_%s_" % link
    st = {k: (float(g[tag + "s0_" + k]) if k in ("lam2", "sigma2") else g[tag + "s0_" + k].copy())
          for k in ("W", "V", "Tau2", "lam2", "sigma2")}
    return g, tag, st, [int(x) for x in g[tag + "dims"]]


@pytest.mark.parametrize("link", ["log", "identity"])
def test_nonconjugate_joint_slice_steps_vs_reference(golden, link):
    """NonconjugateBayesianTensorFiltering._resample_W / _resample_V (factor.py:567-590) run by the reference with
    a Poisson likelihood callback: the oracle's restatement must land on the same states after the same number
    of likelihood evaluations, from the same legacy-RNG seeds."""
    g, tag, st, (N, M, T, R, K, tf) = _nc_case(golden, link)
    Y = g[tag + "Y"]
    Delta = orc.trend_penalty(T, tf)
    info = {}
    np.random.seed(1100)
    orc.nonconjugate_w_step(st, Y, link=link, info=info)
    assert relerr(st["W"], g[tag + "W_after"]) < 1e-12 and info["evaluations"] == int(g[tag + "W_nev"])
    np.random.seed(1200)
    orc.nonconjugate_v_step(st, Y, Delta, link=link, perm="twist", info=info)
    assert relerr(st["V"], g[tag + "V_after"]) < 1e-9 and info["evaluations"] == int(g[tag + "V_nev"])


def student_t_tanh_loglik(W, V, data):
    """The arbitrary callback of fixture g9 case (e) (tests/golden/make_golden_ess.py): Student-t(4) residuals of scale 0.3
    around 3 tanh(w.v) - none of the build's device likelihoods."""
    mu = 3.0 * np.tanh(np.einsum("nk,mtk->nmt", W, V))[..., None]
    r = (data - mu) / 0.3
    return float(np.sum(np.where(np.isnan(data), 0.0, -2.5 * np.log1p(r * r / 4.0))))


def _cb_case(golden):
    g = golden("g9_ess.npz")
    st = {k: (float(g["cb_s0_" + k]) if k in ("lam2", "sigma2") else g["cb_s0_" + k].copy()) for k in ("W", "V", "Tau2", "lam2", "sigma2")}
    return g, st, [int(x) for x in g["cb_dims"]]


def test_nonconjugate_steps_with_an_arbitrary_callback_vs_reference(golden):
    """The reference's NonconjugateBayesianTensorFiltering run with an arbitrary Python log-likelihood (factor.py:567-612):
    three W / V slices in a row - the oracle, handed the same function, lands on the same states after the same numbers of
    evaluations."""
    g, st, (N, M, T, R, K, tf) = _cb_case(golden)
    Y = g["cb_Y"]
    Delta = orc.trend_penalty(T, tf)
    info = {}
    for sweep in range(3):
        np.random.seed(1300 + 2 * sweep)
        orc.nonconjugate_w_step(st, Y, link=student_t_tanh_loglik, info=info)
        assert relerr(st["W"], g["cb_W_chain"][sweep]) < 1e-12 and info["evaluations"] == int(g["cb_nev"][2 * sweep])
        np.random.seed(1301 + 2 * sweep)
        orc.nonconjugate_v_step(st, Y, Delta, link=student_t_tanh_loglik, perm="twist", info=info)
        assert relerr(st["V"], g["cb_V_chain"][sweep]) < 1e-9 and info["evaluations"] == int(g["cb_nev"][2 * sweep + 1])


@pytest.mark.parametrize("family", ["bernoulli_logit", "gaussian", "negbin_logit"])
def test_nonconjugate_steps_with_other_likelihoods_vs_reference(golden, family):
    """The reference's NonconjugateBayesianTensorFiltering run with scipy.stats callbacks for a Bernoulli-logit, a
    Gaussian and a Negative-Binomial-logit likelihood (tests/golden/make_golden_lik.py): the oracle's family_loglik equals
    the callback's value, and its slice steps land on the reference's states after the same number of evaluations."""
    g = golden("g11_likelihoods.npz")
    tag = "lk_%s_" % family
    N, M, T, R, K, tf = [int(x) for x in g[tag + "dims"]]
    par = None if np.isnan(g[tag + "param"]) else float(g[tag + "param"])
    st = {k: (float(g[tag + "s0_" + k]) if k in ("lam2", "sigma2") else g[tag + "s0_" + k].copy()) for k in ("W", "V", "Tau2", "lam2", "sigma2")}
    Y = g[tag + "Y"]
    ll0 = orc.family_loglik(st["W"], st["V"], Y, family, par)
    assert abs(ll0 - float(g[tag + "ll0"])) < 1e-10 * abs(ll0)
    Delta = orc.trend_penalty(T, tf)
    info = {}
    np.random.seed(int(g[tag + "seeds"][0]))
    orc.nonconjugate_w_step(st, Y, link=family, info=info, param=par)
    assert relerr(st["W"], g[tag + "W_after"]) < 1e-12 and info["evaluations"] == int(g[tag + "W_nev"])
    np.random.seed(int(g[tag + "seeds"][1]))
    orc.nonconjugate_v_step(st, Y, Delta, link=family, perm="twist", info=info, param=par)
    assert relerr(st["V"], g[tag + "V_after"]) < 1e-9 and info["evaluations"] == int(g[tag + "V_nev"])


# ---- constrained non-conjugate model: the reference's own worker functions (make_golden_gass.py) ----
def _gass_case(golden):
    g = golden("g10_gass.npz")
    N, M, T, R, K, tf = [int(x) for x in g["dims"]]
    st = {k: (float(g["s0_" + k]) if k in ("lam2", "sigma2") else g["s0_" + k].copy()) for k in ("W", "V", "Tau2", "lam2", "sigma2")}
    return g, st, (N, M, T, R, K, tf)


def test_constrained_row_and_column_updates_vs_reference(golden):
    """_resample_W_i / _resample_V_j (factor.py:665-855) run by the reference itself, row by row and column by column,
    each from its own seed: the oracle's restatement with the same per-chain streams must land on the same states."""
    g, st, (N, M, T, R, K, tf) = _gass_case(golden)
    ngrid = int(g["ngrid"])
    a = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in st.items()}
    info = {}
    orc.constrained_w_step(a, g["Y"], g["Cons"], link="identity", ngrid=ngrid,
                           rngs=[np.random.RandomState(2000 + i) for i in range(N)], Row_constraints=g["Row_constraints"], info=info)
    assert relerr(a["W"], g["W_after"]) < 1e-12
    assert min(info["grid"]) > 0 and max(info["accepted"]) > 0
    b = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in st.items()}
    orc.constrained_v_step(b, g["Y"], orc.trend_penalty(T, tf), g["Cons"], link="identity", ngrid=ngrid, perm="twist",
                           rngs=[np.random.RandomState(3000 + j) for j in range(M)])
    assert relerr(b["V"], g["V_after"]) < 1e-9


def test_gass_valid_grid_is_the_intersection_of_the_constraint_arcs():
    """Every returned grid angle satisfies all constraints on the ellipse, every dropped one violates one (up to the
    eps margin of gass.py:47)."""
    rs = np.random.RandomState(2)
    D = 4
    A = rs.normal(size=(30, D))
    x0 = np.abs(rs.normal(size=D)) + 0.2
    c = A @ x0 - np.abs(rs.normal(size=30)) * 0.5           # x0 strictly feasible
    v = rs.normal(size=D)
    grid, restricted = orc.gass_valid_grid(x0, v, A, c)
    assert restricted and 0 < len(grid) < 10000
    full = np.linspace(-np.pi, np.pi, 10000)
    slack = (A @ (x0[None] * np.cos(full[:, None]) + v[None] * np.sin(full[:, None])).T - c[:, None]).min(axis=0)
    assert np.all(slack[np.isin(full, grid)] >= -1e-9)
    assert np.all(slack[~np.isin(full, grid)] < 1e-4)
