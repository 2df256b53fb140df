"""Parity at BASELINE.json's full headline size (512,256,64,4) K=5: the half-sweeps against the
vectorised CPU oracle (validated against the reference fixtures in test_oracle_golden.py) from
identical state and normals, plus size-independent properties (linearity of the conditional
means in the data, order-invariance of the mean term).  Needs an MI355X."""
import numpy as np
import pytest
from conftest import relerr

pytestmark = pytest.mark.gpu

N, M, T, R, K = 512, 256, 64, 4, 5


@pytest.fixture(scope="module")
def c3():
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    from oracle import btf_oracle as orc
    rs = np.random.RandomState(1)
    Wt = rs.normal(size=(N, K))
    Wt[np.triu_indices(K, 1)] = 0
    Vt = 0.1 * np.cumsum(rs.normal(size=(M, T, K)), axis=1)
    Y = np.einsum("nk,mtk->nmt", Wt, Vt)[..., None] + rs.normal(0, 0.5, size=(N, M, T, R))
    np.random.seed(2)
    model = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, sigma2_init=0.5, lam2_init=0.1, nu2_init=1.0)
    for _ in range(2):                       # leave the prior draw (host-RNG sweeps)
        model.resample(Y)
    Delta = orc.trend_penalty(T, 2)
    Rr, ybar = orc.hoisted_stats(Y)
    return model, Y, Delta, Rr, ybar


def snapshot(model):
    return dict(W=model.W.copy(), V=model.V.copy(), Tau2=np.array(model.Tau2, float).copy(), lam2=float(model.lam2),
                sigma2=float(model.sigma2), nu2=float(model.nu2))


def test_full_size_half_sweeps_vs_cpu(c3):
    from oracle import btf_oracle as orc
    model, Y, Delta, Rr, ybar = c3
    from functionalmf_amd import _native
    model._ctx.call("btf_set_option", _native.OPT_SAMPLER, _native.SAMPLERS["chain"])   # depth-major single chain: the CPU path's order
    st = snapshot(model)
    np.random.seed(7)
    zw = np.random.normal(size=K * (K + 1) // 2 + (N - K) * K)
    zv = np.random.normal(size=(M, K * T))
    np.random.seed(7)
    model._resample_W(Y)
    model._resample_V(Y)
    orc.w_step_strong(st, Rr, ybar, z=zw)
    assert relerr(model.W, st["W"]) < 1e-10
    # the V half-sweep is checked from identical inputs: the ~1e-12 the two W's differ by is amplified by the
    # conditioning of the column systems like any other rounding
    st["W"] = model.W.copy()
    Vgpu = model.V.copy()
    orc.v_step_strong(st, Rr, ybar, Delta, z=zv)
    model._ctx.call("btf_set_option", _native.OPT_SAMPLER, _native.SAMPLERS["banded"])
    _assert_columns_within_conditioning(Vgpu, st, Rr, ybar, Delta)


def _assert_columns_within_conditioning(Vgpu, st, Rr, ybar, Delta, factor=50.0):
    """Two correct fp64 solves of Q_j x = b agree to ~cond(Q_j) eps.  Column by column: the relative difference is held
    to `factor` * cond(Q_j) * eps with cond(Q_j) MEASURED for that column (its extreme eigenvalues), and to the
    north-star tolerance 1e-5 over the whole factor.  st["V"]: the oracle's result."""
    from oracle import btf_oracle as orc
    eps = np.finfo(float).eps
    conds = orc.v_column_conds(st, Rr, ybar, Delta)
    err = np.abs(Vgpu - st["V"]).reshape(M, -1).max(axis=1) / np.abs(st["V"]).reshape(M, -1).max(axis=1)
    worst = int(np.argmax(err / conds))
    print("V columns: max rel err %.2e, cond range %.1e .. %.1e, worst err / (cond eps) = %.1f (column %d)"
          % (err.max(), conds.min(), conds.max(), err[worst] / (conds[worst] * eps), worst))
    assert (err < factor * conds * eps).all(), (err[worst], conds[worst])
    assert relerr(Vgpu, st["V"]) < 1e-5


def test_full_size_spectral_half_sweep_vs_cpu(c3):
    """The bench's own instantiation - C3 complete data, accum<5,0,16> -> w_solve<5,false,8> -> v_spectral<3> - with
    host normals against the oracle's spectral square root (v_step_strong(order="spectral"): the declared
    (U (x) Pi') blockdiag(L_k^-T D_k^-1/2) z of include/btf.h) from identical state and normals."""
    from oracle import btf_oracle as orc
    from functionalmf_amd import _native
    model, Y, Delta, Rr, ybar = c3
    model._ctx.call("btf_set_option", _native.OPT_SAMPLER, _native.SAMPLERS["spectral"])
    st = snapshot(model)
    np.random.seed(17)
    zw = np.random.normal(size=K * (K + 1) // 2 + (N - K) * K)
    zv = np.random.normal(size=(M, K * T))
    np.random.seed(17)
    model._resample_W(Y)
    model._resample_V(Y)
    assert model.v_sampler() == "spectral"
    orc.w_step_strong(st, Rr, ybar, z=zw)
    assert relerr(model.W, st["W"]) < 1e-10
    st["W"] = model.W.copy()
    Vgpu = model.V.copy()
    orc.v_step_strong(st, Rr, ybar, Delta, z=zv, order="spectral")
    model._ctx.call("btf_set_option", _native.OPT_SAMPLER, _native.SAMPLERS["banded"])
    _assert_columns_within_conditioning(Vgpu, st, Rr, ybar, Delta)      # (cond(Q_j) reaches 6e11 in this state: 1.6e-6 seen)


def test_full_size_device_rng_draws_are_white(c3):
    """rng="device" at C3 size (the benchmarked mode: Philox normals drawn inside the spectral kernel): for 16 columns
    spread over the tensor, C_j' (x - Q_j^-1 mu_j) with C_j C_j' = Q_j must be standard normal - pooled over 150 draws:
    mean, variance, fourth moment, KS, and the per-coordinate variances."""
    from scipy import stats
    from oracle import btf_oracle as orc
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    model0, Y, Delta, Rr, ybar = c3
    st = snapshot(model0)
    model = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, sigma2_init=st["sigma2"], lam2_init=st["lam2"],
                                            nu2_init=st["nu2"], W_init=st["W"], V_init=st["V"], Tau2_init=st["Tau2"],
                                            rng="device", device_seed=11)
    cols = list(range(0, M, 16))
    Cs, means = [], []
    for j in cols:
        Q, mu = orc.v_column_system_strong(st, Rr, ybar, Delta, j)
        Cs.append(np.linalg.cholesky(Q))
        means.append(np.linalg.solve(Q, mu))
    reps, n = 150, K * T
    res = np.empty((reps, len(cols), n))
    for r in range(reps):
        model._resample_V(Y)
        Vd = model.V.reshape(M, n)
        for c, j in enumerate(cols):
            res[r, c] = Cs[c].T @ (Vd[j] - means[c])
    assert model.v_sampler() == "spectral"
    x = res.reshape(-1)
    assert abs(x.mean()) < 5 / np.sqrt(x.size)
    assert abs(x.var() - 1) < 5 * np.sqrt(2 / x.size)
    assert abs((x ** 4).mean() - 3) < 5 * np.sqrt(96 / x.size)
    assert stats.kstest(x[::5], "norm").pvalue > 1e-3
    assert np.abs(res.var(axis=0) - 1).max() < 6.5 * np.sqrt(2 / reps)
    del model


def test_full_size_mean_term_is_order_invariant_and_matches_cpu(c3, monkeypatch):
    """z = 0: V = Q^-1 mu does not depend on the elimination order - the default twisted kernel
    must agree with the depth-major CPU solve."""
    from oracle import btf_oracle as orc
    model, Y, Delta, Rr, ybar = c3
    st = snapshot(model)
    monkeypatch.setattr(model, "_v_normals", lambda: np.zeros((M, K * T)))
    model._resample_V(Y)
    Vgpu = model.V.copy()
    orc.v_step_strong(st, Rr, ybar, Delta, z=np.zeros((M, K * T)))
    # (after two sweeps lam2 sits on its 1e-5 floor and Tau2 spans 1e-7..2e6: two correct fp64 solves in
    #  different elimination orders agree to cond*eps ~ 1e-6 here - scripts/p4err.py: twisted vs CPU 1.2e-6,
    #  generic GPU kernel vs CPU 0.6e-6, panelised vs unpanelised twisted kernel 5e-10): the bound is the measured
    #  conditioning of each column
    _assert_columns_within_conditioning(Vgpu, st, Rr, ybar, Delta)


def test_full_size_conditional_means_are_linear_in_the_data(c3, monkeypatch):
    """With the noise switched off the W and V updates are linear maps of the observations."""
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    model, Y, Delta, Rr, ybar = c3
    st = snapshot(model)
    rs = np.random.RandomState(3)
    Y2 = rs.normal(size=Y.shape)
    outs = []
    for data in (Y, Y2, Y + 2.0 * Y2):
        m = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, sigma2_init=st["sigma2"], lam2_init=st["lam2"],
                                            nu2_init=st["nu2"], W_init=st["W"], V_init=st["V"], Tau2_init=st["Tau2"])
        monkeypatch.setattr(m, "_w_normals", lambda: np.zeros(K * (K + 1) // 2 + (N - K) * K))
        monkeypatch.setattr(m, "_v_normals", lambda: np.zeros((M, K * T)))
        m._resample_W(data)
        Wd = m.W.copy()
        m.W = st["W"]                          # V step from the common W
        m._resample_V(data)
        outs.append((Wd, m.V.copy()))
        del m
    assert relerr(outs[2][0], outs[0][0] + 2.0 * outs[1][0]) < 1e-10
    assert relerr(outs[2][1], outs[0][1] + 2.0 * outs[1][1]) < 1e-7


def test_full_size_sse_matches_cpu(c3):
    from oracle import btf_oracle as orc
    model, Y, Delta, Rr, ybar = c3
    import ctypes
    st = snapshot(model)
    model._push_state()
    sse, nobs = ctypes.c_double(), ctypes.c_double()
    model._ctx.call("btf_sse", ctypes.byref(sse), ctypes.byref(nobs))
    ref, n = orc.sse_and_count(st, Y)
    assert abs(sse.value - ref) / ref < 1e-11 and nobs.value == n


# ---- config C4: Binomial (512,256,64), 4 trials per cell, K=5 - PG draw + weighted half-sweeps at full size ----
@pytest.fixture(scope="module")
def c4():
    from functionalmf_amd.factor import BinomialBayesianTensorFiltering
    from bench import synth_rows, synth_V
    Vt = synth_V(1, M, T, K)
    _, Wt = synth_rows(1, range(N), M, T, 1, K, Vt, noise=0.0)
    Mu = np.einsum("nk,mtk->nmt", Wt, Vt)
    rs = np.random.RandomState(7)
    Nt = np.full((N, M, T), 4.0)
    Ys = rs.binomial(4, 1 / (1 + np.exp(-Mu))).astype(float)
    return Ys, Nt


@pytest.mark.parametrize("compat,pg_exact", [("reference", None), ("exact", None), ("exact", "series")])
def test_c4_binomial_full_size_pg_draw_and_weighted_half_sweeps(c4, compat, pg_exact):
    """factor.py:437-460 at BASELINE config 4, with the default Polya-Gamma sampler (four exact Devroye draws per cell,
    as pypolyagamma) and with the opt-in series.  The device Polya-Gamma draw is checked against the closed-form
    moments over all 8.4 M cells, then - GIVEN that omega (btf_get_omega) - the weighted W and V half-sweeps
    against the oracle from identical state and normals (compat="reference": rows >= K reuse row K-1's weights
    and every column reuses column 0's, quirks Q1/Q2; the oracle's W step always carries Q1)."""
    from functionalmf_amd.factor import BinomialBayesianTensorFiltering
    from oracle import btf_oracle as orc
    Ys, Nt = c4
    np.random.seed(3)
    # (local scales of moderate size: under the raw horseshoe+ prior draw, Tau2 down to 1e-9, two correct fp64
    #  factorisations in different orders agree only to cond * eps ~ 5e-6 - see the C3 mean-term test above)
    tau0 = np.random.RandomState(4).gamma(2.0, 0.5, size=(M, 3 * T - 1))
    model = BinomialBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, sigma2_init=0.5, lam2_init=0.1, compat=compat,
                                            Tau2_init=tau0, pg_exact=pg_exact)
    model.V = 0.5 * model.V                                    # keep |psi| moderate under the prior draw
    W0, V0 = model.W.copy(), model.V.copy()
    model._resample_nu2((Ys, Nt))
    omega = 1.0 / model.nu2
    psi = np.einsum("nk,mtk->nmt", W0, V0)
    zs = (omega - orc.pg_mean(4.0, psi)) / np.sqrt(orc.pg_var(4.0, psi))
    n = zs.size
    assert abs(zs.mean()) < 5 / np.sqrt(n), zs.mean()
    assert abs((zs ** 2).mean() - 1) < 1e-2, (zs ** 2).mean()
    assert (omega > 0).all()
    st = dict(W=W0.copy(), V=V0.copy(), Tau2=np.array(model.Tau2, float).copy(), lam2=float(model.lam2),
              sigma2=float(model.sigma2), nu2=1.0 / omega)
    Delta = orc.trend_penalty(T, 2)
    np.random.seed(11)
    zw = np.random.normal(size=K * (K + 1) // 2 + (N - K) * K)
    zv = np.random.normal(size=(M, K * T))
    np.random.seed(11)
    model._resample_W((Ys, Nt))
    if compat == "reference":
        orc.binomial_w_step(st, Ys, Nt, z=zw)
        assert relerr(model.W, st["W"]) < 1e-10
    else:
        st["W"] = model.W.copy()
    model._resample_V((Ys, Nt))
    assert model.v_sampler() == "banded"
    orc.binomial_v_step(st, Ys, Nt, Delta, z=zv, compat=compat, perm=orc.perm_from_order(model.v_order(), K, T))
    assert relerr(model.V, st["V"]) < 1e-6


# ---- config C5: (4096,1024,64,4) K=8 row/column-sharded over 8 GPUs - one rank's slabs on one GPU ----
C5 = dict(N=4096, M=1024, T=64, R=4, K=8, world=8)


@pytest.mark.parametrize("rank", [0, 5])
@pytest.mark.parametrize("sampler", ["banded", "spectral"])
def test_c5_rank_slab_half_sweeps(rank, sampler):
    """BASELINE config 5 as rank `rank` of 8 sees it: btf_set_shard with a 512-row slab (512,1024,64,4) for the W
    half-sweep and a 128-column slab (4096,128,64,4) for the V half-sweep, through the C ABI, against
    w_step_strong / v_step_strong on exactly those blocks from identical state and normals."""
    import ctypes as C
    from functionalmf_amd import _native
    from oracle import btf_oracle as orc
    from bench import synth_rows, synth_V
    N_, M_, T_, R_, K_, world = (C5[k] for k in ("N", "M", "T", "R", "K", "world"))
    nl, ml = N_ // world, M_ // world
    row0, col0 = rank * nl, rank * ml
    Vt = synth_V(1, M_, T_, K_)
    rows, _ = synth_rows(1, range(row0, row0 + nl), M_, T_, R_, K_, Vt)                 # (nl, M, T, R)
    cols, Wt = synth_rows(2, range(N_), ml, T_, R_, K_, Vt[col0:col0 + ml])             # (N, ml, T, R)
    rs = np.random.RandomState(9)
    W = Wt + 0.05 * rs.normal(size=Wt.shape)
    W[np.triu_indices(K_, 1)] = 0
    V = Vt + 0.05 * rs.normal(size=Vt.shape)
    Delta = orc.trend_penalty(T_, 2)
    nD = Delta.shape[0]
    Tau2 = rs.gamma(2.0, 0.5, size=(M_, nD))
    lam2, sigma2, nu2 = 0.1, 0.5, 0.3
    ctx = _native.Context(N_, M_, T_, K_, 2)
    ctx.call("btf_set_shard", row0, nl, col0, ml)
    ctx.call("btf_set_option", _native.OPT_SAMPLER, _native.SAMPLERS[sampler])
    ctx.call("btf_set_data_gaussian", _native.dptr(rows), _native.dptr(cols), R_)
    ctx.call("btf_set_W", _native.dptr(W))
    ctx.call("btf_set_V", _native.dptr(V))
    ctx.call("btf_set_hyper", _native.dptr(Tau2), lam2, sigma2)
    ctx.call("btf_set_nu2", nu2)
    nzw = K_ * (K_ + 1) // 2 + (N_ - K_) * K_
    zw = rs.normal(size=nzw)
    zv = rs.normal(size=(M_, K_ * T_))
    # ---- W half-sweep on the row slab
    ctx.call("btf_resample_W", _native.dptr(zw), 1, _native.COMPAT["exact"])
    Wg = np.empty_like(W)
    ctx.call("btf_get_W", _native.dptr(Wg))
    assert np.array_equal(np.delete(Wg, np.s_[row0:row0 + nl], axis=0), np.delete(W, np.s_[row0:row0 + nl], axis=0))
    ybar_r = rows.mean(axis=3)
    st = dict(W=W[row0:row0 + nl].copy(), V=V.copy(), nu2=nu2, sigma2=sigma2)
    z0 = int(orc_w_offset(row0, K_))
    orc.w_step_strong(st, R_, ybar_r, z=zw[z0:int(orc_w_offset(row0 + nl, K_))], row0=row0)
    assert relerr(Wg[row0:row0 + nl], st["W"]) < 1e-10
    # ---- V half-sweep on the column slab (from the ORIGINAL W: restore it)
    ctx.call("btf_set_W", _native.dptr(W))
    ctx.call("btf_resample_V", _native.dptr(zv), 2, _native.COMPAT["exact"], 1e-6, 4)
    Vg = np.empty_like(V)
    ctx.call("btf_get_V", _native.dptr(Vg))
    assert np.array_equal(np.delete(Vg, np.s_[col0:col0 + ml], axis=0), np.delete(V, np.s_[col0:col0 + ml], axis=0))
    which = C.c_int32()
    ctx.call("btf_get_V_sampler", C.byref(which))
    assert which.value == _native.SAMPLERS[sampler]
    order = np.zeros(K_ * T_, dtype=np.int32)
    ctx.call("btf_get_V_order", order.ctypes.data_as(_native._c_ip))
    ybar_c = cols.mean(axis=3)
    stv = dict(W=W.copy(), V=V[col0:col0 + ml].copy(), Tau2=Tau2[col0:col0 + ml], lam2=lam2, nu2=nu2)
    orc.v_step_strong(stv, R_, ybar_c, Delta, z=zv[col0:col0 + ml], order="spectral" if sampler == "spectral" else order)
    assert relerr(Vg[col0:col0 + ml], stv["V"]) < 1e-6
    ctx.close()


def orc_w_offset(i, K_):
    return i * (i + 1) // 2 if i < K_ else K_ * (K_ + 1) // 2 + (i - K_) * K_


@pytest.mark.parametrize("sampler", ["spectral", "banded"])
def test_full_size_held_out_curves_vs_weighted_conditionals(sampler):
    """(512,256,64,4) K=5 with the reference examples' held-out block and 3 % more whole curves missing: the
    curve-counts form (complete-data kernels + per-row / per-column corrections) against the weighted conditionals of
    factor.py:343-346 / :388-391 - every row of W, a spread of columns of V (deficient and complete ones), and the
    residual sum of squares, from identical state and normals."""
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    from oracle import btf_oracle as orc
    rs = np.random.RandomState(4)
    Wt = rs.normal(size=(N, K))
    Wt[np.triu_indices(K, 1)] = 0
    Vt = 0.1 * np.cumsum(rs.normal(size=(M, T, K)), axis=1)
    Y = np.einsum("nk,mtk->nmt", Wt, Vt)[..., None] + rs.normal(0, 0.5, size=(N, M, T, R))
    Y[:3, :3] = np.nan
    Y[rs.rand(N, M) < 0.03] = np.nan
    Y[rs.rand(N, M) < 0.02, :, 1] = np.nan            # thinned curves: one replicate gone at every depth
    st = dict(W=Wt + 0.1 * rs.normal(size=Wt.shape), V=Vt + 0.05 * rs.normal(size=Vt.shape), Tau2=rs.gamma(2.0, 0.5, size=(M, 3 * T - 1)),
              lam2=0.2, sigma2=0.6, nu2=0.3)
    st["W"][np.triu_indices(K, 1)] = 0
    model = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, sigma2_init=st["sigma2"], lam2_init=st["lam2"],
                                            nu2_init=st["nu2"], W_init=st["W"], V_init=st["V"], Tau2_init=st["Tau2"],
                                            compat="exact", sampler=sampler)
    np.random.seed(3)
    zw = np.random.normal(size=K * (K + 1) // 2 + (N - K) * K)
    zv = np.random.normal(size=(M, K * T))
    np.random.seed(3)
    model._resample_W(Y)
    assert model.likelihood_form() == "curve_counts" and model.v_sampler() == sampler
    # W: vectorised weighted conditional, row by row from the per-cell counts
    cnt = np.sum(~np.isnan(Y), axis=3).astype(float)                 # (N,M,T)
    S1 = np.nansum(Y, axis=3)
    Vf = st["V"].reshape(-1, K)
    c = cnt.reshape(N, -1) / st["nu2"]
    m = (S1.reshape(N, -1) / st["nu2"]) @ Vf                          # (N,K)
    W = st["W"].copy()
    zpos = 0
    for i in range(N):
        d = min(i + 1, K)
        Q = (Vf[:, :d] * c[i][:, None]).T @ Vf[:, :d] + np.eye(d) / st["sigma2"]
        L = np.linalg.cholesky(Q)
        W[i, :d] = np.linalg.solve(Q, m[i, :d]) + np.linalg.solve(L.T, zw[zpos:zpos + d])
        zpos += d
    assert relerr(model.W, W) < 1e-10
    # SSE identity with the corrected Grams (the nu2 draw of a device sweep uses it): through the public statistic
    np.random.seed(8)
    model._resample_nu2(Y)
    ost = dict(st, W=W)
    sse, n = orc.sse_and_count(ost, Y)
    np.random.seed(8)
    ref = 1.0 / np.random.gamma(0.1 + n / 2.0, 1.0 / (0.1 + sse / 2.0))
    assert abs(model.nu2 - ref) / ref < 1e-9
    model.nu2 = st["nu2"]
    # V: a spread of columns
    np.random.seed(5)
    model._v_normals = lambda: zv
    model._resample_V(Y)
    cols = [0, 1, 2, 3, 17, 100, 255]
    ost = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in dict(st, W=W).items()}
    perm = "spectral" if sampler == "spectral" else orc.perm_from_order(model.v_order(), K, T)
    orc.v_step(ost, Y, orc.trend_penalty(T, 2), perm=perm, z=zv, compat="exact", cols=cols)
    assert relerr(model.V[cols], ost["V"][cols]) < 1e-6


# ---- K = 10 (a configuration the reference runs: flutrends/benchmark.py:33, `for nembeds in [5, 10]`) at C3 size ----
@pytest.mark.parametrize("variant", ["complete", "missing"])
def test_full_size_k10_half_sweeps(variant):
    """(512,256,64,4) with nembeds = 10: the accumulation kernels of K = 10 (complete data: 8-wave workgroups; weighted:
    one output per lane, 65 accumulators - the instances that spilled hundreds of VGPRs before round 3), w_solve
    with 10 x 10 systems and the any-bandwidth V sampler (half-bandwidth 30).  W: every row against the weighted
    conditional from the per-cell counts; V: a spread of columns against the oracle in the kernel's declared order."""
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    from oracle import btf_oracle as orc
    K10 = 10
    rs = np.random.RandomState(21)
    Wt = rs.normal(size=(N, K10))
    Wt[np.triu_indices(K10, 1)] = 0
    Vt = 0.1 * np.cumsum(rs.normal(size=(M, T, K10)), axis=1)
    Y = np.einsum("nk,mtk->nmt", Wt, Vt)[..., None] + rs.normal(0, 0.5, size=(N, M, T, R))
    if variant == "missing":
        Y[rs.rand(N, M) < 0.05] = np.nan
        Y[rs.rand(N, M, T, R) < 0.05] = np.nan
    st = dict(W=Wt + 0.1 * rs.normal(size=Wt.shape), V=Vt + 0.05 * rs.normal(size=Vt.shape), Tau2=rs.gamma(2.0, 0.5, size=(M, 3 * T - 1)),
              lam2=0.2, sigma2=0.6, nu2=0.3)
    st["W"][np.triu_indices(K10, 1)] = 0
    model = GaussianBayesianTensorFiltering(N, M, T, nembeds=K10, tf_order=2, sigma2_init=st["sigma2"], lam2_init=st["lam2"],
                                            nu2_init=st["nu2"], W_init=st["W"], V_init=st["V"], Tau2_init=st["Tau2"],
                                            compat="exact", sampler="banded")
    np.random.seed(3)
    zw = np.random.normal(size=K10 * (K10 + 1) // 2 + (N - K10) * K10)
    zv = np.random.normal(size=(M, K10 * T))
    np.random.seed(3)
    model._resample_W(Y)
    assert model.likelihood_form() == ("weighted" if variant == "missing" else "complete")
    cnt = np.sum(~np.isnan(Y), axis=3).astype(float)
    S1 = np.nansum(Y, axis=3)
    Vf = st["V"].reshape(-1, K10)
    c = cnt.reshape(N, -1) / st["nu2"]
    m = (S1.reshape(N, -1) / st["nu2"]) @ Vf
    W = st["W"].copy()
    zpos = 0
    for i in range(N):
        d = min(i + 1, K10)
        Q = (Vf[:, :d] * c[i][:, None]).T @ Vf[:, :d] + np.eye(d) / st["sigma2"]
        L = np.linalg.cholesky(Q)
        W[i, :d] = np.linalg.solve(Q, m[i, :d]) + np.linalg.solve(L.T, zw[zpos:zpos + d])
        zpos += d
    assert relerr(model.W, W) < 1e-10
    np.random.seed(5)
    model._v_normals = lambda: zv
    model._resample_V(Y)
    cols = [0, 1, 17, 100, 255]
    ost = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in dict(st, W=W).items()}
    orc.v_step(ost, Y, orc.trend_penalty(T, 2), perm=orc.perm_from_order(model.v_order(), K10, T), z=zv, compat="exact", cols=cols)
    assert relerr(model.V[cols], ost["V"][cols]) < 1e-6
