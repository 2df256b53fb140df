"""Parity against the oracle over the supported shape space: every nembeds 1..10 (register-
chain kernel with DPP shifts for half-bandwidth <= 15, with shuffles above, generic kernel
for tiny bandwidths), tf_order 0..3, 3-D data, fewer rows than embeddings, missing data,
odd sizes that exercise the padding.  Needs an MI355X."""
import numpy as np
import pytest
from conftest import relerr

pytestmark = pytest.mark.gpu


def make_case(N, M, T, R, K, tf, missing, seed):
    rs = np.random.RandomState(seed)
    Wt = rs.normal(size=(N, K))
    Vt = 0.3 * np.cumsum(rs.normal(size=(M, T, K)), axis=1)
    Y = np.einsum("nk,mtk->nmt", Wt, Vt)[..., None] + rs.normal(0, 0.7, size=(N, M, T, R))
    if missing == "curves":                 # counts constant along the depth axis: the curve-counts form
        Y[rs.rand(N, M) < 0.15] = np.nan
        Y[rs.rand(N, M) < 0.1, :, 0] = np.nan
        Y[0, min(1, M - 1)] = np.nan
    elif missing:
        Y[rs.rand(N, M, T, R) < 0.2] = np.nan
        Y[0, min(1, M - 1)] = np.nan
    if R == 1 and seed % 2:
        Y = Y[..., 0]                       # 3-D input: one replicate implied (factor.py:323-324)
    nD = {0: T, 1: 2 * T, 2: 3 * T - 1, 3: 4 * T - 1}[tf]
    st = dict(W=rs.normal(size=(N, K)), V=0.2 * rs.normal(size=(M, T, K)),
              Tau2=np.exp(rs.normal(size=(M, nD))), lam2=0.3, sigma2=0.8, nu2=0.6)
    if N > 1:
        st["W"][np.triu_indices(N, 1, K)] = 0
    return Y, st


CASES = [
    # N,  M,  T, R, K, tf, missing
    (9, 4, 7, 2, 1, 2, False),
    (9, 4, 7, 2, 1, 0, True),       # half-bandwidth 1: generic kernel
    (11, 5, 9, 1, 2, 0, False),     # half-bandwidth 2: generic kernel
    (11, 5, 9, 1, 2, 2, True),
    (13, 3, 8, 3, 3, 1, False),
    (70, 3, 6, 2, 4, 2, True),      # > 64 rows: two w_solve workgroups
    (16, 6, 10, 2, 5, 2, False),    # bw 15: DPP path, headline shape
    (16, 6, 10, 2, 5, 3, False),    # bw 20: shuffle path
    (12, 4, 8, 2, 6, 2, True),      # bw 18
    (12, 3, 9, 1, 7, 1, False),
    (20, 3, 8, 2, 8, 2, False),     # bw 24 (config C5's embedding size)
    (14, 2, 7, 2, 9, 2, True),      # bw 27
    (14, 2, 6, 2, 10, 2, False),    # bw 30
    (3, 4, 8, 2, 5, 2, False),      # fewer rows than embeddings
    (1, 3, 6, 2, 2, 1, False),      # a single row
    (130, 67, 5, 1, 3, 2, True),    # odd sizes across the 128-wide tiles
    (6, 3, 2, 2, 1, 2, False),      # shortest depth axis the reference accepts, one embedding
    (6, 3, 2, 2, 5, 2, False),      # T = 2 with a band wider than the whole system
    (7, 2, 3, 1, 2, 1, True),
    (7, 2, 4, 2, 5, 2, False),
    (9, 2, 8, 2, 3, 2, False),      # T = 2(tf+1)+2: smallest depth the twisted kernel takes
    (9, 2, 7, 2, 3, 2, False),      # one less: single chain
    (600, 2, 6, 2, 3, 1, False),    # w_solve with 16 rows per workgroup
    (1100, 2, 6, 1, 5, 2, True),    # ... 32, weighted
    (2100, 3, 5, 2, 2, 1, False),   # ... 64
    (40, 9, 12, 3, 5, 2, "curves"), # whole / thinned curves missing: complete-data kernels plus corrections
    (600, 5, 8, 2, 3, 1, "curves"),
    (23, 70, 9, 2, 8, 2, "curves"), # more deficient columns than one side workgroup takes; K = 8
    (20, 3, 64, 2, 8, 2, True),     # K = 8 at C3's depth, weighted: the twisted layout fits only without the staged likelihood blocks
    (20, 3, 64, 2, 7, 2, True),     # ... K = 7: fits with them
    (31, 6, 9, 1, 4, 2, "curves"),  # 3-D input: counts 0 / 1
]


@pytest.mark.parametrize("variant", ["banded", "chain", "banded_nopanel", "spectral"])
@pytest.mark.parametrize("N,M,T,R,K,tf,missing", CASES)
def test_half_sweeps_match_oracle(N, M, T, R, K, tf, missing, variant):
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    from oracle import btf_oracle as orc
    Y, st = make_case(N, M, T, R, K, tf, missing, seed=N * 131 + K * 7 + tf)
    model = GaussianBayesianTensorFiltering(
        N, M, T, nembeds=K, tf_order=tf, sigma2_init=st["sigma2"], lam2_init=st["lam2"], nu2_init=st["nu2"],
        W_init=st["W"], V_init=st["V"], Tau2_init=st["Tau2"], sampler=variant)
    assert model.Delta.shape[0] == st["Tau2"].shape[1]
    Delta = orc.trend_penalty(T, tf)
    nzw = sum(min(i + 1, K) for i in range(N))
    np.random.seed(5)
    zw = np.random.normal(size=nzw)
    zv = np.random.normal(size=(M, K * T))
    ost = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in st.items()}
    np.random.seed(5)
    model._resample_W(Y)
    model._resample_V(Y)
    orc.w_step(ost, Y, z=zw)
    assert relerr(model.W, ost["W"]) < 1e-10
    spectral = model.v_sampler() == "spectral"
    # weighted data falls back to the banded sampler; curve counts keep the spectral one
    assert spectral == (variant == "spectral" and missing in (False, "curves"))
    if missing == "curves":
        assert model.likelihood_form() == ("curve_counts" if variant in ("banded", "banded_nopanel", "spectral") else "weighted")
    orc.v_step(ost, Y, Delta, z=zv, perm="spectral" if spectral else orc.perm_from_order(model.v_order(), K, T))
    assert relerr(model.V, ost["V"]) < 1e-8
    # nu2 statistics on the new state
    np.random.seed(6)
    model._resample_nu2(Y)
    sse, n = orc.sse_and_count(ost, Y)
    np.random.seed(6)
    ref = 1.0 / np.random.gamma(0.1 + n / 2.0, 1.0 / (0.1 + sse / 2.0))
    assert abs(model.nu2 - ref) / ref < 1e-10


def test_all_missing_column_and_row_are_prior_draws():
    """A column without any observation is a pure prior draw; a row without any is N(0, sigma2)."""
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    from oracle import btf_oracle as orc
    Y, st = make_case(10, 5, 8, 2, 3, 2, False, seed=3)
    Y[:, 2] = np.nan
    Y[7] = np.nan
    model = GaussianBayesianTensorFiltering(10, 5, 8, nembeds=3, tf_order=2, sigma2_init=0.8, lam2_init=0.3,
                                            nu2_init=0.6, W_init=st["W"], V_init=st["V"], Tau2_init=st["Tau2"])
    np.random.seed(1)
    zw = np.random.normal(size=sum(min(i + 1, 3) for i in range(10)))
    zv = np.random.normal(size=(5, 24))
    np.random.seed(1)
    model._resample_W(Y)
    model._resample_V(Y)
    ost = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in st.items()}
    orc.w_step(ost, Y, z=zw)
    orc.v_step(ost, Y, orc.trend_penalty(8, 2), z=zv, perm=orc.perm_from_order(model.v_order(), 3, 8))
    assert relerr(model.W, ost["W"]) < 1e-10 and relerr(model.V, ost["V"]) < 1e-8
    off = sum(min(i + 1, 3) for i in range(7))
    assert np.allclose(model.W[7], np.sqrt(0.8) * zw[off:off + 3], rtol=1e-12)


def test_rejects_unsupported_shapes():
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    from functionalmf_amd._native import BTFError
    with pytest.raises(BTFError):
        GaussianBayesianTensorFiltering(8, 4, 6, nembeds=11, tf_order=2)
    with pytest.raises(BTFError):
        GaussianBayesianTensorFiltering(8, 4, 6, nembeds=3, tf_order=4)


@pytest.mark.parametrize("S,N,M,T,K,transform", [(37, 5, 4, 6, 3, None), (200, 9, 7, 5, 2, "ilogit"),
                                                  (1000, 3, 5, 11, 5, None), (1025, 2, 3, 4, 1, "square"),
                                                  (2, 4, 3, 2, 4, None), (1, 3, 2, 2, 2, "ilogit")])
def test_posterior_summary_matches_numpy(S, N, M, T, K, transform):
    """mean / percentiles over kept samples against the reference scripts' host
    computation (einsum + mean + np.percentile, examples/gaussian_tensor_filtering.py:82-85)."""
    from functionalmf_amd.utils import posterior_summary
    rs = np.random.RandomState(S + N)
    Ws = rs.normal(size=(S, N, K))
    Vs = rs.normal(size=(S, M, T, K))
    q = (5, 50, 95, 0, 100, 33.3)
    mean, quant = posterior_summary(Ws, Vs, q=q, transform=transform)
    Mu = np.einsum("znk,zmtk->znmt", Ws, Vs)
    if transform == "ilogit":
        Mu = 1 / (1 + np.exp(-Mu))
    elif transform == "square":
        Mu = Mu ** 2
    assert np.max(np.abs(mean - Mu.mean(0))) < 1e-12 * max(1.0, np.abs(Mu).max())
    ref = np.percentile(Mu, q, axis=0)
    assert np.max(np.abs(quant - ref)) < 1e-12 * max(1.0, np.abs(Mu).max())


@pytest.mark.parametrize("variant", ["banded", "banded_nopanel", "chain", "spectral"])
def test_bw15_indefinite_column_is_reported_and_jitter_recovers(variant):
    """K = 5, tf = 2 (bw = 15: the panelised MFMA factorisation records a bad pivot instead of branching on
    it): a grossly indefinite column must still end in NotPositiveDefiniteError with its index, and a
    mildly indefinite one must be rescued by the jitter retries exactly like the unpanelised kernel."""
    from functionalmf_amd._native import NotPositiveDefiniteError
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    N, M, T, R, K = 16, 6, 12, 2, 5
    Y, st = make_case(N, M, T, R, K, 2, False, seed=5)
    def build():
        m = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, sigma2_init=st["sigma2"], lam2_init=st["lam2"],
                                            nu2_init=st["nu2"], W_init=st["W"], V_init=st["V"], Tau2_init=st["Tau2"].copy(),
                                            sampler=variant)
        return m
    m = build()
    m.Tau2[3, 9] = -1e-3
    np.random.seed(0)
    m._resample_V(Y)
    with pytest.raises(NotPositiveDefiniteError) as e:
        m.sync()
    assert e.value.index == 3
    # a well-posed problem right after the failure: the context must be usable again
    m.Tau2 = st["Tau2"].copy()
    np.random.seed(0)
    m._resample_V(Y)
    m.sync()
    from functionalmf_amd import _native
    ref = build()
    ref._ctx.call("btf_set_option", _native.OPT_SAMPLER, _native.SAMPLERS["generic"])
    np.random.seed(0)
    ref._resample_V(Y)
    order_dependent = variant != "chain"             # other square roots differ from depth-major in the noise term
    if not order_dependent:
        assert relerr(m.V, ref.V) < 1e-8
    assert np.isfinite(m.V).all()


@pytest.mark.parametrize("sampler", ["banded", "generic"])
@pytest.mark.parametrize("weighted", [False, True])
def test_long_depth_axis_band_in_hbm_scratch(weighted, sampler):
    """K*T = 3700 (the size of the reference's flu-trends application: flutrends/, 370 weeks x 10 embeddings):
    the block-banded factor does not fit the 160 KB of LDS.  Default: the chunked chain (btf_banded_chunk.h) - the band is
    assembled and eliminated ~200 columns at a time in LDS, the finished factor columns parked in HBM; sampler="generic":
    the any-size kernel with the whole band in HBM scratch.  One column block against the oracle from identical state and
    normals, both ways."""
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    from oracle import btf_oracle as orc
    N, M, T, R, K, tf = 12, 2, 370, 2, 10, 2
    Y, st = make_case(N, M, T, R, K, tf, weighted, seed=77)
    model = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=tf, sigma2_init=st["sigma2"], lam2_init=st["lam2"],
                                            nu2_init=st["nu2"], W_init=st["W"], V_init=st["V"], Tau2_init=st["Tau2"], sampler=sampler)
    np.random.seed(5)
    zv = np.random.normal(size=(M, K * T))
    np.random.seed(5)
    model._resample_V(Y)
    assert model.v_sampler() == ("chain" if sampler == "banded" else "generic")
    ost = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in st.items()}
    orc.v_step(ost, Y, orc.trend_penalty(T, tf), z=zv, perm=orc.perm_from_order(model.v_order(), K, T))
    assert relerr(model.V, ost["V"]) < 1e-8


@pytest.mark.parametrize("N,M,T,R,K,tf,missing", [
    (20, 5, 64, 2, 10, 2, True),      # C3's depth, nembeds 10, weighted: 640 x 31 band, two chunks
    (20, 5, 64, 2, 9, 2, True),
    (20, 5, 64, 2, 10, 2, False),     # complete data, sampler="banded" (the spectral sampler is the default there)
    (20, 3, 64, 1, 8, 3, True),       # bw 32: the widest band the chain kernels take
    (15, 3, 150, 2, 6, 2, True),      # 900 x 19
    (15, 2, 41, 2, 10, 2, True),      # the last chunk shorter than the others
])
def test_chunked_chain_matches_oracle_and_the_any_size_kernel(N, M, T, R, K, tf, missing):
    """Bands that do not fit LDS in one piece (weighted data from nembeds 8 at C3's depth): the chunked chain draws the
    same depth-major x = Q^-1 mu + L^-T D^-1/2 z as the any-size kernel it replaces - and as the oracle."""
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    from oracle import btf_oracle as orc
    Y, st = make_case(N, M, T, R, K, tf, missing, seed=N * 31 + K)
    out = {}
    for sampler in ("banded", "generic"):
        model = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=tf, sigma2_init=st["sigma2"], lam2_init=st["lam2"],
                                                nu2_init=st["nu2"], W_init=st["W"], V_init=st["V"], Tau2_init=st["Tau2"], sampler=sampler)
        np.random.seed(5)
        model._resample_V(Y)
        assert model.v_sampler() == ("chain" if sampler == "banded" else "generic")
        assert list(model.v_order()) == list(range(K * T))
        out[sampler] = model.V.copy()
    np.random.seed(5)
    zv = np.random.normal(size=(M, K * T))
    ost = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in st.items()}
    orc.v_step(ost, Y, orc.trend_penalty(T, tf), z=zv, perm="depth")
    assert relerr(out["banded"], ost["V"]) < 1e-8
    assert relerr(out["banded"], out["generic"]) < 1e-9


def test_chunked_chain_jitter_retries_and_failure():
    """The chunked chain restarts from its first chunk when a pivot fails: a column made indefinite must be rescued by
    the same cumulative jitter schedule as the any-size kernel (same retries, same draw), a grossly indefinite one
    reported with its index, and the context usable afterwards."""
    import ctypes as C
    from functionalmf_amd._native import NotPositiveDefiniteError
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    N, M, T, R, K, tf = 20, 4, 64, 2, 10, 2
    Y, st = make_case(N, M, T, R, K, tf, True, seed=91)

    def run(sampler, tau_entry):
        m = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=tf, sigma2_init=st["sigma2"], lam2_init=st["lam2"],
                                            nu2_init=st["nu2"], W_init=st["W"], V_init=st["V"], Tau2_init=st["Tau2"].copy(),
                                            sampler=sampler, force_psd_eps=10.0, force_psd_attempts=4)
        m.Tau2[2, 150] = tau_entry                      # a penalty row in the second chunk of column 2
        np.random.seed(3)
        m._resample_V(Y)
        return m

    tries = {}
    out = {}
    for sampler in ("banded", "generic"):
        m = run(sampler, -0.05)                         # 1 / (lam2 Tau2) = -67 on a few diagonal entries: shift 10 fails, 110 rescues
        m.sync()
        assert m.v_sampler() == ("chain" if sampler == "banded" else "generic")
        t = np.zeros(M, dtype=np.int32)
        m._ctx.call("btf_get_V_attempts", t.ctypes.data_as(C.POINTER(C.c_int32)))
        tries[sampler], out[sampler] = t.copy(), m.V.copy()
    assert tries["banded"][2] >= 1 and (tries["banded"] == tries["generic"]).all()
    assert relerr(out["banded"], out["generic"]) < 1e-8
    m = run("banded", -1e-6)                            # hopeless: every retry fails
    with pytest.raises(NotPositiveDefiniteError) as e:
        m.sync()
    assert e.value.index == 2
    m.Tau2 = st["Tau2"].copy()
    np.random.seed(3)
    m._resample_V(Y)
    m.sync()
    assert np.isfinite(m.V).all()


@pytest.mark.parametrize("N,M,T,R,K,tf", [(2100, 3, 5, 2, 2, 1), (9, 40, 64, 1, 3, 2), (2300, 36, 64, 1, 8, 2)])
def test_long_row_ranges_per_workgroup(N, M, T, R, K, tf):
    """Row ranges of >= 2048 rows per workgroup take the three-rows-in-flight build of the complete-data accumulation
    (what C5-sized slabs run): forced here through btf_set_tuning on shapes the oracle handles."""
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    from oracle import btf_oracle as orc
    Y, st = make_case(N, M, T, R, K, tf, False, seed=N + M)
    model = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=tf, sigma2_init=st["sigma2"], lam2_init=st["lam2"],
                                            nu2_init=st["nu2"], W_init=st["W"], V_init=st["V"], Tau2_init=st["Tau2"], sampler="spectral")
    model._ctx.call("btf_set_tuning", 2048, 2048)
    np.random.seed(5)
    zw = np.random.normal(size=sum(min(i + 1, K) for i in range(N)))
    zv = np.random.normal(size=(M, K * T))
    np.random.seed(5)
    model._resample_W(Y)
    model._resample_V(Y)
    ost = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in st.items()}
    Rr, ybar = orc.hoisted_stats(Y if Y.ndim == 4 else Y[..., None])
    orc.w_step_strong(ost, Rr, ybar, z=zw)
    assert relerr(model.W, ost["W"]) < 1e-10
    orc.v_step_strong(ost, Rr, ybar, orc.trend_penalty(T, tf), z=zv, order="spectral")
    assert relerr(model.V, ost["V"]) < 1e-7


def test_spectral_sampler_long_depth_axis_records_in_hbm():
    """The reference's flu data (flutrends/benchmark.py:31-34: 50 states x 1 x 370 weeks, nembeds 10): the pivot records of
    the spectral sampler (T K (tf + 3) doubles = 148 KB) no longer fit LDS next to everything else, so
    v_spectral_kernel<S, true> keeps them in HBM scratch - same declared square root, checked against the oracle's
    (v_step_strong(order="spectral")) from identical state and normals; the any-size banded kernel took 9 ms per draw
    here (3700 sequential pivots of one wave), this one 87 us."""
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    from oracle import btf_oracle as orc
    N, M, T, R, K = 50, 1, 370, 1, 10
    rs = np.random.RandomState(3)
    Wt = rs.normal(size=(N, K))
    Wt[np.triu_indices(K, 1)] = 0
    Vt = 0.1 * np.cumsum(rs.normal(size=(M, T, K)), axis=1)
    Y = np.einsum("nk,mtk->nmt", Wt, Vt)[..., None] + rs.normal(0, 0.5, size=(N, M, T, R))
    st = dict(W=Wt + 0.1 * rs.normal(size=Wt.shape), V=Vt + 0.05 * rs.normal(size=Vt.shape), Tau2=rs.gamma(2.0, 0.5, size=(M, 3 * T - 1)),
              lam2=0.2, sigma2=0.6, nu2=0.3)
    st["W"][np.triu_indices(K, 1)] = 0
    m = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, sigma2_init=st["sigma2"], lam2_init=st["lam2"], nu2_init=st["nu2"],
                                        W_init=st["W"], V_init=st["V"], Tau2_init=st["Tau2"], compat="exact", sampler="spectral")
    zv = np.random.RandomState(5).normal(size=(M, K * T))
    m._v_normals = lambda: zv
    m._resample_V(Y)
    assert m.v_sampler() == "spectral"
    Rr, ybar = orc.hoisted_stats(Y)
    ost = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in st.items()}
    orc.v_step_strong(ost, Rr, ybar, orc.trend_penalty(T, 2), z=zv, order="spectral")
    assert np.abs(m.V - ost["V"]).max() / np.abs(ost["V"]).max() < 1e-9
    # a device-RNG sweep loop at this shape runs (records scratch reused, warm eigen-solves)
    m2 = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, sigma2_init=0.5, lam2_init=0.1, nu2_init=1.0, rng="device")
    for _ in range(5):
        m2.resample(Y)
    assert m2.v_sampler() == "spectral" and np.all(np.isfinite(m2.V)) and np.all(np.isfinite(m2.W))
