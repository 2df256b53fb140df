"""The two-launch W+V step (BTF_OPT_FUSED_STEP, csrc/btf_fused.h) against the four-launch path.

The batched K x K solve of `_resample_W` (factor.py:349-362) runs as the tail of the W accumulation launch, the spectral
sampler of `_resample_V` (factor.py:377-409 with fast_mvn.py:35-47) as the tail of the V accumulation launch.  Both tails
repeat the arithmetic of w_solve_kernel / v_spectral_kernel in the same order, so the chains must agree BIT FOR BIT with
the four-launch path - for host normals (the reference-reproducible mode) and for device normals, under both `compat`s,
at shapes with one and with several chunks per tile, whole and ragged tiles.  (The barrier-free form of the V tail, the
library's default where it applies, deals the rows over fewer waves: equal to rounding, pinned by its own tests below.)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _synth(N, M, T, R, K, seed=3):
    rs = np.random.RandomState(seed)
    Wt = rs.normal(size=(N, K))
    Vt = 0.1 * np.cumsum(rs.normal(size=(M, T, K)), axis=1)
    return np.einsum("nk,mtk->nmt", Wt, Vt)[..., None] + rs.normal(0, 0.5, size=(N, M, T, R))


def _make(dims, fused, rng, compat, sampler="spectral", rpb=None, seed=11, dataflow=0):
    from functionalmf_amd import _native
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    N, M, T, R, K = dims
    np.random.seed(seed)
    m = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, sigma2_init=0.5, lam2_init=0.1, nu2_init=1.0,
                                        rng=rng, device_seed=5, compat=compat, sampler=sampler)
    m._ctx.call("btf_set_option", _native.OPT_FUSED_STEP, int(fused))      # 0: four launches, 1: V launch fused, 2: W launch too
    # the tail of the fused V launch: 0 - the barrier tail, bit-identical to the four-launch form (what most tests here
    # compare with); 1 - the barrier-free (dataflow) tail, the default of the library: same draw, column sums grouped differently
    m._ctx.call("btf_set_option", _native.OPT_FUSED_DATAFLOW, int(dataflow))
    if rpb:
        m._ctx.call("btf_set_tuning", rpb[0], rpb[1])
    return m


def _launches(m):
    kt = m._ctx.kernel_times()
    return {k: v[1] for k, v in kt.items() if v[1]}


# (N, M, T, R, K): one chunk per V tile and 2 columns per tile (T = 64); four columns per tile (T = 32); one (T = 128);
# a ragged last tile (M T not a multiple of 128); several W tiles; nembeds 8
SHAPES = [(96, 6, 64, 2, 5), (70, 9, 32, 2, 3), (40, 3, 128, 1, 4), (300, 5, 64, 2, 8), (130, 7, 64, 3, 1)]


@pytest.mark.parametrize("dims", SHAPES)
@pytest.mark.parametrize("rng", ["host", "device"])
@pytest.mark.parametrize("mode", [1, 2])
def test_two_launch_step_is_bit_identical_to_four_launches(dims, rng, mode):
    Y = _synth(*dims)
    a, b = _make(dims, mode, rng, "reference"), _make(dims, 0, rng, "reference")
    for m in (a, b):
        np.random.seed(21)
        m._bind_data(Y)
        m._ctx.kernel_times()
        for _ in range(4):
            m._resample_W(Y)
            m._resample_V(Y)
        m.sync()
    la, lb = _launches(a), _launches(b)
    # two / three launches per step, and ONE prior_band launch in all: the fused V tails load the precomputed prior band of
    # their columns, rebuilt only when Tau2 / lam2 change
    assert la.pop("prior_band") == 1, la
    assert la == ({"w_accum": 4, "v_accum": 4} if mode == 2 else {"w_accum": 4, "w_solve": 4, "v_accum": 4}), la
    assert lb.pop("prior_band") == 1, lb         # (the sampler launch of its own loads the same band from its second step on)
    assert lb == {"w_accum": 4, "w_solve": 4, "v_accum": 4, "v_banded": 4}, lb
    assert np.array_equal(a.W, b.W)
    assert np.array_equal(a.V, b.V)
    assert np.isfinite(a.W).all() and np.isfinite(a.V).all() and np.abs(a.V).max() > 0


@pytest.mark.parametrize("compat", ["reference", "exact"])
def test_two_launch_step_several_chunks_per_tile(compat):
    """Rows-per-workgroup overrides force several chunks per tile in BOTH launches: the last arriver of a tile sums the
    chunks of the other workgroups (write-through partials, tickets)."""
    dims = (640, 6, 64, 2, 5)
    Y = _synth(*dims)
    a, b = _make(dims, 2, "device", compat, rpb=(64, 128)), _make(dims, 0, "device", compat, rpb=(64, 128))
    for m in (a, b):
        for _ in range(6):
            m._resample_W(Y)
            m._resample_V(Y)
        m.sync()
    assert np.array_equal(a.W, b.W) and np.array_equal(a.V, b.V)


def test_two_launch_full_sweeps_walk_the_same_chain():
    """Full device sweeps (nu2, sigma2, Tau2 chain, lam2, W, V): the scalar draws that ride in the accumulation launches as
    side workgroups reach the tails of the same launch through their published copies."""
    dims = (96, 8, 64, 2, 5)
    Y = _synth(*dims)
    a, b = _make(dims, 2, "device", "reference"), _make(dims, 0, "device", "reference")
    for m in (a, b):
        m.resample(Y)
        m.resample_sweeps(Y, 6)
        m.sync()
    assert np.array_equal(a.W, b.W) and np.array_equal(a.V, b.V) and np.array_equal(np.asarray(a.Tau2), np.asarray(b.Tau2))
    assert (a.nu2, a.sigma2, a.lam2) == (b.nu2, b.sigma2, b.lam2)
    c = _make(dims, 1, "device", "reference")
    for _ in range(7):
        c.resample(Y)                                                   # Python-driven: the same decisions
    assert np.array_equal(a.W, c.W) and np.array_equal(a.V, c.V)


@pytest.mark.parametrize("dims", [(96, 8, 64, 2, 5), (40, 3, 128, 1, 4), (70, 9, 32, 2, 3)])
@pytest.mark.parametrize("dataflow", [0, 1])
def test_full_sweeps_with_lam2_and_the_prior_band_inside_the_w_solve_launch(monkeypatch, dataflow, dims):
    """Full device sweeps: lam2 | rest as one more workgroup of the w_solve launch and the prior band (+ its LDS image) by one
    workgroup per column behind it (BandSide, csrc/btf_kernels.h: they wait for the lam2 workgroup's flag), against the band
    as a launch of its own in front of the V launch (BTF_BAND_IN_WSOLVE=0) and against lam2 drawn by a side workgroup of
    the V launch whose tails form the band themselves (BTF_LAM_IN_WSOLVE=0): the same conditionals, the same Philox
    streams, the same band bits - with the barrier tail all three chains coincide bit for bit (and with the four-launch
    sweep: test_two_launch_full_sweeps_walk_the_same_chain); with the dataflow tail the first two do (the third runs the
    barrier tail: equal to rounding).  No launch for the band in the first form.  Depth axes of 64, 128 and 32 (two, one, four
    columns per tile; 256 / 512 / 128 band entries and 191 / 383 / 95 penalty rows per column)."""
    Y = _synth(*dims)
    outs, launches = [], []
    for lam_in, band_in in ((1, 1), (1, 0), (0, 0)):
        monkeypatch.setenv("BTF_LAM_IN_WSOLVE", str(lam_in))
        monkeypatch.setenv("BTF_BAND_IN_WSOLVE", str(band_in))
        m = _make(dims, 1, "device", "reference", dataflow=dataflow)
        m.resample(Y)
        m.resample(Y)
        m._ctx.kernel_times()
        m.resample_sweeps(Y, 6)
        m.sync()
        launches.append(_launches(m))
        outs.append((m.W.copy(), m.V.copy(), np.asarray(m.Tau2).copy(), (m.nu2, m.sigma2, m.lam2)))
    assert "prior_band" not in launches[0] and launches[1].get("prior_band") == 6 and "prior_band" not in launches[2], launches
    a, b, c = outs
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3]
    if dataflow == 0:
        assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1]) and np.array_equal(a[2], c[2]) and a[3] == c[3]
    else:
        # (eight sweeps: the differently grouped sums move V at its conditioning-limited 1e-6 - the tolerance of the V fixtures)
        assert np.abs(a[1] - c[1]).max() < 1e-5 * np.abs(c[1]).max() and abs(a[3][2] - c[3][2]) < 1e-6 * c[3][2]
    assert np.isfinite(a[1]).all()


def test_two_launch_step_at_c3_size_matches_and_repeats():
    """BASELINE config 3 (512,256,64,4) nembeds 5: 128 + 1 workgroups per launch, every tail polling the eigen side
    workgroup's flag; 30 steps, fused against unfused, and the fused run twice (the tickets / epochs carry no state
    from one run into the next)."""
    dims = (512, 256, 64, 4, 5)
    Y = _synth(*dims, seed=1)
    runs = []
    for fused in (2, 0, 1, 2):
        m = _make(dims, fused, "device", "reference")
        for _ in range(30):
            m._resample_W(Y)
            m._resample_V(Y)
        m.sync()
        runs.append((m.W.copy(), m.V.copy()))
    for W, V in runs[1:]:
        assert np.array_equal(runs[0][0], W) and np.array_equal(runs[0][1], V)


@pytest.mark.parametrize("rng", ["host", "device"])
def test_precomputed_prior_band_follows_the_hyper_parameters(rng):
    """The fused V tails load a prior band that is rebuilt only when Tau2 / lam2 change (prior_version in btf_abi.hip): new
    values pushed from the host between steps - the Tau2 array, lam2 alone, both - must reach the next V draw, i.e. the chain
    stays bit-identical to the four-launch path, which forms the band from Tau2 and lam2 inside the sampler every time."""
    dims = (96, 6, 64, 2, 5)
    Y = _synth(*dims)
    a, b = _make(dims, 1, rng, "reference"), _make(dims, 0, rng, "reference")
    rs = np.random.RandomState(4)
    nD = np.asarray(a.Tau2).shape[1]
    edits = [None, ("Tau2", rs.gamma(2.0, 0.5, size=(dims[1], nD))), ("lam2", 0.37), None,
             ("both", rs.gamma(2.0, 0.5, size=(dims[1], nD)), 0.05), None]
    for m in (a, b):
        np.random.seed(5)
        m._bind_data(Y)
        m._ctx.kernel_times()
        for e in edits:
            if e is not None:
                if e[0] in ("Tau2", "both"):
                    m.Tau2 = e[1].copy()
                if e[0] == "lam2":
                    m.lam2 = e[1]
                if e[0] == "both":
                    m.lam2 = e[2]
            m._resample_W(Y)
            m._resample_V(Y)
        m.sync()
    assert np.array_equal(a.W, b.W) and np.array_equal(a.V, b.V)
    la = _launches(a)
    assert la["prior_band"] == 4, la            # first use + three edits; the steps without an edit reuse the band
    assert "v_banded" not in la and la["v_accum"] == len(edits), la


@pytest.mark.parametrize("rng", ["host", "device"])
def test_twisted_sampler_with_the_precomputed_band_walks_the_same_chain(rng):
    """Weighted data (missing replicates): from the second V half-sweep on with unchanged Tau2 / lam2 the twisted sampler's
    workgroups load the prior band of their column (prior_band_kernel in the sampler's own arithmetic) instead of forming it
    from their stencil.  A model whose Tau2 is re-sent before every step (same values: the band is formed in the kernel every
    time) must walk the same chain bit for bit."""
    dims = (96, 6, 64, 2, 5)
    Y = _synth(*dims)
    Y[np.random.RandomState(8).rand(*Y.shape) < 0.1] = np.nan
    a, b = _make(dims, 1, rng, "reference", sampler="banded"), _make(dims, 1, rng, "reference", sampler="banded")
    for m, resend in ((a, False), (b, True)):
        np.random.seed(9)
        m._bind_data(Y)
        m._ctx.kernel_times()
        for _ in range(5):
            if resend:
                m.Tau2 = np.array(m.Tau2).copy()
            m._resample_W(Y)
            m._resample_V(Y)
        m.sync()
    assert a.v_sampler() == "banded" and a.likelihood_form() == "weighted"
    assert np.array_equal(a.W, b.W) and np.array_equal(a.V, b.V)
    la, lb = _launches(a), _launches(b)
    assert la.get("prior_band") == 1 and "prior_band" not in lb, (la, lb)


def test_dataflow_tail_is_the_default_where_it_applies_and_equals_the_barrier_tail():
    """The barrier-free tail of the fused V launch (csrc/btf_fused.h, v_fused_df: LDS counters instead of workgroup barriers;
    the columns' chain waves do not stream - they draw the normals, take the eigenvalues and factor while the other waves
    stream) against the barrier tail (BTF_OPT_FUSED_DATAFLOW 0) and the four-launch form at the shapes it takes - two
    columns per tile (T = 64), four (T = 32), one (T = 128), a ragged last tile, few rows, nembeds 1 and 6 - with host
    and device normals.  The same draw from the same sums: the rows are dealt over 16 - NG waves instead of 16, so the
    column sums are grouped differently - equal to rounding (1e-12 of the factor's scale over three steps), not bit for
    bit; the dataflow form itself is deterministic (two runs: identical bits), and the barrier tail still equals the
    four-launch form bit for bit."""
    shapes = [(96, 6, 64, 2, 5), (70, 9, 32, 2, 3), (40, 3, 128, 1, 4), (130, 7, 64, 3, 1), (24, 5, 64, 1, 6), (600, 4, 64, 2, 5)]
    for dims in shapes:
        Y = _synth(*dims)
        for rng in ("device", "host"):
            outs = []
            for mode, df in ((1, 1), (1, 1), (1, 0), (0, 1)):
                m = _make(dims, mode, rng, "reference", dataflow=df)
                np.random.seed(31)
                for _ in range(3):
                    m._resample_W(Y)
                    m._resample_V(Y)
                m.sync()
                outs.append((m.W.copy(), m.V.copy()))
            assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]), (dims, rng)      # deterministic
            assert np.array_equal(outs[2][0], outs[3][0]) and np.array_equal(outs[2][1], outs[3][1]), (dims, rng)      # barrier tail == four launches
            sw, sv = np.abs(outs[2][0]).max(), np.abs(outs[2][1]).max()
            assert np.abs(outs[0][0] - outs[2][0]).max() < 1e-11 * sw and np.abs(outs[0][1] - outs[2][1]).max() < 1e-11 * sv, (dims, rng)
            assert np.isfinite(outs[0][1]).all() and np.abs(outs[0][1]).max() > 0


@pytest.mark.timeout(600)
def test_in_launch_hand_offs_hold_over_thousands_of_launches():
    """Stress of the in-launch hand-offs: 3000 W+V steps each of (a) the dataflow tail at C3 size - 128 + 1
    workgroups per V launch, LDS counters inside every workgroup, the eigenvalue granules and the eigen-system published by
    the side workgroup and read under full streaming load - run twice: one stale partial, one early read of a granule or
    one lost count anywhere in 3000 steps changes the final state, which must be bit-identical between the runs (the sums
    of this form are grouped differently from the other forms': its values are pinned over short chains above); and (b)
    the ticketed barrier tail with several chunks per tile in both launches (write-through partials, last arriver per
    tile) against the four-launch chain, bit for bit."""
    cases = [((512, 256, 64, 4, 5), 1, 1, None, 3000), ((640, 6, 64, 2, 5), 2, 0, (64, 128), 3000)]
    for dims, mode, other, rpb, steps in cases:
        Y = _synth(*dims, seed=2)
        ends = []
        for fused in (mode, other):
            m = _make(dims, fused, "device", "reference", rpb=rpb, dataflow=1)
            for _ in range(steps):
                m._resample_W(Y)
                m._resample_V(Y)
            m.sync()
            ends.append((m.W.copy(), m.V.copy()))
        assert np.array_equal(ends[0][0], ends[1][0]) and np.array_equal(ends[0][1], ends[1][1]), dims
        assert np.isfinite(ends[0][1]).all()


@pytest.mark.parametrize("dims,variant", [((96, 6, 64, 2, 5), "complete"), ((70, 9, 32, 2, 3), "missing"), ((40, 3, 24, 1, 4), "complete")])
def test_c_driven_wv_steps_walk_the_python_driven_chain(dims, variant):
    """model.wv_steps(data, n) (btf_wv_steps: n W+V updates queued by one call into the C side - what bench.py times) against
    n Python-driven (_resample_W, _resample_V) pairs: the same launches, the same seeds, the same chain bit for bit; and
    the two may be mixed (the draw counter moves on by two per step either way)."""
    Y = _synth(*dims)
    if variant == "missing":
        rs = np.random.RandomState(5)
        Y[rs.rand(*Y.shape) < 0.1] = np.nan
    a, b = _make(dims, 1, "device", "reference", sampler="auto", dataflow=1), _make(dims, 1, "device", "reference", sampler="auto", dataflow=1)
    a.wv_steps(Y, 7)
    for _ in range(7):
        b._resample_W(Y)
        b._resample_V(Y)
    a.sync()
    b.sync()
    assert np.array_equal(a.W, b.W) and np.array_equal(a.V, b.V)
    a._resample_W(Y)
    a._resample_V(Y)
    a.wv_steps(Y, 3)
    b.wv_steps(Y, 2)
    for _ in range(2):
        b._resample_W(Y)
        b._resample_V(Y)
    assert np.array_equal(a.W, b.W) and np.array_equal(a.V, b.V) and np.isfinite(a.V).all()
