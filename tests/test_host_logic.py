"""CPU-only tests: the C-ABI library loads and exports every symbol include/btf.h
declares, and the host logic around it (penalty, stale-source maps, driver, sharding
plan) agrees with the fixtures / oracle.  No compute call into the library here."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
from conftest import ROOT, relerr
from oracle import btf_oracle as orc


def header_functions():
    text = open(os.path.join(ROOT, "include", "btf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(btf_[a-z_A-Z0-9]+)\s*\(", text)))


def test_abi_exports_every_declared_symbol():
    from functionalmf_amd import _native
    _native.build()
    lib = _native.load()
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libbtf_hip.so does not export %s" % n
    assert set(names) == set(_native.SIGNATURES), set(names) ^ set(_native.SIGNATURES)


def test_abi_signatures_are_plain_c():
    text = open(os.path.join(ROOT, "include", "btf.h")).read()
    assert 'extern "C"' in text
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    assert "torch" not in code and "std::" not in code and "&" not in code


def test_no_cpu_fallback_without_library(tmp_path, monkeypatch):
    from functionalmf_amd import _native
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_native, "_lib", None)
    with pytest.raises(ImportError):
        _native.load()


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "functionalmf_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("the oracle", ""), f


def test_penalty_matches_reference_fixture(golden):
    from functionalmf_amd.utils import bayes_grid_penalty
    g = golden("g0_delta.npz")
    for key, ref in g.items():
        _, T, k = key.split("_")
        D = bayes_grid_penalty(int(T[1:]), int(k[1:]))
        assert D.format == "csc" and np.array_equal(D.toarray(), ref)


def test_stale_sources_match_oracle(golden):
    from functionalmf_amd.factor import stale_col_sources, stale_row_sources
    for name in ("g1_c1_heldout.npz", "g3_partial_reps.npz"):
        Y = golden(name)["Y"]
        cnt, ybar = orc.replicate_stats(Y)
        assert np.array_equal(stale_col_sources(np.isnan(Y).all(-1)), orc.stale_column_sources(ybar))
    g = golden("g4_binomial_nan.npz")
    assert np.array_equal(stale_col_sources(np.isnan(g["Ysucc"])), [0, 0, 0, 3, 3, 3, 6, 7])
    assert np.array_equal(stale_row_sources(7, 3, False), [0, 1, 2, 2, 2, 2, 2])
    assert np.array_equal(stale_row_sources(7, 3, True), np.arange(7))


def test_run_gibbs_driver_layout():
    from functionalmf_amd.genlasso import _BayesianModel

    class Toy(_BayesianModel):
        def __init__(self):
            super().__init__(nthreads=3)
            self.t = 0

        def resample(self, data):
            self.t += 1

        def _inferred_variables(self, m):
            m["a"] = float(self.t)
            m["B"] = np.full((2, 3), self.t)

    seen = []
    res = Toy().run_gibbs(None, nburn=3, nthin=2, nsamples=4, verbose=False,
                          callback=lambda m, d, s: seen.append(s))
    assert res["a"].shape == (4, 1) and res["B"].shape == (4, 2, 3)
    assert res["a"][:, 0].tolist() == [4, 6, 8, 10] and seen == list(range(11))


def test_inverse_gamma_update_matches_reference_stream(golden):
    from functionalmf_amd.genlasso import ConjugateInverseGammaPrior
    g = golden("g1_c1_heldout.npz")
    prior = ConjugateInverseGammaPrior(1, 0.1, 0.1)
    np.random.seed(200)
    nu2 = 1 / prior.resample_from_stats(float(g["h_sse"]), int(g["h_nobs"]))
    assert abs(nu2 - g["h_nu2"]) / g["h_nu2"] < 1e-13
    np.random.seed(5)
    a = prior.resample((np.zeros(4), np.array([1.0, np.nan, 2.0, -1.0])))
    np.random.seed(5)
    b = np.random.gamma(0.1 + 1.5, 1 / (0.1 + 3.0))
    assert a == b


def test_shard_plan_blocks_cover_axes():
    from functionalmf_amd.parallel import ShardPlan
    for n, m, world in ((512, 256, 8), (10, 11, 2), (10, 11, 3), (5, 2, 4)):
        rows, cols = [], []
        for r in range(world):
            p = ShardPlan(n, m, r, world)
            rows += list(range(p.row0, p.row0 + p.nl))
            cols += list(range(p.col0, p.col0 + p.ml))
            assert p.nl <= p.row_chunk and p.ml <= p.col_chunk
            assert p.world * p.row_chunk <= n + 64 and p.world * p.col_chunk <= m + 64
        assert rows == list(range(n)) and cols == list(range(m))


def test_w_z_offsets_follow_reference_stream_order():
    """z for row i starts at sum_{i'<i} min(i'+1, K) (factor.py:361 draws d normals per row)."""
    K = 3
    offs = [sum(min(i + 1, K) for i in range(r)) for r in range(8)]
    closed = [r * (r + 1) // 2 if r < K else K * (K + 1) // 2 + (r - K) * K for r in range(8)]
    assert offs == closed


@pytest.mark.timeout(300)
def test_sharded_half_sweeps_equal_unsharded_gloo_world2():
    """world_size-2 gloo run of the row/column sharding: every rank updates its block with
    the oracle (standing in for the kernels, CPU box) and the product's ShardPlan/Exchange
    reassemble W, V and the SSE scalars; must equal the unsharded update bit for bit."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29571",
           os.path.join(ROOT, "tests", "dist_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count("DIST_OK") == 2, out.stdout[-2000:]


@pytest.mark.timeout(300)
def test_bench_launches_its_own_ranks_and_prints_one_line():
    """`python bench.py --gpus 2` outside torchrun must start its two ranks as a child process (the way the
    driver starts the 1-GPU run) and relay exactly one JSON line with n_gpus = 2, strong scaling on config C5.
    BTF_BENCH_DRY=1: the launch / barrier / max-over-ranks / reporting plumbing on CPU over gloo, no GPU work."""
    import json
    env = dict(os.environ, BTF_BENCH_DRY="1", PYTHONPATH=ROOT)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                          "--master-port", "29577"], env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 2 and d["scaling"] == "strong"
    assert d["config"]["rccl_ranks"] == 2 and d["config"]["backend"] == "gloo" and "c5" in d["config"]["workload"]
    # the same entry point under torchrun (the driver's N>1 form) is a rank, not a launcher
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29578", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_shard_plan_finds_the_one_halo_source_and_appends_it_to_the_slabs():
    """compat="reference" sharded (btf_set_shard_halo): per contiguous block at most one stale-weight source row and one
    source column lie outside it; ShardPlan.halo_of finds them from the reference's source patterns (factor.py:320,349 and
    :394-401, restated by stale_row_sources / stale_col_sources) and ShardPlan.slabs carries them LAST."""
    from functionalmf_amd.factor import stale_col_sources, stale_row_sources
    from functionalmf_amd.parallel import ShardPlan
    N, M, T, K = 12, 10, 4, 3
    group = np.array([0, 0, 0, 1, 1, 1, 1, 2, 2, 2])
    pat = np.random.RandomState(0).rand(3, N, T) < 0.3
    miss = pat[group].transpose(1, 0, 2)                                  # (N, M, T)
    src_col = stale_col_sources(miss)
    assert list(src_col) == [0, 0, 0, 3, 3, 3, 3, 7, 7, 7]
    Y4 = np.arange(N * M * T, dtype=float).reshape(N, M, T, 1)
    for world, want in ((2, [(-1, -1), (-1, 3)]), (3, [(-1, -1), (-1, 3), (-1, 7)])):
        for r in range(world):
            p = ShardPlan(N, M, r, world)
            assert p.halo_of(stale_row_sources(N, K, True), src_col) == want[r]
            p.halo_row, p.halo_col = want[r]
            rows, cols = p.slabs(Y4)
            assert rows.shape[0] == p.nl and cols.shape[1] == p.ml + (want[r][1] >= 0)
            if want[r][1] >= 0:
                assert np.array_equal(cols[:, -1], Y4[:, want[r][1]]) and np.array_equal(cols[:, :-1], Y4[:, p.col0:p.col0 + p.ml])
    # no NaN anywhere: every row >= nembeds reads row nembeds-1, every column reads column 0
    src_row, src_col = stale_row_sources(N, K, False), stale_col_sources(np.zeros((N, M, T), bool))
    for r in range(1, 3):
        p = ShardPlan(N, M, r, 3)
        assert p.halo_of(src_row, src_col) == (K - 1, 0)
        p.halo_row, p.halo_col = K - 1, 0
        rows, cols = p.slabs(Y4)
        assert rows.shape[0] == p.nl + 1 and np.array_equal(rows[-1], Y4[K - 1]) and np.array_equal(cols[:, -1], Y4[:, 0])
    assert ShardPlan(N, M, 0, 3).halo_of(src_row, src_col) == (-1, -1)         # rank 0 owns both sources
    with pytest.raises(ValueError):
        ShardPlan(N, M, 1, 2).halo_of(src_row, np.array([0, 1, 2, 3, 4, 0, 1, 7, 8, 9], dtype=np.int32))


def test_shard_plan_rejects_more_shards_than_the_padding_allows():
    from functionalmf_amd.parallel import ShardPlan
    with pytest.raises(ValueError):
        ShardPlan(1000, 1000, 0, 65)


def test_no_vgpr_spills_in_the_hot_kernels():
    """(and bounded SGPR spills in the instances of BASELINE configs 3 / 4)  Code-object notes of the built library (scripts/kernel_notes.py: the NT_AMDGPU_METADATA note of every embedded
    gfx950 ELF): the streaming accumulation, the W solve and the flat Polya-Gamma kernels must not spill VGPRs for any
    supported nembeds (1..10) - K = 9 / 10 weighted accumulation, w_solve<6, weighted> and the K = 10 eigen side task
    did before round 3."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("kernel_notes", os.path.join(ROOT, "scripts", "kernel_notes.py"))
    kn = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kn)
    rows = kn.kernels()
    assert len(rows) > 600
    hot = [r for r in rows if any(t in r["mangled"] for t in ("accum_kernel", "w_solve_kernel", "pgx_tile_kernel", "pgx_kernel",
                                                             "v_spectral_kernel", "v_banded_twist_kernel"))]
    assert len(hot) > 150
    bad = [(r["mangled"], r["vgpr_spill"]) for r in hot if r["vgpr_spill"]]
    assert not bad, bad
    ks = {int(m) for r in hot for m in __import__("re").findall(r"accum_kernelILi(\d+)E", r["mangled"])}
    assert ks == set(range(1, 11)), ks
    # SGPR spills of the instances BASELINE configs 3 / 4 launch (VERDICT r03 asked for an honest count): ceilings a little
    # above what the shipped build has (kernel_notes: accumulation 19 - all in the side-task code, the streaming waves
    # execute none -, spectral sampler 6, W solves and the flat Polya-Gamma kernel 0, twisted sampler 121 - its target of
    # < 32 is NOT met), so that a change that brings spills back is seen here
    c3 = {"accum_kernelILi5ELi0ELi16EddLi0ELi2ELi0E": 24, "accum_kernelILi5ELi0ELi16EddLi0ELi2ELi2E": 24,
          "accum_kernelILi5ELi1ELi12EhdLi0ELi2ELi0E": 24, "accum_kernelILi5ELi2ELi12EdaLi0ELi2ELi0E": 24,
          "w_solve_kernelILi5ELb0ELi8E": 0, "w_solve_kernelILi5ELb1ELi8E": 0, "v_spectral_kernelILi3ELb0ELi5E": 16,
          "pgx_tile_kernelILi5ELi4ELi4E": 0, "v_banded_twist_kernelILi2ELb1ELi5ELi2ELi64E": 128}
    for tag, cap in c3.items():
        inst = [r for r in rows if tag in r["mangled"]]
        assert len(inst) == 1, (tag, len(inst))
        assert inst[0]["sgpr_spill"] <= cap and inst[0]["vgpr_spill"] == 0, (tag, inst[0]["sgpr_spill"], inst[0]["vgpr_spill"])


def test_host_selftest_of_tables_layouts_and_chunk_maps():
    """btf_host_selftest (include/btf.h): the host-made stencil tables, LDS layouts, elimination orders, band assembly
    program and (split) accumulation chunk maps over a grid of shapes - no GPU involved."""
    from functionalmf_amd import _native
    assert _native.load().btf_host_selftest() == 0


@pytest.mark.timeout(900)
def test_host_side_under_address_and_ub_sanitizers():
    """scripts/asan_host.sh: the same self-test on a build whose host side carries AddressSanitizer and
    UndefinedBehaviorSanitizer (device code unchanged; CPU only - GPU sanitizers are not available on this pool)."""
    out = subprocess.run(["bash", os.path.join(ROOT, "scripts", "asan_host.sh")], capture_output=True, text=True, timeout=850)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "btf_host_selftest: 0" in out.stdout and "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr


def test_one_standard_gamma_call_equals_the_references_per_column_gamma_calls():
    """What the vectorised parity-mode Tau2 step relies on (functionalmf_amd/factor.py:_resample_Tau2; the reference draws,
    per column, gamma((K+1)/2, scale) and three gamma(1, scale) vectors: factor.py:134-141): a legacy gamma(shape, scale) is
    scale * standard_gamma(shape) and consumes the generator by its shape alone, so ONE standard_gamma call over the shapes
    in that order gives the same variates bit for bit and leaves the legacy stream at the same position."""
    M, nD, shape = 7, 23, 3.0
    rs = np.random.RandomState(3)
    sc = np.array([rs.rand(4, nD) + 0.1 for _ in range(M)])
    np.random.seed(11)
    ref = np.array([[np.random.gamma(shape if lev == 0 else 1, sc[j][lev]) for lev in range(4)] for j in range(M)])
    after_ref = np.random.normal(size=3)
    np.random.seed(11)
    shapes = np.ones((M, 4, nD))
    shapes[:, 0] = shape
    sg = np.random.standard_gamma(shapes)
    after = np.random.normal(size=3)
    assert np.array_equal(ref, sc * sg) and np.array_equal(after, after_ref)
