"""Worker of test_two_ranks_share_one_gpu: two processes (gloo control plane, exchange staged
through the host because RCCL refuses two ranks on one device) run the SHARDED kernels - row
and column offsets, local slabs, per-rank partial SSE - on cuda:0 and must reproduce the
unsharded oracle.
Also the worker of test_rccl_exchange_with_one_rank_reproduces_the_plain_chain (BTF_DIST_BACKEND=nccl,
BTF_EXERCISE_EXCHANGE=1, one rank): the same checks with the DEVICE collectives - btf_allgather_W / btf_allgather_V on the
context's W / V buffers and btf_allreduce_sse, the 8-byte all-reduce of the residual sum of squares: RCCL calls of the
library itself on the context's communicator and stream (the process group only carries the communicator id) - in the
call sequence of an N-rank RCCL run."""
import os
import sys

import numpy as np
import torch.distributed as dist

OVERLAP = os.environ.get("BTF_DIST_OVERLAP", "0") == "1"      # base section: all-gathers on the communication stream
TRANSPORT = os.environ.get("BTF_EXCHANGE_TRANSPORT") or ("rccl" if os.environ.get("BTF_DIST_BACKEND", "gloo") == "nccl" else "host")
# BTF_DIST_GPU_PER_RANK=1 (test_rccl_ranks_on_their_own_gpus: a box with >= 2 GPUs): rank r runs on cuda:LOCAL_RANK and the
# collectives are real RCCL traffic between devices; otherwise every rank shares cuda:0
DEV = int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("BTF_DIST_GPU_PER_RANK", "0") == "1" else 0
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, state_from  # noqa: E402
from oracle import btf_oracle as orc  # noqa: E402
from functionalmf_amd.factor import GaussianBayesianTensorFiltering  # noqa: E402


def split_section(rank, world):
    """The overlapped exchange (BTF_OPT_SPLIT_ACCUM) on shapes whose chunking lines up with the shards: each half-sweep's
    accumulation runs as two launches - the chunks of this rank's own block queued behind the previous draw, the rest
    (a chunk map with a hole at the start, in the middle or at the end, by rank) behind the gather - and must give the
    unsharded oracle's W and V; complete data and data with missing replicates (weighted accumulation), host normals;
    then whole rng="device" sweeps against the plain one-launch path."""
    from functionalmf_amd import _native
    N, M, T, R, K = 192 * world, 8 * world, 64, 2, 3
    rs = np.random.RandomState(11)
    Wt = rs.normal(size=(N, K))
    Wt[np.triu_indices(K, 1)] = 0
    Vt = 0.1 * np.cumsum(rs.normal(size=(M, T, K)), axis=1)
    Yc = np.einsum("nk,mtk->nmt", Wt, Vt)[..., None] + rs.normal(0, 0.5, size=(N, M, T, R))
    Ym = Yc.copy()
    Ym[rs.rand(N, M, T, R) < 0.1] = np.nan
    Delta = orc.trend_penalty(T, 2)
    st0 = dict(W=Wt + 0.1 * rs.normal(size=Wt.shape), V=Vt + 0.05 * rs.normal(size=Vt.shape), Tau2=rs.gamma(2.0, 0.5, size=(M, Delta.shape[0])),
               lam2=0.2, sigma2=0.6, nu2=0.4)
    st0["W"][np.triu_indices(K, 1)] = 0
    for Y, form in ((Yc, "complete"), (Ym, "weighted")):
        model = GaussianBayesianTensorFiltering(
            N, M, T, nembeds=K, tf_order=2, sigma2_init=st0["sigma2"], lam2_init=st0["lam2"], nu2_init=st0["nu2"],
            W_init=st0["W"], V_init=st0["V"], Tau2_init=st0["Tau2"], compat="exact", shard=(rank, world), device=DEV, sampler="banded",
            overlap_exchange=True)
        model._ctx.call("btf_set_tuning", 64, 64)          # 64-row chunks: every shard is a whole number of them
        ost = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in st0.items()}
        model._ctx.kernel_times()
        for it in range(3):
            np.random.seed(60 + it)
            model._resample_W(Y)
            model._resample_V(Y)
            np.random.seed(60 + it)
            orc.w_step(ost, Y)
            orc.v_step(ost, Y, Delta, compat="exact", perm=orc.perm_from_order(model.v_order(), K, T))
            ew = np.abs(model.W - ost["W"]).max() / np.abs(ost["W"]).max()
            ev = np.abs(model.V - ost["V"]).max() / np.abs(ost["V"]).max()
            assert ew < 1e-10 and ev < 1e-6, (form, it, ew, ev)
            model.W, model.V = ost["W"], ost["V"]
        assert model.likelihood_form() == form
        # (the host pushes of W / V above drop what was queued ahead: the first accumulation of a pair runs whole.  Now
        #  leave the state on the device: from the second half-sweep on every accumulation must be two launches)
        model._ctx.kernel_times()
        for it in range(2):
            np.random.seed(70 + it)
            model._resample_W(Y)
            model._resample_V(Y)
            np.random.seed(70 + it)
            orc.w_step(ost, Y)
            orc.v_step(ost, Y, Delta, compat="exact", perm=orc.perm_from_order(model.v_order(), K, T))
        kt = model._ctx.kernel_times()
        assert kt["w_accum"][1] == 4 and kt["v_accum"][1] == 4, kt       # W: whole + ahead, rest + ahead; V: ahead + rest, twice
        ew = np.abs(model.W - ost["W"]).max() / np.abs(ost["W"]).max()
        ev = np.abs(model.V - ost["V"]).max() / np.abs(ost["V"]).max()
        assert ew < 1e-9 and ev < 1e-5, (form, ew, ev)
        del model
    chains = []
    for overlap in (True, False):
        np.random.seed(7)
        m = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, sigma2_init=st0["sigma2"], lam2_init=st0["lam2"],
                                            nu2_init=st0["nu2"], W_init=st0["W"], V_init=st0["V"], compat="exact", shard=(rank, world),
                                            device=DEV, rng="device", device_seed=9, overlap_exchange=overlap)
        m._ctx.call("btf_set_tuning", 64, 64)
        for _ in range(4):
            m.resample(Yc)
        chains.append((m.W.copy(), m.V.copy(), float(m.nu2), float(m.sigma2), float(m.lam2)))
        del m
    a, b = chains                # (the own-block chunks have their own size: other partial sums, same totals up to rounding)
    assert np.abs(a[0] - b[0]).max() / np.abs(b[0]).max() < 1e-8 and np.abs(a[1] - b[1]).max() / np.abs(b[1]).max() < 1e-6
    assert all(abs(x - y) / abs(y) < 1e-8 for x, y in zip(a[2:], b[2:])), (a[2:], b[2:])


def reference_section(rank, world):
    """compat="reference" SHARDED (compat caveat; btf_set_shard_halo): the stale cached weights of quirks Q1/Q2
    (reference factor.py:320,349 and :394-401) come from a source row / column that may lie outside the rank's blocks - one
    more row / column of its slabs.  The sharded chains must reproduce the unsharded compat="reference" chains:
    Gaussian data whose missing curves repeat over groups of columns that straddle the shard borders (static counts in the
    halo column), Binomial data without a NaN (every row >= nembeds reads row nembeds-1's Polya-Gamma weights, every column
    those of column 0: both halos redrawn each sweep from the source's own cell streams), Binomial with missing cells,
    Negative-Binomial counts."""
    from functionalmf_amd.factor import BinomialBayesianTensorFiltering, NegativeBinomialBayesianTensorFiltering
    rs = np.random.RandomState(5)
    N, M, T, R, K, tf = 26, 10, 12, 2, 3, 1
    Wt = rs.normal(size=(N, K))
    Vt = 0.2 * np.cumsum(rs.normal(size=(M, T, K)), axis=1)
    Y = np.einsum("nk,mtk->nmt", Wt, Vt)[..., None] + rs.normal(0, 0.5, size=(N, M, T, R))
    group = np.array([0, 0, 0, 1, 1, 1, 1, 2, 2, 2])                  # columns sharing one pattern of missing curves
    pat = rs.rand(3, N, T) < 0.15
    Y[np.broadcast_to(pat[group].transpose(1, 0, 2)[..., None], Y.shape)] = np.nan
    W0, V0 = Wt + 0.1 * rs.normal(size=Wt.shape), Vt + 0.05 * rs.normal(size=Vt.shape)
    W0[np.triu_indices(N, 1, K)] = 0
    chains = []
    for shard in ((rank, world), None):
        np.random.seed(7)
        m = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=tf, sigma2_init=0.6, lam2_init=0.2, nu2_init=0.4, W_init=W0.copy(),
                                            V_init=V0.copy(), compat="reference", shard=shard, device=DEV, rng="device", device_seed=9,
                                            sampler="banded")
        for _ in range(3):
            m.resample(Y)
        if shard is not None and world == 2:
            assert (m._plan.halo_row, m._plan.halo_col) == ((-1, -1) if rank == 0 else (-1, 3)), (m._plan.halo_row, m._plan.halo_col)
        chains.append((m.W.copy(), m.V.copy(), float(m.nu2), float(m.sigma2), float(m.lam2)))
        del m
    a, b = chains
    assert np.abs(a[0] - b[0]).max() / np.abs(b[0]).max() < 1e-7 and np.abs(a[1] - b[1]).max() / np.abs(b[1]).max() < 1e-5, "gaussian"
    assert all(abs(x - y) / abs(y) < 1e-7 for x, y in zip(a[2:], b[2:])), (a[2:], b[2:])
    # the quirk is really in play: own weights (compat="exact") give another chain
    np.random.seed(7)
    m = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=tf, sigma2_init=0.6, lam2_init=0.2, nu2_init=0.4, W_init=W0.copy(),
                                        V_init=V0.copy(), compat="exact", shard=(rank, world), device=DEV, rng="device", device_seed=9,
                                        sampler="banded")
    for _ in range(3):
        m.resample(Y)
    assert np.abs(m.V - b[1]).max() / np.abs(b[1]).max() > 1e-3
    del m
    # ---- Binomial: without and with missing cells
    Nt = rs.randint(1, 40, size=(N, M, T)).astype(float)
    Ys = rs.binomial(Nt.astype(int), 1.0 / (1.0 + np.exp(-np.einsum("nk,mtk->nmt", Wt, Vt)))).astype(float)
    Ym, Nm = Ys.copy(), Nt.copy()
    holes = np.broadcast_to(pat[group].transpose(1, 0, 2), Ys.shape)
    Ym[holes] = np.nan
    Nm[holes] = np.nan
    for name, data in (("binomial", (Ys, Nt)), ("binomial+missing", (Ym, Nm))):
        chains = []
        for shard in ((rank, world), None):
            np.random.seed(7)
            m = BinomialBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=tf, sigma2_init=0.6, lam2_init=0.2, W_init=W0.copy(),
                                                V_init=V0.copy(), compat="reference", shard=shard, device=DEV, rng="device", device_seed=9)
            for _ in range(3):
                m.resample(data)
            if shard is not None and rank > 0 and name == "binomial":
                assert (m._plan.halo_row, m._plan.halo_col) == (K - 1, 0)
            chains.append((m.W.copy(), m.V.copy(), np.array(m.nu2).copy()))
            del m
        a, b = chains
        fin = np.isfinite(b[2])
        assert np.array_equal(fin, np.isfinite(a[2])) and np.abs(a[2][fin] - b[2][fin]).max() / np.abs(b[2][fin]).max() < 1e-9, name
        assert np.abs(a[0] - b[0]).max() / np.abs(b[0]).max() < 1e-7 and np.abs(a[1] - b[1]).max() / np.abs(b[1]).max() < 1e-5, name
    # host-fed weights (btf_set_omega with the halo row / column in the slabs): one W+V update from a given nu2 with host
    # normals, against the oracle's restatement of the reference's half-sweeps (quirks included)
    nu2 = 1.0 / np.random.RandomState(3).gamma(2.0, 0.2, size=(N, M, T))
    m = BinomialBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=tf, sigma2_init=0.6, lam2_init=0.2, W_init=W0.copy(),
                                        V_init=V0.copy(), compat="reference", shard=(rank, world), device=DEV)
    m.nu2 = nu2.copy()
    Delta = orc.trend_penalty(T, tf)
    ost = dict(W=W0.copy(), V=V0.copy(), Tau2=np.array(m.Tau2).copy(), lam2=0.2, sigma2=0.6, nu2=nu2.copy())
    np.random.seed(31)
    m._resample_W((Ys, Nt))
    m._resample_V((Ys, Nt))
    np.random.seed(31)
    orc.binomial_w_step(ost, Ys, Nt)
    orc.binomial_v_step(ost, Ys, Nt, Delta, compat="reference", perm=orc.perm_from_order(m.v_order(), K, T))
    ew, ev = np.abs(m.W - ost["W"]).max() / np.abs(ost["W"]).max(), np.abs(m.V - ost["V"]).max() / np.abs(ost["V"]).max()
    assert ew < 1e-9 and ev < 1e-6, ("host omega", ew, ev)
    del m
    # ---- Negative-Binomial counts
    cnts = rs.poisson(3.0, size=(N, M, T, 2)).astype(float)
    cnts[np.broadcast_to(holes[..., None], cnts.shape)] = np.nan
    chains = []
    for shard in ((rank, world), None):
        np.random.seed(7)
        m = NegativeBinomialBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=tf, sigma2_init=0.7, lam2_init=0.2, W_init=0.3 * W0,
                                                    V_init=0.3 * V0, rdims=(1, 2), nmetropolis=5, compat="reference", shard=shard, device=DEV,
                                                    rng="device", device_seed=9)
        for _ in range(2):
            m.resample(cnts)
        chains.append((m.W.copy(), m.V.copy(), np.array(m.R).copy()))
        del m
    a, b = chains
    assert np.abs(a[2] - b[2]).max() / np.abs(b[2]).max() < 1e-9, "R"
    assert np.abs(a[0] - b[0]).max() / np.abs(b[0]).max() < 1e-7 and np.abs(a[1] - b[1]).max() / np.abs(b[1]).max() < 1e-5, "negbinom"


def main():
    backend = os.environ.get("BTF_DIST_BACKEND", "gloo")
    exercise = os.environ.get("BTF_EXERCISE_EXCHANGE", "0")
    if backend == "nccl":
        import torch
        torch.cuda.set_device(DEV)
        dist.init_process_group("nccl", device_id=torch.device("cuda", DEV))
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    sections = os.environ.get("BTF_DIST_SECTION", "base,split").split(",")
    if "split" in sections and world > 1:
        split_section(rank, world)
    if "reference" in sections and world > 1:
        reference_section(rank, world)
    if "base" not in sections:
        print("SHARD_GPU_OK rank", rank, "exchange", TRANSPORT, flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    for name in ("g2_c2_complete.npz", "g1_c1_heldout.npz"):
        g = load_golden(name)
        N, M, T, R, K, tf = [int(x) for x in g["dims"]]
        st = state_from(g, "s0_")
        model = GaussianBayesianTensorFiltering(
            N, M, T, nembeds=K, tf_order=tf, sigma2_init=st["sigma2"], lam2_init=st["lam2"], nu2_init=st["nu2"],
            W_init=st["W"], V_init=st["V"], Tau2_init=st["Tau2"], compat="exact", shard=(rank, world), device=DEV, overlap_exchange=OVERLAP)
        assert model._exchange.active and model._plan.world == world
        # "nccl" process group: the exchange is the context's own RCCL communicator (C ABI); gloo: staged through the host;
        # BTF_EXCHANGE_TRANSPORT=peer: the library's peer-window transport (the ranks map each other's W / V: hipIpc)
        assert model._exchange.transport == TRANSPORT, model._exchange.transport
        if TRANSPORT == "rccl":
            info = model._exchange.comm_info()
            assert info["active"] == 1 and info["rank"] == rank and info["world"] == world and info["comm_count"] == world, info
        elif TRANSPORT == "peer":
            info = model._exchange.comm_info()
            assert info["active"] == 2 and info["rank"] == rank and info["world"] == world and info["gather_world"] == world, info
        Delta = orc.trend_penalty(T, tf)
        ost = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in st.items()}
        for it in range(2):
            np.random.seed(40 + it)                    # every rank draws the same host normals
            model._resample_W(g["Y"])
            model._resample_V(g["Y"])
            np.random.seed(40 + it)
            orc.w_step(ost, g["Y"])
            orc.v_step(ost, g["Y"], Delta, compat="exact", perm=orc.perm_from_order(model.v_order(), K, T))
            ew = np.abs(model.W - ost["W"]).max() / np.abs(ost["W"]).max()
            ev = np.abs(model.V - ost["V"]).max() / np.abs(ost["V"]).max()
            assert ew < 1e-10 and ev < 1e-6, (name, it, ew, ev)
            model.W, model.V = ost["W"], ost["V"]       # keep both chains on the same state
        np.random.seed(50)
        model._resample_nu2(g["Y"])
        sse, n = orc.sse_and_count(ost, g["Y"])
        np.random.seed(50)
        ref = 1.0 / np.random.gamma(0.1 + n / 2.0, 1.0 / (0.1 + sse / 2.0))
        assert abs(model.nu2 - ref) / ref < 1e-10, (model.nu2, ref)
    # ---- rng="device": whole sweeps with every scalar drawn on the GPU.  The Philox streams are keyed by global row /
    # column / cell indices and the residual sum of squares is all-reduced on the device, so the sharded chain must
    # reproduce the unsharded one (up to the rounding of differently grouped sums).
    from functionalmf_amd.factor import BinomialBayesianTensorFiltering
    for name in ("g2_c2_complete.npz", "g1_c1_heldout.npz"):
        g = load_golden(name)
        N, M, T, R, K, tf = [int(x) for x in g["dims"]]
        st = state_from(g, "s0_")
        chains = []
        for shard in ((rank, world), None):
            np.random.seed(7)
            os.environ["BTF_EXERCISE_EXCHANGE"] = exercise if shard is not None else "0"     # the plain chain: no collectives
            m = GaussianBayesianTensorFiltering(
                N, M, T, nembeds=K, tf_order=tf, sigma2_init=st["sigma2"], lam2_init=st["lam2"], nu2_init=st["nu2"],
                W_init=st["W"], V_init=st["V"], compat="exact", shard=shard, device=DEV, rng="device", device_seed=9, overlap_exchange=OVERLAP,
                # (held-out cells: a rank whose slabs happen to be complete would pick the spectral sampler where the
                #  unsharded run - one weighted tensor - uses the banded one: same distribution, another square root)
                sampler="auto" if name.startswith("g2") else "banded")
            assert m._dev_scalars and (m._exchange.active == (shard is not None))
            for _ in range(3):
                m.resample(g["Y"])
            chains.append((m.W.copy(), m.V.copy(), float(m.nu2), float(m.sigma2), float(m.lam2), np.array(m.Tau2).copy()))
            del m
        a, b = chains
        assert np.abs(a[0] - b[0]).max() / np.abs(b[0]).max() < 1e-7, name
        assert np.abs(a[1] - b[1]).max() / np.abs(b[1]).max() < 1e-5, name
        assert abs(a[2] - b[2]) / b[2] < 1e-9 and abs(a[3] - b[3]) / b[3] < 1e-9 and abs(a[4] - b[4]) / b[4] < 1e-7, (name, a[2:5], b[2:5])
        assert np.abs(a[5] - b[5]).max() / np.abs(b[5]).max() < 1e-5, name
    g = load_golden("g4_binomial_full.npz")
    N, M, T, R, K, tf = [int(x) for x in g["dims"]]
    st = state_from(g, "s0_")
    chains = []
    for shard in ((rank, world), None):
        np.random.seed(7)
        os.environ["BTF_EXERCISE_EXCHANGE"] = exercise if shard is not None else "0"
        m = BinomialBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=tf, sigma2_init=st["sigma2"], lam2_init=st["lam2"],
                                            W_init=st["W"], V_init=st["V"], compat="exact", shard=shard, device=DEV, rng="device",
                                            device_seed=9, overlap_exchange=OVERLAP)
        for _ in range(2):
            m.resample((g["Ysucc"], g["Ntrials"]))
        chains.append((m.W.copy(), m.V.copy(), np.array(m.nu2).copy()))
        del m
    a, b = chains
    assert np.abs(a[2] - b[2]).max() / np.abs(b[2]).max() < 1e-12          # the same Polya-Gamma draws, cell by cell
    assert np.abs(a[0] - b[0]).max() / np.abs(b[0]).max() < 1e-7 and np.abs(a[1] - b[1]).max() / np.abs(b[1]).max() < 1e-5
    # ---- Negative-Binomial counts sharded: every rank holds the whole count tensor (the rate update is a function of all of
    # it and is computed by every rank from the same Philox streams, like the other hyper-parameters); the augmented Binomial
    # model - pseudo-data, Polya-Gamma weights, W and V half-sweeps - lives on the rank's two slabs.  Rates per row (the
    # sharing pattern of examples/negbinom_tensor_filtering.py: the one-launch MH loop) and per cell (the stepwise loop).
    from functionalmf_amd.factor import NegativeBinomialBayesianTensorFiltering
    rs = np.random.RandomState(11)
    N, M, T, K, tf = 18, 7, 9, 3, 1
    cnts = rs.poisson(3.0, size=(N, M, T, 2)).astype(float)
    cnts[rs.rand(N, M, T, 2) < 0.1] = np.nan
    W0, V0 = 0.3 * rs.normal(size=(N, K)), 0.3 * rs.normal(size=(M, T, K))
    W0[np.triu_indices(N, 1, K)] = 0
    for rdims in ((1, 2), ()):
        chains = []
        for shard in ((rank, world), None):
            np.random.seed(7)
            os.environ["BTF_EXERCISE_EXCHANGE"] = exercise if shard is not None else "0"
            m = NegativeBinomialBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=tf, sigma2_init=0.7, lam2_init=0.2, W_init=W0.copy(),
                                                        V_init=V0.copy(), rdims=rdims, nmetropolis=5, compat="exact", shard=shard, device=DEV,
                                                        rng="device", device_seed=9, overlap_exchange=OVERLAP)
            for _ in range(2):
                m.resample(cnts)
            with np.errstate(divide="ignore"):
                om = np.where(np.isfinite(m.nu2), 1.0 / np.array(m.nu2), 0.0)      # (unobserved cells: weight 0, nu2 = inf)
            chains.append((m.W.copy(), m.V.copy(), np.array(m.R).copy(), om))
            del m
        a, b = chains
        assert np.abs(a[2] - b[2]).max() / np.abs(b[2]).max() < 1e-9, ("R", rdims)
        assert np.abs(a[3] - b[3]).max() / np.abs(b[3]).max() < 1e-9, ("omega", rdims)
        assert np.abs(a[0] - b[0]).max() / np.abs(b[0]).max() < 1e-7 and np.abs(a[1] - b[1]).max() / np.abs(b[1]).max() < 1e-5, rdims
    print("SHARD_GPU_OK rank", rank, "backend", dist.get_backend(), "world", dist.get_world_size(), "device", DEV,
          "exchange", {"rccl": "ctx-owned RCCL communicator", "peer": "peer windows", "host": "host-staged"}[TRANSPORT], flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
