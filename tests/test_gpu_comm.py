"""The ctx-owned communicator of include/btf.h (btf_comm_* / btf_allgather_* / btf_allreduce_*): RCCL called by the
library itself on the context's buffers and stream - what SURVEY 8(b) means by "ctx owns ... RCCL communicators".
Everything here goes through the raw C ABI with NO torch.distributed process group: the communicator id is made and
consumed in one process (a one-rank communicator is all a one-GPU box can hold - RCCL refuses two ranks on one
device); the N-rank form of the same calls runs in test_rccl_ranks_on_their_own_gpus on a box with several GPUs.
Needs an MI355X."""
import ctypes as C

import numpy as np
import pytest
from conftest import load_golden, relerr, state_from

pytestmark = pytest.mark.gpu


def _gaussian_ctx(g, st, shard=None, dev_scalars=False):
    """A context over the fixture's data: whole tensor, or the two slabs of rank shard[0] of shard[1] (equal chunks)."""
    from functionalmf_amd import _native
    N, M, T, R, K, tf = [int(x) for x in g["dims"]]
    ctx = _native.Context(N, M, T, K, tf)
    Y = np.ascontiguousarray(g["Y"], dtype=np.float64).reshape(N, M, T, R)
    rows = cols = Y
    if shard is not None:
        lo, ln = C.c_int32(), C.c_int32()
        blk = []
        for n in (N, M):
            assert ctx.lib.btf_comm_block(n, shard[0], shard[1], C.byref(lo), C.byref(ln)) == 0
            blk += [lo.value, ln.value]
        ctx.call("btf_set_shard", *blk)
        rows = np.ascontiguousarray(Y[blk[0]:blk[0] + blk[1]])
        cols = np.ascontiguousarray(Y[:, blk[2]:blk[2] + blk[3]])
    if dev_scalars:
        ctx.call("btf_device_scalars", 1)
    ctx.call("btf_set_data_gaussian", _native.dptr(rows), _native.dptr(cols), R)
    ctx.call("btf_set_W", _native.dptr(_native.as_f64(st["W"])))
    ctx.call("btf_set_V", _native.dptr(_native.as_f64(st["V"])))
    ctx.call("btf_set_hyper", _native.dptr(_native.as_f64(st["Tau2"])), float(st["lam2"]), float(st["sigma2"]))
    ctx.call("btf_set_nu2", float(st["nu2"]))
    if dev_scalars:
        ctx.call("btf_set_scalars", float(st["nu2"]), float(st["sigma2"]), float(st["lam2"]), 1.0)
    return ctx, (N, M, T, R, K, tf)


def _one_rank_comm(ctx):
    from functionalmf_amd import _native
    ident = (C.c_ubyte * _native.COMM_ID_BYTES)()
    assert ctx.lib.btf_comm_unique_id(ident, _native.COMM_ID_BYTES) == 0, ctx.lib.btf_last_error(None)
    assert any(ident)
    ctx.call("btf_comm_init", 0, 1, ident, _native.COMM_ID_BYTES)
    return ident


def _info(ctx):
    out = (C.c_int32 * 8)()
    ctx.call("btf_comm_info", out)
    return list(out)


def test_one_rank_communicator_walks_the_plain_chain():
    """btf_comm_unique_id -> btf_comm_init(0 of 1) -> after every half-sweep the all-gather of the sharded step, and the
    nu2 draw split around btf_allreduce_sse (btf_draw_scalars which | 8, | 16): the chain must equal the one of a
    context that never heard of a communicator, bit for bit - an all-gather or a draw kernel running out of stream order,
    or a gather that moved anything in a one-rank world, would change it.  Both orderings of the exchange: in line, and
    on the context's communication stream (BTF_OPT_SPLIT_ACCUM)."""
    from functionalmf_amd import _native
    import torch
    import torch.distributed as dist
    assert not dist.is_initialized()
    g = load_golden("g2_c2_complete.npz")
    st = state_from(g, "s0_")
    chains = []
    for comm, split in ((False, 0), (True, 0), (True, 1)):
        ctx, (N, M, T, R, K, tf) = _gaussian_ctx(g, st, dev_scalars=True)
        if comm:
            _one_rank_comm(ctx)
            info = _info(ctx)
            assert info[:6] == [1, 0, 1, 0, 1, 0] and info[6] > 20000 and info[7] == 1, info      # RCCL version code, ncclCommCount
            ctx.call("btf_set_option", _native.OPT_SPLIT_ACCUM, split)
        for it in range(3):
            seed = 100 + 10 * it
            if comm:
                ctx.call("btf_draw_scalars", seed, 1 | 8, 0.1, 0.1, 0.1, 0.1)
                ctx.call("btf_allreduce_sse")
                ctx.call("btf_draw_scalars", seed, 1 | 16, 0.1, 0.1, 0.1, 0.1)
            else:
                ctx.call("btf_draw_scalars", seed, 1, 0.1, 0.1, 0.1, 0.1)
            ctx.call("btf_resample_W", None, seed + 1, _native.COMPAT["exact"])
            if comm:
                ctx.call("btf_allgather_W")
            ctx.call("btf_resample_V", None, seed + 2, _native.COMPAT["exact"], 1e-6, 4)
            if comm:
                ctx.call("btf_allgather_V")
        W, V, sc = np.empty((N, K)), np.empty((M, T, K)), np.zeros(6)
        ctx.call("btf_get_W", _native.dptr(W))
        ctx.call("btf_get_V", _native.dptr(V))
        ctx.call("btf_get_scalars", _native.dptr(sc))
        ctx.call("btf_sync")
        chains.append((W, V, sc[:4].copy()))
        if comm:
            vals = np.array([1.5, -2.0, 3.25])
            ctx.call("btf_allreduce_sum", _native.dptr(vals), 3)          # one rank: the sum is the value
            assert vals.tolist() == [1.5, -2.0, 3.25]
            ctx.call("btf_comm_destroy")
            assert _info(ctx)[0] == 0
        ctx.close()
    assert np.all(np.isfinite(chains[0][0])) and np.all(np.isfinite(chains[0][1]))
    for other in chains[1:]:
        assert np.array_equal(chains[0][0], other[0]) and np.array_equal(chains[0][1], other[1])
        assert np.array_equal(chains[0][2], other[2])
    torch.cuda.synchronize()


def test_rehearsed_rank_moves_the_full_messages_without_a_process_group():
    """btf_comm_rehearse(3 of 8): the context plays rank 3's slabs, its all-gathers move the whole gathered W / V between
    scratch buffers on a one-rank communicator.  The rank's own blocks must come out as the same half-sweeps without any
    exchange give them (the scratch traffic touches nothing), the other ranks' blocks stay as they were."""
    from functionalmf_amd import _native
    import torch.distributed as dist
    assert not dist.is_initialized()
    g = load_golden("g2_c2_complete.npz")
    st = state_from(g, "s0_")
    out = []
    for rehearse in (False, True):
        ctx, (N, M, T, R, K, tf) = _gaussian_ctx(g, st, shard=(3, 8))
        if rehearse:
            ctx.call("btf_comm_rehearse", 3, 8)
            assert _info(ctx)[:6] == [1, 0, 1, 3, 8, 1]
        for it in range(2):
            ctx.call("btf_resample_W", None, 7 + it, _native.COMPAT["exact"])
            if rehearse:
                ctx.call("btf_allgather_W")
            ctx.call("btf_resample_V", None, 17 + it, _native.COMPAT["exact"], 1e-6, 4)
            if rehearse:
                ctx.call("btf_allgather_V")
        W, V = np.empty((N, K)), np.empty((M, T, K))
        ctx.call("btf_get_W", _native.dptr(W))
        ctx.call("btf_get_V", _native.dptr(V))
        ctx.call("btf_sync")
        out.append((W, V))
        ctx.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    chunk = -(-64 // 8)
    assert not np.array_equal(out[0][0][3 * chunk:4 * chunk], np.asarray(st["W"])[3 * chunk:4 * chunk])      # the own rows moved
    assert np.array_equal(np.delete(out[0][0], np.s_[3 * chunk:4 * chunk], axis=0), np.delete(np.asarray(st["W"]), np.s_[3 * chunk:4 * chunk], axis=0))


def test_model_rehearsal_needs_no_process_group():
    """The Python surface of the same thing: shard=(rank, world), rehearse_rank=True builds its communicator through
    btf_comm_rehearse - bench.py --as-rank runs without torch.distributed."""
    import torch.distributed as dist
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    assert not dist.is_initialized()
    g = load_golden("g2_c2_complete.npz")
    N, M, T, R, K, tf = [int(x) for x in g["dims"]]
    st = state_from(g, "s0_")
    m = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=tf, sigma2_init=st["sigma2"], lam2_init=st["lam2"], nu2_init=st["nu2"],
                                        W_init=st["W"], V_init=st["V"], compat="exact", shard=(1, 4), rng="device", device_seed=3,
                                        rehearse_rank=True)
    assert m._exchange.transport == "rccl"
    info = m._exchange.comm_info()
    assert info["rehearsal"] == 1 and info["gather_rank"] == 1 and info["gather_world"] == 4 and info["world"] == 1
    for _ in range(3):
        m.resample(g["Y"])
    m.sync()
    assert np.all(np.isfinite(m.W)) and np.all(np.isfinite(m.V))


def test_comm_entry_points_refuse_what_they_cannot_do():
    """Status codes, not faults: collectives without a communicator (BTF_ESTATE), blocks that are not the equal chunks one
    in-place all-gather reassembles (BTF_ESTATE), a communicator id of the wrong size or a rank outside the world
    (BTF_EINVAL), more than 64 ranks (the padding of W / V)."""
    from functionalmf_amd import _native
    g = load_golden("g2_c2_complete.npz")
    st = state_from(g, "s0_")
    ctx, (N, M, T, R, K, tf) = _gaussian_ctx(g, st)
    lib, h = ctx.lib, ctx.h
    for name in ("btf_allgather_W", "btf_allgather_V", "btf_allreduce_sse"):
        assert getattr(lib, name)(h) == _native.BTF_ESTATE, name
    vals = np.zeros(2)
    assert lib.btf_allreduce_sum(h, _native.dptr(vals), 2) == _native.BTF_ESTATE
    ident = (C.c_ubyte * 128)()
    assert lib.btf_comm_unique_id(ident, 64) == _native.BTF_EINVAL
    assert lib.btf_comm_unique_id(ident, 128) == 0
    assert lib.btf_comm_init(h, 0, 1, ident, 64) == _native.BTF_EINVAL
    assert lib.btf_comm_init(h, 1, 1, ident, 128) == _native.BTF_EINVAL
    assert lib.btf_comm_init(h, 0, 65, ident, 128) == _native.BTF_EINVAL
    assert lib.btf_comm_rehearse(h, 8, 8) == _native.BTF_EINVAL
    assert lib.btf_comm_init(h, 0, 1, ident, 128) == 0
    assert lib.btf_allreduce_sum(h, _native.dptr(vals), 17) == _native.BTF_EINVAL
    assert lib.btf_allreduce_sse(h) == _native.BTF_ESTATE          # no device-resident scalars in this context
    assert lib.btf_allgather_W(h) == 0
    ctx.close()
    # a shard that is not rank r's equal chunk: the gather would scatter rows over the wrong offsets
    ctx = _native.Context(N, M, T, K, tf)
    ctx.call("btf_set_shard", 3, 10, 0, M)
    Y = np.ascontiguousarray(g["Y"], dtype=np.float64).reshape(N, M, T, R)
    ctx.call("btf_set_data_gaussian", _native.dptr(np.ascontiguousarray(Y[3:13])), _native.dptr(Y), R)
    ctx.call("btf_set_W", _native.dptr(_native.as_f64(st["W"])))
    ctx.call("btf_comm_rehearse", 1, 4)
    assert ctx.lib.btf_allgather_W(ctx.h) == _native.BTF_ESTATE
    assert b"equal chunks" in ctx.lib.btf_last_error(ctx.h)
    lo, ln = C.c_int32(), C.c_int32()
    got = []
    for r in range(5):
        assert ctx.lib.btf_comm_block(10, r, 5, C.byref(lo), C.byref(ln)) == 0
        got.append((lo.value, ln.value))
    assert got == [(0, 2), (2, 2), (4, 2), (6, 2), (8, 2)]
    got = []
    for r in range(4):
        assert ctx.lib.btf_comm_block(9, r, 4, C.byref(lo), C.byref(ln)) == 0
        got.append((lo.value, ln.value))
    assert got == [(0, 3), (3, 3), (6, 3), (9, 0)]                 # the tail rank is empty: ceil(9 / 4) = 3
    assert ctx.lib.btf_comm_block(9, 4, 4, C.byref(lo), C.byref(ln)) == _native.BTF_EINVAL
    ctx.close()


def test_device_likelihood_entry_points_reject_the_host_family():
    """BTF_ESS_HOST_LIKELIHOOD (-1) is for btf_ess_begin / btf_ess_eval only: btf_ess_run and every btf_gass_* entry
    point launch device likelihood kernels and must answer BTF_EINVAL instead of handing family -1 to them."""
    from functionalmf_amd import _native
    g = load_golden("g2_c2_complete.npz")
    st = state_from(g, "s0_")
    ctx, (N, M, T, R, K, tf) = _gaussian_ctx(g, st)
    lib, h = ctx.lib, ctx.h
    host = _native.ESS_HOST_LIKELIHOOD
    assert lib.btf_ess_run(h, 0, host, 1, None, 1, 8, 1e-6, 4) == _native.BTF_EINVAL
    assert lib.btf_ess_run(h, 1, host, 0, None, 1, 8, 1e-6, 4) == _native.BTF_EINVAL
    z, u = np.zeros(N * K), np.full(N, 0.5)
    assert lib.btf_gass_begin(h, 0, host, _native.dptr(z), _native.dptr(u), 1, 1e-6, 0, 0) == _native.BTF_EINVAL
    assert lib.btf_gass_run(h, 0, host, 1, 16, 1e-6, 0) == _native.BTF_EINVAL
    assert b"btf_ess_begin / btf_ess_eval only" in lib.btf_last_error(h)
    # ... while the two that are documented to take it still do
    assert lib.btf_ess_begin(h, 0, None, 5, 1e-6, 4) == 0
    ll = C.c_double(1.0)
    assert lib.btf_ess_eval(h, 0, 0.3, 0, host, C.byref(ll)) == 0 and ll.value == 0.0
    ctx.close()


def _peer_group(ctxs):
    """btf_peer_export on every context, the descriptors side by side, btf_peer_init on every context: what MPI_Allgather
    (or Exchange._init_peer) does between processes, here between the contexts of one."""
    from functionalmf_amd import _native
    nb, world = _native.PEER_DESC_BYTES, len(ctxs)
    descs = (C.c_ubyte * (nb * world))()
    for r, ctx in enumerate(ctxs):
        one = (C.c_ubyte * nb)()
        ctx.call("btf_peer_export", one, nb)
        descs[r * nb:(r + 1) * nb] = list(one)
    for r, ctx in enumerate(ctxs):
        ctx.call("btf_peer_init", r, world, descs, nb * world)
        assert _info(ctx)[:6] == [2, r, world, r, world, 0]


@pytest.mark.parametrize("world,split", [(2, 0), (3, 0)])
def test_peer_windows_run_the_sharded_chain_between_contexts(world, split):
    """The peer-window transport (btf_peer_export / btf_peer_init): `world` contexts - here in one process, each on its own
    stream, mapped into each other by raw pointers - hold the slabs of ranks 0 .. world-1 and run the sharded step with the
    library's own collectives: after every half-sweep each rank's exchange kernel stores its block into the other ranks'
    W / V and waits for theirs; the nu2 draw sums the per-rank residuals through the mailboxes.  Every rank must end
    with the same replicated W, V and scalars, and they must be the unsharded chain's (same device seeds; the only
    difference is the order of the residual sum).  (Few streams at a time: the runtime gives a process 4 hardware
    queues, and a waiting exchange kernel holds its own - btf_comm.h; the process-per-rank form of this test, with the
    gathers on the communication streams too, is test_peer_windows_between_processes_on_one_gpu.)"""
    from functionalmf_amd import _native
    g = load_golden("g2_c2_complete.npz")
    st = state_from(g, "s0_")
    ref, dims = _gaussian_ctx(g, st, dev_scalars=True)
    N, M, T, R, K, tf = dims

    def state(ctx):
        W, V, sc = np.empty((N, K)), np.empty((M, T, K)), np.zeros(6)
        ctx.call("btf_get_W", _native.dptr(W))
        ctx.call("btf_get_V", _native.dptr(V))
        ctx.call("btf_get_scalars", _native.dptr(sc))
        ctx.call("btf_sync")
        return W, V, sc[:4].copy()

    for it in range(4):
        seed = 300 + 10 * it
        ref.call("btf_draw_scalars", seed, 1, 0.1, 0.1, 0.1, 0.1)
        ref.call("btf_resample_W", None, seed + 1, _native.COMPAT["exact"])
        ref.call("btf_resample_V", None, seed + 2, _native.COMPAT["exact"], 1e-6, 4)
    W0, V0, s0 = state(ref)
    ref.close()
    assert np.all(np.isfinite(W0)) and np.all(np.isfinite(V0))
    ctxs = [_gaussian_ctx(g, st, shard=(r, world), dev_scalars=True)[0] for r in range(world)]
    _peer_group(ctxs)
    Y = np.asarray(g["Y"]).reshape(N, M, T, R)
    for ctx in ctxs:
        ctx.call("btf_set_option", _native.OPT_SPLIT_ACCUM, split)
        ctx.call("btf_set_global_nobs", float(np.isfinite(Y).sum()))      # the nu2 draw's count is over all ranks
    for it in range(4):
        seed = 300 + 10 * it
        for ctx in ctxs:
            ctx.call("btf_draw_scalars", seed, 1 | 8, 0.1, 0.1, 0.1, 0.1)
        for ctx in ctxs:
            ctx.call("btf_allreduce_sse")
        for ctx in ctxs:
            ctx.call("btf_draw_scalars", seed, 1 | 16, 0.1, 0.1, 0.1, 0.1)
        for ctx in ctxs:
            ctx.call("btf_resample_W", None, seed + 1, _native.COMPAT["exact"])
        for ctx in ctxs:
            ctx.call("btf_allgather_W")
        for ctx in ctxs:
            ctx.call("btf_resample_V", None, seed + 2, _native.COMPAT["exact"], 1e-6, 4)
        for ctx in ctxs:
            ctx.call("btf_allgather_V")

    got = [state(ctx) for ctx in ctxs]
    for W, V, sc in got[1:]:
        assert np.array_equal(W, got[0][0]) and np.array_equal(V, got[0][1]) and np.array_equal(sc, got[0][2])     # replicas agree bit for bit
    assert relerr(got[0][0], W0) < 1e-9 and relerr(got[0][1], V0) < 1e-9 and np.allclose(got[0][2], s0, rtol=1e-9)
    vals = [np.array([1.0 + r, -2.0 * r, 0.5]) for r in range(world)]
    # btf_allreduce_sum synchronises: every rank's call must be on its way before the first one waits -> threads
    import threading
    th = [threading.Thread(target=lambda c=c, v=v: c.call("btf_allreduce_sum", _native.dptr(v), 3)) for c, v in zip(ctxs, vals)]
    for t in th:
        t.start()
    for t in th:
        t.join(60)
    want = [sum(1.0 + r for r in range(world)), sum(-2.0 * r for r in range(world)), 0.5 * world]
    for v in vals:
        assert v.tolist() == want
    for ctx in ctxs:
        ctx.call("btf_comm_destroy")
        assert _info(ctx)[0] == 0
        ctx.close()


def test_peer_exchange_with_a_missing_rank_times_out_instead_of_hanging(monkeypatch):
    """A rank whose peer never issues the collective: the exchange kernel gives up after BTF_PEER_TIMEOUT_MS and the
    next status check reports BTF_EHIP naming the rank - the GPU is not left spinning."""
    from functionalmf_amd import _native
    monkeypatch.setenv("BTF_PEER_TIMEOUT_MS", "200")
    g = load_golden("g2_c2_complete.npz")
    st = state_from(g, "s0_")
    ctxs = [_gaussian_ctx(g, st, shard=(r, 2))[0] for r in range(2)]
    _peer_group(ctxs)
    ctxs[0].call("btf_resample_W", None, 5, _native.COMPAT["exact"])
    ctxs[0].call("btf_allgather_W")                      # rank 1 never comes
    with pytest.raises(_native.BTFError) as e:
        ctxs[0].call("btf_resample_V", None, 6, _native.COMPAT["exact"], 1e-6, 4)
        ctxs[0].call("btf_sync")
    assert "rank 1" in str(e.value) and "timed out" in str(e.value)
    for ctx in ctxs:
        ctx.close()
