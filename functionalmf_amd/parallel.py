"""Row / column sharding of the two half-sweeps over the GPUs of one node.

Given V the rows of W are conditionally independent (factor.py:333) and given W
the columns of V are (factor.py:378), so rank p updates a contiguous block of
rows in the W half-sweep and a contiguous block of columns in the V half-sweep,
from two slabs of the sufficient statistics (its rows x everything, everything
x its columns).  W and V are replicated; the only exchange per half-sweep is
one all-gather of the freshly drawn block: RCCL over xGMI, issued by the library
itself on the context's communicator (btf_allgather_W / btf_allgather_V of
include/btf.h), or any torch.distributed backend on host copies - used by the
CPU tests with gloo).
"""
import numpy as np


def _block(n, rank, world):
    """Contiguous block [start, start+len) of an axis of length n; equal chunk
    ceil(n/world) (the tail ranks may be short or empty) so that an in-place
    all-gather with a fixed chunk size reassembles the axis."""
    chunk = -(-n // world)
    lo = min(rank * chunk, n)
    hi = min(lo + chunk, n)
    return lo, hi - lo, chunk


class ShardPlan:
    MAX_WORLD = 64      # the context pads W / V by 64 rows / columns for the in-place all-gather (btf_create)

    def __init__(self, nrows, ncols, rank=0, world=1):
        if not (0 <= rank < world):
            raise ValueError("rank %d outside world of %d" % (rank, world))
        if world > self.MAX_WORLD:
            raise ValueError("at most %d shards (the device buffers are padded for that many)" % self.MAX_WORLD)
        self.nrows, self.ncols, self.rank, self.world = nrows, ncols, rank, world
        self.row0, self.nl, self.row_chunk = _block(nrows, rank, world)
        self.col0, self.ml, self.col_chunk = _block(ncols, rank, world)
        # compat="reference": the one stale-weight source row / column outside the blocks (-1: none), carried as one more
        # row / column of every slab (btf_set_shard_halo)
        self.halo_row = self.halo_col = -1

    def halo_of(self, src_row, src_col):
        """The source row and the source column (global indices, -1: none) that the weights of this rank's rows / columns
        come from but that lie outside its blocks.  The row sources of the reference are nembeds-1 or the row itself
        (factor.py:320,349), its column sources never decrease and are a column's own index or its left neighbour's source
        (factor.py:394-401): at most one of each per contiguous block."""
        out = []
        for src, lo, n in ((src_row, self.row0, self.nl), (src_col, self.col0, self.ml)):
            s = np.asarray(src[lo:lo + n])
            far = np.unique(s[(s < lo) | (s >= lo + n)])
            if far.size > 1:
                raise ValueError("more than one stale-weight source outside a shard (%r): not the reference's source pattern" % (far,))
            out.append(int(far[0]) if far.size else -1)
        return tuple(out)

    def mine(self):
        return self.row0, self.nl, self.col0, self.ml

    def slabs(self, Y4):
        """Row slab Y[row0:row0+nl] and column slab Y[:, col0:col0+ml] as contiguous
        float64 arrays (the same object twice when unsharded); with a halo source row / column, that row / column
        appended."""
        if self.world == 1:
            a = np.ascontiguousarray(Y4, dtype=np.float64)
            return a, a
        ri = np.arange(self.row0, self.row0 + self.nl)
        ci = np.arange(self.col0, self.col0 + self.ml)
        if self.halo_row >= 0:
            ri = np.append(ri, self.halo_row)              # the halo source LAST (btf_set_shard_halo)
        if self.halo_col >= 0:
            ci = np.append(ci, self.halo_col)
        rows = np.ascontiguousarray(Y4[ri], dtype=np.float64)
        cols = np.ascontiguousarray(Y4[:, ci], dtype=np.float64)
        return rows, cols

    def assemble(self, blocks, axis_len):
        """Concatenate per-rank blocks (each padded to the common chunk along axis 0)
        back into the full axis."""
        full = np.concatenate(blocks, axis=0)
        return full[:axis_len]


class _DevView:
    """__cuda_array_interface__ holder so torch can wrap a raw device pointer."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class Exchange:
    """The per-half-sweep exchange.  world == 1: no-ops.

    transport "rccl" (the default whenever the ranks have a GPU each): the collectives are the library's own -
    btf_allgather_W / btf_allgather_V / btf_allreduce_sse of include/btf.h, RCCL calls on the context's communicator,
    buffers and stream (csrc/btf_comm.h).  torch.distributed is then only the channel that carries rank 0's 128-byte
    communicator id to the other ranks (any backend; what MPI_Bcast is to an NCCL program) - a C caller uses its own.
    transport "peer": the same four calls of the C ABI on the library's peer-window transport (btf_peer_export /
    btf_peer_init, csrc/btf_comm.h): every rank maps the other ranks' W / V buffers (hipIpc) and one kernel per collective
    stores this rank's block straight into them.  The process group carries the 256-byte descriptors once.  The default
    when the ranks of a "gloo" group have a device context each (several ranks on ONE GPU, where RCCL refuses to run).
    transport "host": the exchange staged through host copies over torch.distributed (any backend) - what the CPU tests
    (gloo) use."""

    def __init__(self, plan, ctx=None, group=None, overlap=True, rehearse=False, transport=None):
        self.plan, self.ctx, self.group = plan, ctx, group
        self.overlap = overlap          # all-gathers on the context's communication stream (BTF_OPT_SPLIT_ACCUM)
        # rehearse: ONE process plays rank `plan.rank` of `plan.world` (bench.py --as-rank): its kernels run on that
        # rank's real slabs, and every collective of the sharded step is issued on a one-rank communicator over scratch
        # buffers of the full message size (btf_comm_rehearse) - the blocks of the other ranks in W / V are never
        # refreshed, so the chain is not a sampler of anything; only the clock is read
        self.rehearse = rehearse
        self._tstream = None
        self.timing, self._events = False, {"all_gather_W": [], "all_gather_V": [], "all_reduce_sse": []}
        # BTF_EXERCISE_EXCHANGE=1: issue the collectives even in a 1-rank group (lets a 1-GPU box
        # run the exact RCCL call sequence of the sharded path)
        import os
        self.active = plan.world > 1 or rehearse or os.environ.get("BTF_EXERCISE_EXCHANGE", "0") == "1"
        self.transport = None
        if not self.active:
            return
        if rehearse:
            if ctx is None:
                raise RuntimeError("a rehearsal needs a device context")
            self.transport = "rccl"
            ctx.call("btf_comm_rehearse", plan.rank, plan.world)
            return
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("shard=(rank, world) needs an initialised torch.distributed process group "
                               "(the channel the communicator id travels over)")
        if dist.get_world_size(group) != plan.world or dist.get_rank(group) != plan.rank:
            raise RuntimeError("shard does not match the process group")
        want = transport or os.environ.get("BTF_EXCHANGE_TRANSPORT") or \
            ("rccl" if dist.get_backend(group) == "nccl" else "host")
        if want not in ("rccl", "peer", "host"):
            raise ValueError("transport must be 'rccl', 'peer' or 'host'")
        self.transport = want if ctx is not None else "host"
        if self.transport == "rccl":
            self._init_comm()
        elif self.transport == "peer":
            self._init_peer()

    # -- device path (RCCL under the C ABI) ----------------------------------------------
    def _init_comm(self):
        """btf_comm_unique_id on rank 0 -> broadcast over the process group -> btf_comm_init on every rank."""
        import ctypes as C
        import torch
        import torch.distributed as dist
        from . import _native
        nb = _native.COMM_ID_BYTES
        ident = (C.c_ubyte * nb)()
        if self.plan.rank == 0:
            rc = self.ctx.lib.btf_comm_unique_id(ident, nb)
            if rc != _native.BTF_OK:
                raise _native.BTFError(rc, self.ctx.lib.btf_last_error(None).decode())
        on_gpu = dist.get_backend(self.group) == "nccl"
        t = torch.tensor(list(ident), dtype=torch.uint8, device=torch.device("cuda", self.ctx.device) if on_gpu else "cpu")
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        dist.broadcast(t, src=src, group=self.group)
        ident = (C.c_ubyte * nb)(*t.cpu().tolist())
        self.ctx.call("btf_comm_init", self.plan.rank, self.plan.world, ident, nb)

    def _init_peer(self):
        """btf_peer_export on every rank -> all-gather of the descriptors over the process group -> btf_peer_init."""
        import ctypes as C
        import torch
        import torch.distributed as dist
        from . import _native
        nb, p = _native.PEER_DESC_BYTES, self.plan
        desc = (C.c_ubyte * nb)()
        self.ctx.call("btf_peer_export", desc, nb)
        on_gpu = dist.get_backend(self.group) == "nccl"
        dev = torch.device("cuda", self.ctx.device) if on_gpu else "cpu"
        mine = torch.tensor(list(desc), dtype=torch.uint8, device=dev)
        out = [torch.empty_like(mine) for _ in range(p.world)]
        dist.all_gather(out, mine, group=self.group)
        flat = [b for t in out for b in t.cpu().tolist()]
        descs = (C.c_ubyte * (nb * p.world))(*flat)
        self.ctx.call("btf_peer_init", p.rank, p.world, descs, nb * p.world)
        dist.barrier(group=self.group)          # every rank has mapped every mailbox before the first collective

    def comm_info(self):
        """{has communicator, rank, world, gather rank, gather world, rehearsal, RCCL version, ncclCommCount} of the
        context's communicator (btf_comm_info)."""
        import ctypes as C
        out = (C.c_int32 * 8)()
        self.ctx.call("btf_comm_info", out)
        return dict(zip(("active", "rank", "world", "gather_rank", "gather_world", "rehearsal", "rccl_version", "comm_count"), out))

    def device_views(self):
        """The context's W / V device buffers as torch tensors over the SAME memory (btf_dev_W / btf_dev_V, including the
        padding the in-place all-gather uses): for a caller that prefers its own collectives to btf_allgather_W / _V - it
        must issue them on the context's stream (btf_stream) or order its stream against it."""
        import torch
        p, (N, M, T, K, _) = self.plan, self.ctx.dims
        lib = self.ctx.lib
        dev = torch.device("cuda", self.ctx.device)
        Wt = torch.as_tensor(_DevView(lib.btf_dev_W(self.ctx.h), (p.world * p.row_chunk * K,)), device=dev)
        Vt = torch.as_tensor(_DevView(lib.btf_dev_V(self.ctx.h), (p.world * p.col_chunk * T * K,)), device=dev)
        return Wt, Vt

    def _stream(self):
        """The ctx's HIP stream as a torch stream (event timing of the collectives only)."""
        import torch
        if self._tstream is None:
            self._tstream = torch.cuda.ExternalStream(self.ctx.stream_handle, device=torch.device("cuda", self.ctx.device))
        return self._tstream

    def _timed(self, name, fn):
        if not self.timing:
            return fn()
        import torch
        with torch.cuda.stream(self._stream()):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            self._events[name].append((e0, e1))

    def collective_us(self):
        """Mean microseconds per collective since timing was switched on (synchronises)."""
        import torch
        out = {}
        if any(self._events.values()):
            torch.cuda.synchronize()
        for k, evs in self._events.items():
            if evs:
                out[k] = round(1e3 * sum(a.elapsed_time(b) for a, b in evs) / len(evs), 2)
            self._events[k] = []
        return out

    def _staged(self):
        """True when the exchange goes through host copies (transport "host")."""
        return self.transport not in ("rccl", "peer")

    def _staged_gather(self, getter, setter, shape, block0, blocklen, gather):
        import ctypes as C
        full = np.empty(shape)
        dp = C.POINTER(C.c_double)
        self.ctx.call(getter, full.ctypes.data_as(dp))
        full = np.ascontiguousarray(gather(full[block0:block0 + blocklen]))
        self.ctx.call(setter, full.ctypes.data_as(dp))

    def after_W(self):
        if not self.active:
            return
        if self._staged():
            N, M, T, K, _ = self.ctx.dims
            return self._staged_gather("btf_get_W", "btf_set_gathered_W", (N, K), self.plan.row0, self.plan.nl, self.gather_rows_host)
        self._timed("all_gather_W", lambda: self.ctx.call("btf_allgather_W"))

    def after_V(self):
        if not self.active:
            return
        if self._staged():
            N, M, T, K, _ = self.ctx.dims
            return self._staged_gather("btf_get_V", "btf_set_gathered_V", (M, T, K), self.plan.col0, self.plan.ml, self.gather_cols_host)
        self._timed("all_gather_V", lambda: self.ctx.call("btf_allgather_V"))

    def all_reduce_sse(self):
        """Sum the rank-local residual sum of squares (device scalar slot 4, btf_dev_hyp) over the ranks, in place,
        on the ctx's stream: the one exchange of a sharded nu2 draw (include/btf.h, btf_draw_scalars which | 8 / | 16)."""
        if not self.active:
            return
        if self._staged():
            import ctypes as C
            out = np.zeros(6)
            self.ctx.call("btf_get_scalars", out.ctypes.data_as(C.POINTER(C.c_double)))
            (tot,) = self.sum_scalars(float(out[4]))
            self.ctx.call("btf_set_scalar_slot", 4, float(tot))
            return
        self._timed("all_reduce_sse", lambda: self.ctx.call("btf_allreduce_sse"))

    def sum_scalars(self, *vals):
        if not self.active:
            return vals
        if not self._staged():
            import ctypes as C
            buf = np.array(vals, dtype=np.float64)
            self.ctx.call("btf_allreduce_sum", buf.ctypes.data_as(C.POINTER(C.c_double)), len(vals))
            return tuple(buf.tolist())
        import torch
        import torch.distributed as dist
        dev = torch.device("cuda", self.ctx.device) if (dist.get_backend(self.group) == "nccl" and self.ctx is not None) else \
            ("cuda" if dist.get_backend(self.group) == "nccl" else "cpu")
        t = torch.tensor(vals, dtype=torch.float64, device=dev)
        dist.all_reduce(t, group=self.group)
        return tuple(t.tolist())

    # -- host path (any backend; CPU tests) --------------------------------------------
    def gather_rows_host(self, block):
        """All-gather a per-rank block of rows held in host memory; returns the full axis."""
        p = self.plan
        if p.world == 1:
            return block
        import torch
        import torch.distributed as dist
        pad = np.zeros((p.row_chunk,) + block.shape[1:])
        pad[:block.shape[0]] = block
        mine = torch.from_numpy(pad)
        if dist.get_backend(self.group) == "nccl":
            mine = mine.cuda()
        out = [torch.empty_like(mine) for _ in range(p.world)]
        dist.all_gather(out, mine, group=self.group)
        return p.assemble([o.cpu().numpy() for o in out], p.nrows)

    def gather_cols_host(self, block):
        p = self.plan
        if p.world == 1:
            return block
        import torch
        import torch.distributed as dist
        pad = np.zeros((p.col_chunk,) + block.shape[1:])
        pad[:block.shape[0]] = block
        mine = torch.from_numpy(pad)
        if dist.get_backend(self.group) == "nccl":
            mine = mine.cuda()
        out = [torch.empty_like(mine) for _ in range(p.world)]
        dist.all_gather(out, mine, group=self.group)
        return p.assemble([o.cpu().numpy() for o in out], p.ncols)
