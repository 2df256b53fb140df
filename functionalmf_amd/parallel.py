"""Row / column sharding of the two half-sweeps over the GPUs of one node.

Given V the rows of W are conditionally independent (factor.py:333) and given W
the columns of V are (factor.py:378), so rank p updates a contiguous block of
rows in the W half-sweep and a contiguous block of columns in the V half-sweep,
from two slabs of the sufficient statistics (its rows x everything, everything
x its columns).  W and V are replicated; the only exchange per half-sweep is
one all-gather of the freshly drawn block (RCCL over xGMI through
torch.distributed's "nccl" backend on the device buffers of the context, or any
other backend on host copies - used by the CPU tests with gloo).
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
    """The per-half-sweep exchange.  world == 1: no-ops."""

    def __init__(self, plan, ctx=None, group=None, overlap=True, rehearse=False):
        self.plan, self.ctx, self.group = plan, ctx, group
        self.overlap = overlap          # all-gathers on a communication stream of their own (see _gather_stream)
        # rehearse: ONE process plays rank `plan.rank` of `plan.world` (bench.py --as-rank): its kernels run on that
        # rank's real slabs, and every collective of the sharded step is issued in a one-rank group on scratch buffers
        # of the full message size - the blocks of the other ranks in W / V are never refreshed, so the chain is not a
        # sampler of anything; only the clock is read
        self.rehearse = rehearse
        self._Wt = self._Vt = None
        self._tstream = self._cstream = None
        self.timing, self._events = False, {"all_gather_W": [], "all_gather_V": [], "all_reduce_sse": []}
        # BTF_EXERCISE_EXCHANGE=1: issue the collectives even in a 1-rank group (lets a 1-GPU box
        # run the exact RCCL call sequence of the sharded path)
        import os
        self.active = plan.world > 1 or rehearse or os.environ.get("BTF_EXERCISE_EXCHANGE", "0") == "1"
        if self.active:
            import torch.distributed as dist
            if not dist.is_initialized():
                raise RuntimeError("shard=(rank, world) needs an initialised torch.distributed process group")
            if rehearse:
                if dist.get_world_size(group) != 1:
                    raise RuntimeError("a rehearsal runs in a one-rank process group")
            elif dist.get_world_size(group) != plan.world or dist.get_rank(group) != plan.rank:
                raise RuntimeError("shard does not match the process group")

    # -- device path (RCCL) ----------------------------------------------------------
    def _stream(self):
        """The ctx's HIP stream as a torch stream: the collectives are issued under it, so that RCCL orders
        them after the draw kernel and the next half-sweep after them - whichever stream is torch's current one."""
        import torch
        if self._tstream is None:
            self._tstream = torch.cuda.ExternalStream(self.ctx.stream_handle, device=torch.device("cuda", self.ctx.device))
        return self._tstream

    def _comm(self):
        """The communication stream of the overlapped exchange: ordered behind the kernel that drew this rank's block
        (btf_comm_fork) - NOT behind the own-block chunks of the next accumulation the ctx queues after it - and
        joined back into the ctx's stream when the gather has been issued (btf_comm_join)."""
        import torch
        if self._cstream is None:
            self._cstream = torch.cuda.Stream(device=torch.device("cuda", self.ctx.device))
        return self._cstream

    def _gather(self, name, fn):
        """One all-gather of the freshly drawn block: overlapped (own stream, fork / join against the ctx's) or in line."""
        if not self.overlap:
            return self._timed(name, fn)
        import ctypes as C
        comm = self._comm()
        self.ctx.call("btf_comm_fork", C.c_void_p(comm.cuda_stream))
        self._timed(name, fn, stream=comm)
        self.ctx.call("btf_comm_join", C.c_void_p(comm.cuda_stream))

    def _timed(self, name, fn, stream=None):
        import torch
        with torch.cuda.stream(stream if stream is not None else self._stream()):
            if self.timing:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                self._events[name].append((e0, e1))
            else:
                fn()

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

    def _views(self):
        import torch
        p, (N, M, T, K, _) = self.plan, self.ctx.dims
        if self._Wt is None:
            lib = self.ctx.lib
            dev = torch.device("cuda", self.ctx.device)
            # the context over-allocates W / V by 64 rows / columns, so world*chunk fits
            self._Wt = torch.as_tensor(_DevView(lib.btf_dev_W(self.ctx.h), (p.world * p.row_chunk * K,)), device=dev)
            self._Vt = torch.as_tensor(_DevView(lib.btf_dev_V(self.ctx.h), (p.world * p.col_chunk * T * K,)), device=dev)
        return self._Wt, self._Vt

    def _scratch(self, which):
        """Rehearsal buffers: a one-rank all-gather moves out = in, so both have the size of the whole gathered factor."""
        import torch
        if getattr(self, "_scr", None) is None:
            p, (N, M, T, K, _) = self.plan, self.ctx.dims
            dev = torch.device("cuda", self.ctx.device)
            nW, nV = p.world * p.row_chunk * K, p.world * p.col_chunk * T * K
            self._scr = {"W": torch.zeros(nW, dtype=torch.float64, device=dev), "W_in": torch.zeros(nW, dtype=torch.float64, device=dev),
                         "V": torch.zeros(nV, dtype=torch.float64, device=dev), "V_in": torch.zeros(nV, dtype=torch.float64, device=dev)}
        return self._scr[which]

    def _staged(self):
        """True when the process group cannot move device memory (e.g. gloo): the exchange is then
        staged through the host (used to rehearse several ranks on one GPU; RCCL refuses that)."""
        import torch.distributed as dist
        return dist.get_backend(self.group) != "nccl"

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
        import torch.distributed as dist
        if self.rehearse:
            return self._gather("all_gather_W", lambda: dist.all_gather_into_tensor(self._scratch("W"), self._scratch("W_in"), group=self.group))
        Wt, _ = self._views()
        K = self.ctx.dims[3]
        n = self.plan.row_chunk * K
        self._gather("all_gather_W", lambda: dist.all_gather_into_tensor(Wt, Wt[self.plan.rank * n:(self.plan.rank + 1) * n], group=self.group))

    def after_V(self):
        if not self.active:
            return
        if self._staged():
            N, M, T, K, _ = self.ctx.dims
            return self._staged_gather("btf_get_V", "btf_set_gathered_V", (M, T, K), self.plan.col0, self.plan.ml, self.gather_cols_host)
        import torch.distributed as dist
        if self.rehearse:
            return self._gather("all_gather_V", lambda: dist.all_gather_into_tensor(self._scratch("V"), self._scratch("V_in"), group=self.group))
        _, Vt = self._views()
        _, _, T, K, _ = self.ctx.dims
        n = self.plan.col_chunk * T * K
        self._gather("all_gather_V", lambda: dist.all_gather_into_tensor(Vt, Vt[self.plan.rank * n:(self.plan.rank + 1) * n], group=self.group))

    def all_reduce_sse(self):
        """Sum the rank-local residual sum of squares (device scalar slot 4, btf_dev_hyp) over the ranks, in place,
        on the ctx's stream: the one exchange of a sharded nu2 draw (include/btf.h, btf_draw_scalars which | 8 / | 16)."""
        if not self.active:
            return
        import torch.distributed as dist
        if self._staged():
            out = np.zeros(6)
            self.ctx.call("btf_get_scalars", out.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_double)))
            (tot,) = self.sum_scalars(float(out[4]))
            self.ctx.call("btf_set_scalar_slot", 4, float(tot))
            return
        import torch
        if getattr(self, "_sse_t", None) is None:
            ptr = self.ctx.lib.btf_dev_hyp(self.ctx.h)
            self._sse_t = torch.as_tensor(_DevView(ptr + 4 * 8, (1,)), device=torch.device("cuda", self.ctx.device))
        self._timed("all_reduce_sse", lambda: dist.all_reduce(self._sse_t, group=self.group))

    def sum_scalars(self, *vals):
        if not self.active:
            return vals
        import torch
        import torch.distributed as dist
        dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
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
