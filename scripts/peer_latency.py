"""Latency of the peer-window collectives (csrc/btf_comm.h) between processes that share ONE GPU - the only N > 1 form a
one-GPU box can run.  `python -m torch.distributed.run --nproc-per-node P scripts/peer_latency.py [N M T K]`: every rank
holds a context of the given shape (default: C5, 4096 x 1024 x 64, nembeds 8), blocks of rank r of P, and times
btf_allgather_W + btf_allgather_V pairs with HIP events on the context's stream.  Not an xGMI number: the stores land
in the same HBM; what it shows is the fixed cost (one launch, two flag round trips through fine-grained memory)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from functionalmf_amd import _native
    from functionalmf_amd.parallel import Exchange, ShardPlan
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    N, M, T, K = [int(x) for x in sys.argv[1:5]] if len(sys.argv) >= 5 else (4096, 1024, 64, 8)
    ctx = _native.Context(N, M, T, K, 2)
    plan = ShardPlan(N, M, rank, world)
    ctx.call("btf_set_shard", *plan.mine())
    rs = np.random.RandomState(rank)
    ctx.call("btf_set_W", _native.dptr(rs.normal(size=(N, K))))
    ctx.call("btf_set_V", _native.dptr(rs.normal(size=(M, T, K))))
    ex = Exchange(plan, ctx, overlap=False, transport="peer")
    stream = torch.cuda.ExternalStream(ctx.stream_handle, device=torch.device("cuda", 0))
    out = {}
    for name, calls in (("W", ("btf_allgather_W",)), ("V", ("btf_allgather_V",)), ("W+V", ("btf_allgather_W", "btf_allgather_V"))):
        for _ in range(20):
            for c in calls:
                ctx.call(c)
        ctx.call("btf_sync")
        dist.barrier()
        reps = 300
        with torch.cuda.stream(stream):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                for c in calls:
                    ctx.call(c)
            e1.record()
        ctx.call("btf_sync")
        out[name] = 1e3 * e0.elapsed_time(e1) / reps
    t = torch.tensor([out["W"], out["V"], out["W+V"]], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        wb, vb = plan.nl * K * 8, plan.ml * T * K * 8
        print("PEER_LATENCY world %d shape %s block bytes W %d V %d: us per collective W %.2f V %.2f, W+V pair %.2f" %
              (world, (N, M, T, K), wb, vb, t[0], t[1], t[2]), flush=True)
    dist.barrier()
    ctx.call("btf_comm_destroy")
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
