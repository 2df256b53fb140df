"""Worker of test_sharded_half_sweeps_equal_unsharded_gloo_world2 (gloo, CPU)."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, state_from  # noqa: E402
from oracle import btf_oracle as orc  # noqa: E402
from functionalmf_amd.parallel import ShardPlan, Exchange  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    g = load_golden("g1_c1_heldout.npz")
    Y = g["Y"]
    N, M, T, R, K, tf = [int(x) for x in g["dims"]]
    Delta = orc.trend_penalty(T, tf)
    plan = ShardPlan(N, M, rank, world)
    ex = Exchange(plan)
    rows, cols = plan.slabs(Y)
    assert rows.shape == (plan.nl, M, T, R) and cols.shape == (N, plan.ml, T, R)

    # ---- W half-sweep: my rows only, from my row slab, then all-gather
    st = state_from(g, "s0_")
    mine = dict(st, W=st["W"][plan.row0:plan.row0 + plan.nl].copy())
    orc.w_step(mine, rows, z=g["z_W"], row0=plan.row0)
    W = ex.gather_rows_host(mine["W"])
    ref = state_from(g, "s0_")
    orc.w_step(ref, Y, z=g["z_W"])
    assert W.shape == ref["W"].shape and np.array_equal(W, ref["W"]), np.abs(W - ref["W"]).max()

    # ---- the overlapped exchange (BTF_OPT_SPLIT_ACCUM): the W half-sweep's sums over (j, t) split into the chunk of this
    # rank's OWN columns of V - known before the all-gather of V - and the rest; the two parts add up to the whole
    cnt_r, yb_r = orc.replicate_stats(rows)                        # (nl, M, T)
    c = np.where(cnt_r > 0, cnt_r, 0.0)
    s1 = np.where(cnt_r > 0, cnt_r * np.nan_to_num(yb_r), 0.0)
    own = slice(plan.col0, plan.col0 + plan.ml)
    rest = np.r_[0:plan.col0, plan.col0 + plan.ml:M]
    Vst = state_from(g, "s0_")["V"]
    m_all = np.einsum("nmt,mtk->nk", s1, Vst)
    q_all = np.einsum("nmt,mtk,mtl->nkl", c, Vst, Vst)
    m_two = np.einsum("nmt,mtk->nk", s1[:, own], Vst[own]) + np.einsum("nmt,mtk->nk", s1[:, rest], Vst[rest])
    q_two = np.einsum("nmt,mtk,mtl->nkl", c[:, own], Vst[own], Vst[own]) + np.einsum("nmt,mtk,mtl->nkl", c[:, rest], Vst[rest], Vst[rest])
    assert np.abs(m_two - m_all).max() <= 1e-12 * np.abs(m_all).max() and np.abs(q_two - q_all).max() <= 1e-12 * np.abs(q_all).max()

    # ---- V half-sweep: my columns only (exact mode: no stale source outside the shard)
    st["W"] = W
    full = dict(st, V=st["V"].copy())
    orc.v_step(full, Y, Delta, z=g["z_V"], compat="exact", cols=range(plan.col0, plan.col0 + plan.ml))
    V = ex.gather_cols_host(full["V"][plan.col0:plan.col0 + plan.ml])
    ref["V"] = st["V"].copy()
    orc.v_step(ref, Y, Delta, z=g["z_V"], compat="exact")
    assert np.array_equal(V, ref["V"])

    # ---- nu2 statistics: between-cell part over my columns + within-cell part / count over my rows
    st["V"] = V
    cnt_c, yb_c = orc.replicate_stats(cols)
    Mu = np.einsum("nk,mtk->nmt", st["W"], V)
    mu_c = Mu[:, plan.col0:plan.col0 + plan.ml]
    between = np.nansum(cnt_c * (np.where(cnt_c > 0, yb_c, 0.0) - mu_c) ** 2)
    cnt_r, yb_r = orc.replicate_stats(rows)
    within = np.nansum((rows - yb_r[..., None]) ** 2)
    sse, nobs = ex.sum_scalars(between + within, float(cnt_r.sum()))
    sse_ref, n_ref = orc.sse_and_count(st, Y)
    assert abs(sse - sse_ref) / sse_ref < 1e-12 and nobs == n_ref
    print("DIST_OK rank", rank, flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
