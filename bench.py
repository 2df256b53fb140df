#!/usr/bin/env python3
"""Headline benchmark: Gibbs sweeps/sec of the full W+V update (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

One "step" = one W half-sweep + one V half-sweep (model._resample_W then
model._resample_V through the C ABI) on synthetic Gaussian data that is already
resident in HBM, with the normals drawn on the device (rng="device": nothing but
scalars crosses PCIe inside the timed region).

N = 1   workload C3 of BASELINE.json: (512,256,64,4), nembeds=5, tf_order=2, complete data.
N > 1   weak scaling over rows: global tensor (512*N,256,64,4); every rank streams one
        (512,256,64,4)-sized slab per half-sweep (its rows in the W step, its 256/N
        columns of all rows in the V step); W and V blocks are all-gathered over
        RCCL after each half-sweep.  value = N * (global sweeps/s) = C3-sized slab
        updates per second over the whole job.
Use --strong to shard a fixed tensor instead, --config c5 for (4096,1024,64,4) K=8.

Besides the contract fields the JSON line carries `roofline` (streaming accumulation
kernel: algorithmic bytes / HIP-event time vs the 8 TB/s HBM peak) and, at N=1,
`cpu_baseline` (the numpy oracle's reference-faithful W+V update on the host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    "c2": dict(N=64, M=32, T=16, R=2, K=3),
    "c3": dict(N=512, M=256, T=64, R=4, K=5),
    "c5": dict(N=4096, M=1024, T=64, R=4, K=8),
    "c3k8": dict(N=512, M=256, T=64, R=4, K=8),      # tuning aid: C3 cells with C5's embedding size
}
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def synth_rows(seed, rows, M, T, R, K, Vt, noise=0.5):
    """Rows `rows` of the SURVEY 8(d) synthetic tensor; row i depends only on (seed, i),
    so any rank can generate any subset consistently."""
    out = np.empty((len(rows), M, T, R))
    W = np.empty((len(rows), K))
    for n, i in enumerate(rows):
        rs = np.random.RandomState((seed * 1000003 + int(i)) % (2 ** 31))
        w = rs.normal(0, 1, size=K)
        if i < K:
            w[i + 1:] = 0
        W[n] = w
        out[n] = (Vt @ w)[..., None] + rs.normal(0, noise, size=(M, T, R))
    return out, W


def synth_V(seed, M, T, K):
    rs = np.random.RandomState(seed)
    return 0.1 * np.cumsum(rs.normal(0, 1, size=(M, T, K)), axis=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--strong", action="store_true", help="fixed global tensor instead of weak scaling")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--rpb", type=int, nargs=2, default=[0, 0], help="rows per workgroup (W, V) tuning override")
    ap.add_argument("--variant", default="complete", choices=["complete", "heldout", "missing5", "binomial", "negbinom"],
                    help="complete: headline; heldout: Y[:3,:3]=NaN; missing5: 5%% curves + 5%% single replicates NaN; "
                         "binomial: 4 trials per cell, device Polya-Gamma draw included in the step (config C4); "
                         "negbinom: NB(4, p) counts, step = 30 MH steps on the rate R + PG draw + W + V (SURVEY 8(f) rank 2)")
    ap.add_argument("--burn", type=int, default=3, help="full Gibbs sweeps before timing (leave the initial state)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("BTF_FORCE_DEVICE") is not None:      # rehearsal aid: several ranks on one GPU
        local_rank = int(os.environ["BTF_FORCE_DEVICE"])
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    exercise = os.environ.get("BTF_EXERCISE_EXCHANGE", "0") == "1"
    if world > 1 or exercise:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from functionalmf_amd.factor import (GaussianBayesianTensorFiltering, BinomialBayesianTensorFiltering,
                                         NegativeBinomialBayesianTensorFiltering)
    from functionalmf_amd.parallel import ShardPlan

    cfg = dict(CONFIGS[args.config])
    weak = world > 1 and not args.strong
    if weak:
        cfg["N"] *= world
    N, M, T, R, K = cfg["N"], cfg["M"], cfg["T"], cfg["R"], cfg["K"]

    # ---- synthetic data: only this rank's two slabs are ever materialised -------------
    Vt = synth_V(1, M, T, K)
    plan = ShardPlan(N, M, rank, world)
    t0 = time.time()
    if world == 1:
        Y, _ = synth_rows(1, range(N), M, T, R, K, Vt)
        slabs = None
    else:
        rows, _ = synth_rows(1, range(plan.row0, plan.row0 + plan.nl), M, T, R, K, Vt)
        cols = np.empty((N, plan.ml, T, R))
        for i0 in range(0, N, 256):
            blk, _ = synth_rows(1, range(i0, min(i0 + 256, N)), M, T, R, K, Vt)
            cols[i0:i0 + blk.shape[0]] = blk[:, plan.col0:plan.col0 + plan.ml]
        slabs = (rows, cols)
        Y = None
    t_data = time.time() - t0

    if args.variant != "complete":
        if world > 1:
            sys.exit("--variant other than complete is single-GPU only in this round")
        rs = np.random.RandomState(7)
        if args.variant == "heldout":
            Y[:3, :3] = np.nan
        elif args.variant == "missing5":
            Y[rs.rand(N, M) < 0.05] = np.nan
            Y[rs.rand(N, M, T, R) < 0.05] = np.nan
        elif args.variant == "negbinom":
            Mu = np.einsum("nk,mtk->nmt", synth_rows(1, range(N), M, T, 1, K, Vt, noise=0.0)[1], Vt)
            P = 1 / (1 + np.exp(-Mu))
            Y = rs.negative_binomial(4.0, 1 - np.repeat(P[..., None], R, axis=-1)).astype(float)
        else:
            Mu = np.einsum("nk,mtk->nmt", synth_rows(1, range(N), M, T, 1, K, Vt, noise=0.0)[1], Vt)
            Ntr = np.full((N, M, T), 4.0)
            Y = (rs.binomial(4, 1 / (1 + np.exp(-Mu))).astype(float), Ntr)
    np.random.seed(1)
    stream = torch.cuda.current_stream().cuda_stream or None     # null stream -> the ctx's own / a dedicated torch stream
    common = dict(nembeds=K, tf_order=2, sigma2_init=0.5, lam2_init=0.1, rng="device",
                  compat="reference" if world == 1 else "exact", device=local_rank, stream=stream,
                  shard=(rank, world) if world > 1 else None, device_seed=1)
    if args.variant == "binomial":
        model = BinomialBayesianTensorFiltering(N, M, T, **common)
    elif args.variant == "negbinom":
        model = NegativeBinomialBayesianTensorFiltering(N, M, T, **common)
    else:
        model = GaussianBayesianTensorFiltering(N, M, T, nu2_init=1.0, **common)
    if args.rpb != [0, 0]:
        model._ctx.call("btf_set_tuning", args.rpb[0], args.rpb[1])
    if world == 1:
        data = Y
    else:
        data = _SlabData(slabs, (N, M, T, R))
        model._upload = lambda d, _m=model: _upload_slabs(_m, d)
    # leave the prior draw: a few full sweeps (nu2, sigma2, Tau2, lam2, W, V)
    for _ in range(args.burn):
        model.resample(data)
    model.sync()

    if args.variant == "negbinom":
        def step():                       # rate update (30 MH steps) + PG draw + W + V
            model._resample_R(data)
            model._resample_nu2(data)
            model._resample_W(data)
            model._resample_V(data)
    elif args.variant == "binomial":
        def step():                       # C4: full Binomial sweep of the device part: PG draw + W + V
            model._resample_nu2(data)
            model._resample_W(data)
            model._resample_V(data)
    else:
        def step():
            model._resample_W(data)
            model._resample_V(data)

    def fence():
        if world > 1 or exercise:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    model.sync()                      # raises if any factorisation failed inside the timed region
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- per-kernel HIP-event timing of the same steps (separate pass: events perturb) ----
    model._ctx.call("btf_set_profiling", 1)
    model._ctx.kernel_times()
    nprof = min(args.steps, 200)
    for _ in range(nprof):
        step()
    fence()
    kt = model._ctx.kernel_times()
    model._ctx.call("btf_set_profiling", 0)

    # full Gibbs sweep (nu2, sigma2, Tau2, lam2, W, V) - device Tau2 chain, host scalar draws
    nfull = max(10, min(args.steps, 100))
    for _ in range(3):
        model.resample(data)
    fence()
    t0 = time.perf_counter()
    for _ in range(nfull):
        model.resample(data)
    fence()
    full_per_s = nfull / (time.perf_counter() - t0)

    sweeps_per_s = args.steps / dt
    units = world if weak else 1
    value = sweeps_per_s * units

    # algorithmic bytes of one accumulation launch: the local slab of the linear
    # statistic once (8 B/cell, complete data) + the small operands / partials it touches
    cells_local = (N // world if world > 1 else N) * M * T
    acc_ms = kt["w_accum"][0] + kt["v_accum"][0]
    acc_n = kt["w_accum"][1] + kt["v_accum"][1]
    acc_us = 1e3 * acc_ms / max(acc_n, 1)
    # complete data: the linear statistic only; missing data: + a byte of replicate count; Binomial: + f64 weights
    alg_bytes = {"complete": 8.0, "heldout": 9.0, "missing5": 9.0}.get(args.variant, 16.0) * cells_local
    achieved = alg_bytes / (acc_us * 1e-6) / 1e9 if acc_us > 0 else 0.0
    kernels_us = {k: round(1e3 * v[0] / max(v[1], 1), 2) for k, v in kt.items() if v[1] > 0}
    traffic = pmc_traffic("accum_kernel") if (world == 1 and args.config == "c3") else None

    out = {
        "metric": "Gibbs sweeps/sec (full W+V update) at (512,256,64) K=5; % HBM roofline",
        "value": round(value, 2),
        "unit": "sweeps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 4),
        "higher_is_better": True,
        "scaling": "weak" if (weak or world == 1) else "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "%s_btf %s (%d,%d,%d,%d) nembeds=%d tf_order=2 %s data, W+V update, rng=device%s"
                               % ({"binomial": "binomial", "negbinom": "negbinom"}.get(args.variant, "gaussian"), args.config, N, M, T, R, K, args.variant,
                                  "; %d-way row/column shards, RCCL all-gather of W and V" % world if world > 1 else ""),
                   "global_sweeps_per_s": round(sweeps_per_s, 2), "units_per_sweep": units,
                   "full_resample_sweeps_per_s": round(full_per_s, 2),
                   "parallelism": "rows(W)/cols(V) x%d" % world},
        "roofline": {"bound": "hbm", "kernel": "accum_kernel (w_accum + v_accum launches)",
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_us": round(acc_us, 2)},
        "kernels_us": kernels_us,
    }

    if world == 1 and not args.no_cpu and args.variant not in ("binomial", "negbinom"):
        out["cpu_baseline"] = cpu_baseline(Y, model, cfg)

    if rank == 0:
        print(json.dumps(out))
    if world > 1 or exercise:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of the streaming kernel from the committed rocprofv3 PMC passes
    (profiles/r*_pmc_summary.json: 2*FETCH_SIZE + WRITE_SIZE, the gfx950 correction of
    MI355X_MICROARCH.md applied); None if no profile of this workload is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        vals = [v["hbm_bytes_per_launch_corrected"] for k, v in d.items() if kernel_substr in k]
        return round(max(vals), 1) if vals else None
    except Exception:
        return None


class _SlabData:
    """Stand-in for the full observation tensor in sharded benchmark runs (it is never
    materialised): carries just this rank's row and column slabs."""

    def __init__(self, slabs, shape):
        self.rows, self.cols = slabs
        self.shape = shape
        self.ndim = len(shape)


def _upload_slabs(model, d):
    from functionalmf_amd import _native
    model._ctx.call("btf_set_data_gaussian", _native.dptr(np.ascontiguousarray(d.rows)),
                    _native.dptr(np.ascontiguousarray(d.cols)), int(d.shape[3]))


def cpu_baseline(Y, model, cfg):
    """The oracle's reference-faithful W+V update (per-row / per-column loops, the
    sufficient statistics re-reduced from the 4-D tensor on every half-sweep as
    factor.py:329-330/:374-375 do, dense LAPACK in place of CHOLMOD) on the host cores,
    from the GPU chain's current state; bounded to roughly 10-30 s."""
    from oracle import btf_oracle as orc
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    st = dict(W=model.W.copy(), V=model.V.copy(), Tau2=np.array(model.Tau2, dtype=float).copy(),
              lam2=float(model.lam2), sigma2=float(model.sigma2), nu2=float(np.asarray(model.nu2).reshape(-1)[0]))
    Delta = orc.trend_penalty(cfg["T"], 2)
    np.random.seed(123)
    n, t0 = 0, time.perf_counter()
    while True:
        orc.w_step(st, Y)
        orc.v_step(st, Y, Delta)
        n += 1
        el = time.perf_counter() - t0
        if el > 12.0 or n >= 50:
            break
    out = {"value": round(n / el, 4), "unit": "sweeps/s", "cores": int(cores), "kind": "port",
           "sample": "%d full W+V updates of the same (%d,%d,%d,%d) K=%d tensor by oracle/btf_oracle.py "
                     "(numpy/LAPACK, BLAS threads=%d)" % (n, cfg["N"], cfg["M"], cfg["T"], cfg["R"], cfg["K"], cores)}
    # second CPU number: statistics hoisted, BLAS + banded LAPACK ("strong CPU")
    try:
        Rr, ybar = orc.hoisted_stats(Y)
        n2, t0 = 0, time.perf_counter()
        while True:
            orc.w_step_strong(st, Rr, ybar)
            orc.v_step_strong(st, Rr, ybar, Delta)
            n2 += 1
            el2 = time.perf_counter() - t0
            if el2 > 8.0 or n2 >= 200:
                break
        out["strong_cpu_value"] = round(n2 / el2, 3)
        out["strong_cpu_sample"] = "%d W+V updates, hoisted statistics + BLAS + scipy banded Cholesky" % n2
    except Exception as e:        # pragma: no cover
        out["strong_cpu_value"] = None
        out["strong_cpu_sample"] = "failed: %r" % (e,)
    return out


if __name__ == "__main__":
    main()
