#!/usr/bin/env python3
"""Headline benchmark: Gibbs sweeps/sec of the full W+V update (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

One "step" = one W half-sweep + one V half-sweep (model._resample_W then model._resample_V through
the C ABI) on synthetic data that is already resident in HBM, with the normals drawn on the device
(rng="device": nothing but scalars crosses PCIe inside the timed region).  The prior precision
Delta' diag(1/(lam2 Tau2_j)) Delta of every column is rebuilt in every step, as factor.py:404-405 does.

N = 1   workload C3 of BASELINE.json: Gaussian (512,256,64,4), nembeds=5, tf_order=2, complete data
        (--variant / --config select the other single-GPU workloads; C4 = --variant binomial).
N > 1   workload C5 of BASELINE.json, STRONG scaling: the fixed tensor (4096,1024,64,4) nembeds=8 is
        row-sharded (W half-sweep) and column-sharded (V half-sweep) over the N ranks; the freshly drawn
        block of W resp. V is all-gathered over RCCL after each half-sweep.  value = sweeps/s of that one
        global tensor.  (--weak: every rank streams one C3-sized slab, value = N x global sweeps/s.)
        `python bench.py --gpus N` launches its own ranks (one child `python -m torch.distributed.run`);
        under torchrun (WORLD_SIZE set) it is a rank.

Besides the contract fields the JSON line carries `roofline` (the streaming accumulation kernel:
algorithmic bytes / HIP-event time of exactly that dispatch against the 8 TB/s HBM peak, plus the
whole-step fraction and a measured device-copy ceiling) and, at N=1, `cpu_baseline` (the numpy
oracle's reference-faithful W+V update on the host cores, all BLAS threads and one thread).
"""
import argparse
import json
import os
import subprocess
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
    "c3k10": dict(N=512, M=256, T=64, R=4, K=10),    # C3 cells with the reference's larger embedding (flutrends/benchmark.py:33)
    "flu": dict(N=50, M=1, T=370, R=1, K=10),        # flutrends/benchmark.py:31-34: 50 states x 1 x 370 weeks, nembeds 10
}
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
METRIC = "Gibbs sweeps/sec (full W+V update) at (512,256,64) K=5; % HBM roofline"      # BASELINE.json's metric: config c3


def metric_for(config, world):
    """BASELINE.json's metric string for its own configuration (C3); any other workload - the C5 tensor of the N > 1 runs
    included - names its shape instead of passing for it."""
    if config == "c3":
        return METRIC
    c = CONFIGS[config]
    return "Gibbs sweeps/sec (full W+V update) at (%d,%d,%d) K=%d%s; %% HBM roofline" % (
        c["N"], c["M"], c["T"], c["K"], ", row/column-sharded over %d GPUs" % world if world > 1 else "")


SYNTH_CB = 128     # columns per noise block of the synthetic tensor


def synth_rows(seed, rows, M, T, R, K, Vt, noise=0.5, cols=None):
    """Rows `rows` (and, with cols = (c0, c1), only columns c0..c1-1) of the SURVEY 8(d) synthetic tensor.  The factor row
    w_i depends only on (seed, i) and the noise of row i in the column block b (SYNTH_CB columns) only on (seed, i, b), so
    any rank generates exactly its two slabs - its rows x everything, everything x its columns - consistently with
    every other rank (a column slab used to cost a pass over the whole tensor: 85 s per rank at C5)."""
    c0, c1 = (0, M) if cols is None else cols
    out = np.empty((len(rows), c1 - c0, T, R))
    W = np.empty((len(rows), K))
    Vs = Vt[c0:c1]
    for n, i in enumerate(rows):
        rs = np.random.RandomState((seed * 1000003 + int(i)) % (2 ** 31))
        w = rs.normal(0, 1, size=K)
        if i < K:
            w[i + 1:] = 0
        W[n] = w
        out[n] = (Vs @ w)[..., None]
        if noise > 0:
            for b in range(c0 // SYNTH_CB, (c1 + SYNTH_CB - 1) // SYNTH_CB):
                b0, b1 = b * SYNTH_CB, min((b + 1) * SYNTH_CB, M)
                # (PCG64 + ziggurat normals: four times the rate of the legacy generator; the data need no legacy stream)
                rb = np.random.Generator(np.random.PCG64([seed, int(i), b]))
                blk = noise * rb.standard_normal(size=(b1 - b0, T, R))
                lo, hi = max(b0, c0), min(b1, c1)
                out[n, lo - c0:hi - c0] += blk[lo - b0:hi - b0]
    return out, W


def synth_legacy(seed, N, M, T, R, K, noise=0.5):
    """SURVEY 8(d)'s synthetic tensor exactly as specified, for the unsharded runs: ONE legacy stream
    (np.random.seed(seed), MT19937 + polar normals, as examples/gaussian_tensor_filtering.py:14-19 draws its data):
    W_true ~ N(0,1) (N,K) with the upper triangle zeroed, V_true = 0.1 cumsum(N(0,1) (M,T,K), axis=1),
    Y = Mu[..., None] + N(0, noise^2) (N,M,T,R).  Returns Y, W_true, V_true."""
    rs = np.random.RandomState(seed)
    Wt = rs.normal(0, 1, size=(N, K))
    Wt[np.triu_indices(K, 1)] = 0
    Vt = 0.1 * np.cumsum(rs.normal(0, 1, size=(M, T, K)), axis=1)
    Y = np.einsum("nk,mtk->nmt", Wt, Vt)[..., None] + rs.normal(0, noise, size=(N, M, T, R))
    return Y, Wt, Vt


LEGACY_MAX_VALUES = 1 << 28     # the one-stream generator up to 2^28 values (C3: 2^25); beyond it - and for every rank slab - synth_rows


def synth_V(seed, M, T, K):
    rs = np.random.RandomState(seed)
    return 0.1 * np.cumsum(rs.normal(0, 1, size=(M, T, K)), axis=1)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 2000 for the C2 / C3-sized workloads - 90 ms of GPU time -, 200 for C5)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS), help="default: c3 at --gpus 1, c5 beyond")
    ap.add_argument("--weak", action="store_true", help="N>1: weak scaling over rows (one C3-sized slab per rank) instead of the fixed C5 tensor")
    ap.add_argument("--strong", action="store_true", help="(default for N>1; kept for compatibility)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-c4", action="store_true", help="skip the short BASELINE-config-4 (Binomial) leg of the default run")
    ap.add_argument("--rpb", type=int, nargs=2, default=[0, 0], help="rows per workgroup (W, V) tuning override")
    ap.add_argument("--variant", default="complete", choices=["complete", "heldout", "curves5", "missing5", "binomial", "negbinom"],
                    help="complete: headline; heldout: Y[:3,:3]=NaN; curves5: 5%% of the (i,j) curves NaN; "
                         "missing5: 5%% curves + 5%% single replicates NaN; "
                         "binomial: 4 trials per cell, device Polya-Gamma draw included in the step (config C4); "
                         "negbinom: NB(4, p) counts, step = 30 MH steps on the rate R + PG draw + W + V (SURVEY 8(f) rank 2)")
    ap.add_argument("--sampler", default="auto", help="V sampler: auto (spectral on complete data, banded otherwise), banded, spectral, chain")
    ap.add_argument("--pg-exact", action="store_true", help="binomial / negbinom: Devroye's exact Polya-Gamma sampler for every count "
                    "(the default already draws integer counts up to 32 exactly)")
    ap.add_argument("--pg-series", action="store_true", help="binomial / negbinom: the approximate sum-of-gammas series for every count")
    ap.add_argument("--burn", type=int, default=10, help="full Gibbs sweeps before timing (leave the initial state)")
    ap.add_argument("--lean", action="store_true", help="profiling runs: nothing but W+V steps after the burn-in - no full-sweep leg, no banded-"
                    "sampler leg, no CPU baseline - so that a rocprofv3 --stats average is over the dispatches the HIP events time")
    ap.add_argument("--as-rank", default=None, metavar="R/P", help="ONE GPU plays rank R of a P-GPU run of --config (default c5): its "
                    "kernels on that rank's real slabs, every collective of the sharded step in a one-rank RCCL group at the full "
                    "message size; adds a `projected` block (never `value`: that stays this GPU's own rate)")
    ap.add_argument("--overlap", action="store_true", help="sharded runs: all-gathers on a communication stream, own-block chunks of the "
                    "next accumulation ahead of them (BTF_OPT_SPLIT_ACCUM); default: collectives in line on the ctx's stream")
    ap.add_argument("--python-loop", action="store_true", help="drive the timed W+V steps from Python, call by call (default on one GPU with "
                    "Gaussian data: one btf_wv_steps call queues them all)")
    ap.add_argument("--master-port", type=int, default=29533)
    return ap.parse_args(argv)


def self_launch(args):
    """python bench.py --gpus N (N>1) outside torchrun: start the ranks as a CHILD process (never exec: nothing here
    has touched the GPU yet, and the child is an ordinary subprocess), relay its one JSON line and its exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(args.master_port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in proc.stdout.splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1])
    return proc.returncode if proc.returncode != 0 or lines else 1


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if world == 0 and args.gpus > 1:
        sys.exit(self_launch(args))
    world = max(world, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("BTF_FORCE_DEVICE") is not None:      # rehearsal aid: several ranks on one GPU
        local_rank = int(os.environ["BTF_FORCE_DEVICE"])
    args.gpus = world
    dry = os.environ.get("BTF_BENCH_DRY", "0") == "1"       # CPU rehearsal of the launch / timing / reporting plumbing
    as_rank = None
    if args.as_rank:
        if world != 1:
            sys.exit("--as-rank is a one-GPU rehearsal")
        as_rank = tuple(int(x) for x in args.as_rank.split("/"))
        if args.config is None:
            args.config = "c5"
    if args.config is None:
        args.config = "c3" if (world == 1 or args.weak) else "c5"
    if args.steps is None:
        args.steps = 200 if args.config == "c5" else (500 if args.variant in ("binomial", "negbinom") else 2000)

    import torch
    import torch.distributed as dist
    # (a rehearsed rank needs no process group: its communicator is the context's own one-rank one, btf_comm_rehearse)
    exercise = os.environ.get("BTF_EXERCISE_EXCHANGE", "0") == "1"
    backend = None
    if not dry:
        torch.cuda.set_device(local_rank)
    if world > 1 or exercise:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(args.master_port))
        backend = "gloo" if dry or os.environ.get("BTF_BACKEND") == "gloo" else "nccl"
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    if dry:
        return dry_run(args, world, rank, dist, backend)

    from functionalmf_amd import _native
    from functionalmf_amd.factor import (GaussianBayesianTensorFiltering, BinomialBayesianTensorFiltering,
                                         NegativeBinomialBayesianTensorFiltering)
    from functionalmf_amd.parallel import ShardPlan

    cfg = dict(CONFIGS[args.config])
    weak = world > 1 and args.weak
    if weak:
        cfg["N"] *= world
    N, M, T, R, K = cfg["N"], cfg["M"], cfg["T"], cfg["R"], cfg["K"]

    # ---- synthetic data: only this rank's two slabs are ever materialised -------------
    Vt = synth_V(1, M, T, K)
    plan = ShardPlan(N, M, *(as_rank if as_rank else (rank, world)))
    legacy_data = world == 1 and not as_rank and N * M * T * R <= LEGACY_MAX_VALUES
    if legacy_data:
        Y, Wt_true, Vt = synth_legacy(1, N, M, T, R, K)         # legacy np.random.seed(1) stream
        slabs = None
    elif world == 1 and not as_rank:
        Y, Wt_true = synth_rows(1, range(N), M, T, R, K, Vt)
        slabs = None
    else:
        rows, _ = synth_rows(1, range(plan.row0, plan.row0 + plan.nl), M, T, R, K, Vt)
        cols, _ = synth_rows(1, range(N), M, T, R, K, Vt, cols=(plan.col0, plan.col0 + plan.ml))
        slabs = (rows, cols)
        Y = None

    if args.variant != "complete":
        if world > 1 or as_rank:
            sys.exit("--variant other than complete is single-GPU only (the sharded runs and --as-rank generate the complete-data slabs)")
        rs = np.random.RandomState(7)
        if args.variant == "heldout":
            Y[:3, :3] = np.nan
        elif args.variant == "curves5":
            Y[rs.rand(N, M) < 0.05] = np.nan
        elif args.variant == "missing5":
            Y[rs.rand(N, M) < 0.05] = np.nan
            Y[rs.rand(N, M, T, R) < 0.05] = np.nan
        elif args.variant == "negbinom":
            Mu = np.einsum("nk,mtk->nmt", Wt_true, Vt)
            P = 1 / (1 + np.exp(-Mu))
            Y = rs.negative_binomial(4.0, 1 - np.repeat(P[..., None], R, axis=-1)).astype(float)
        else:
            Mu = np.einsum("nk,mtk->nmt", Wt_true, Vt)
            Ntr = np.full((N, M, T), 4.0)
            Y = (rs.binomial(4, 1 / (1 + np.exp(-Mu))).astype(float), Ntr)
    np.random.seed(1)
    stream = torch.cuda.current_stream().cuda_stream or None     # null stream -> the ctx's own / a dedicated torch stream
    common = dict(nembeds=K, tf_order=2, sigma2_init=0.5, lam2_init=0.1, rng="device",
                  compat="reference" if (world == 1 and not as_rank) else "exact", device=local_rank, stream=stream,
                  shard=as_rank if as_rank else ((rank, world) if world > 1 else None), device_seed=1, sampler=args.sampler,
                  overlap_exchange=args.overlap, rehearse_rank=as_rank is not None)
    if args.variant == "binomial":
        model = BinomialBayesianTensorFiltering(N, M, T, pg_exact=(True if args.pg_exact else False if args.pg_series else None), **common)
    elif args.variant == "negbinom":
        model = NegativeBinomialBayesianTensorFiltering(N, M, T, pg_exact=(True if args.pg_exact else False if args.pg_series else None), **common)
    else:
        model = GaussianBayesianTensorFiltering(N, M, T, nu2_init=1.0, **common)
    if args.rpb != [0, 0]:
        model._ctx.call("btf_set_tuning", args.rpb[0], args.rpb[1])
    if world == 1 and not as_rank:
        data = Y
    else:
        data = _SlabData(slabs, (N, M, T, R))
        model._upload = lambda d, _m=model: _upload_slabs(_m, d)
    # leave the prior draw: full sweeps (nu2, sigma2, Tau2, lam2, W, V)
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
    # One GPU, Gaussian data: the K timed W+V steps are queued by ONE call into the C side (model.wv_steps -> btf_wv_steps:
    # the same launches with the same seeds as K Python-driven steps - tests/test_gpu_parity.py checks the chains coincide -
    # without a round trip through the interpreter between the launches).  The Python-driven loop is timed too, beside it
    # (config.python_driven_sweeps_per_s).
    c_driven = world == 1 and not as_rank and args.variant in ("complete", "heldout", "curves5", "missing5") and not args.python_loop

    def run_steps(k):
        if c_driven:
            model.wv_steps(data, k)
        else:
            for _ in range(k):
                step()

    def fence():
        if world > 1 or exercise:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    fence()
    t0 = time.perf_counter()
    run_steps(args.steps)
    fence()
    dt = time.perf_counter() - t0
    model.sync()                      # raises if any factorisation failed inside the timed region
    py_per_s = None
    if c_driven and not args.lean:    # the same K steps driven from Python, call by call
        for _ in range(args.warmup):
            step()
        fence()
        t0p = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        py_per_s = args.steps / (time.perf_counter() - t0p)
        model.sync()
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- per-step GPU time by events between the steps (median of >= 50: BASELINE.md section 4) ----
    nmed = max(50, min(args.steps, 200))
    tstream = torch.cuda.ExternalStream(model._ctx.stream_handle, device=local_rank) if model._ctx.stream_handle else torch.cuda.current_stream()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(nmed + 1)]
    evs[0].record(tstream)
    for i in range(nmed):
        step()
        evs[i + 1].record(tstream)
    fence()
    per_step_ms = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(nmed)])

    # ---- per-kernel HIP-event timing of the same steps (separate pass: events perturb) ----
    model._ctx.call("btf_set_profiling", 1)
    model._ctx.kernel_times()
    nprof = min(args.steps, 200)
    model._exchange.timing = True
    # five blocks; per kernel the MEDIAN of the blocks' average durations: one stalled dispatch (seen once: a single
    # ~10 ms launch among 400) would otherwise carry the whole figure
    nblk = 5 if nprof >= 50 else 1
    blocks = []
    for b in range(nblk):
        for _ in range(nprof // nblk):
            step()
        fence()
        blocks.append(model._ctx.kernel_times())
    if os.environ.get("BTF_BENCH_DEBUG") == "1" and rank == 0:
        for k in ("w_accum", "w_solve", "v_accum", "v_banded"):
            print("blocks %s: %s" % (k, ["%.2f" % (1e3 * blk[k][0] / max(blk[k][1], 1)) for blk in blocks]), file=sys.stderr)
    if os.environ.get("BTF_BENCH_DEBUG") == "1" and rank == 0 and model.v_sampler() == "spectral":
        import ctypes as C
        eg = np.zeros(K + K * K + 8)
        fn = model._ctx.lib.btf_debug_eig
        fn.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        sw = []
        for _ in range(8):
            step()
            if fn(model._ctx.h, eg.ctypes.data_as(C.POINTER(C.c_double))) == 0:
                sw.append(int(eg[K + K * K]))
        print("eigen side task: Jacobi sweeps of 8 solves (0 = warm refinement) %s; eigenvalues %s" % (sw, np.array2string(eg[:K], precision=4)), file=sys.stderr)
    if os.environ.get("BTF_ACC_STAMPS_OUT") and rank == 0:      # diagnostic builds (-DBTF_ACC_STAMPS, scripts/acc_stamps.sh)
        import ctypes as C
        lib = model._ctx.lib
        lib.btf_debug_acc_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
        dump = {}
        for name, half in (("w", lambda: model._resample_W(data)), ("v", lambda: model._resample_V(data))):
            buf = np.zeros((8192, 8), dtype=np.int64)
            for _ in range(3):
                (model._resample_V(data) if name == "w" else model._resample_W(data))
                fence()
                lib.btf_debug_acc_stamps(model._ctx.h, buf.ctypes.data_as(C.POINTER(C.c_longlong)))      # (reads and clears)
                half()
                fence()
            lib.btf_debug_acc_stamps(model._ctx.h, buf.ctypes.data_as(C.POINTER(C.c_longlong)))
            dump[name] = buf
        np.savez(os.environ["BTF_ACC_STAMPS_OUT"], **dump)
    kt = {}
    for k in blocks[0]:
        avgs = sorted(blk[k][0] / blk[k][1] for blk in blocks if blk[k][1] > 0)
        n = sum(blk[k][1] for blk in blocks)
        kt[k] = (avgs[len(avgs) // 2] * n, n) if avgs else (0.0, 0)
    coll = model._exchange.collective_us()
    model._exchange.timing = False
    model._ctx.call("btf_set_profiling", 0)

    # full Gibbs sweep (nu2, sigma2, Tau2, lam2, W, V), everything drawn on the device
    full_per_s = None
    if not args.lean:
        nfull = max(10, min(args.steps, 100))
        for _ in range(3):
            model.resample(data)
        fence()
        t0 = time.perf_counter()
        if hasattr(model, "resample_sweeps"):
            model.resample_sweeps(data, nfull)      # one GPU, rng="device": the sweeps are queued by the C side (btf_gibbs_sweeps);
        else:                                       # otherwise this is the same loop of resample() calls
            for _ in range(nfull):
                model.resample(data)
        fence()
        full_per_s = nfull / (time.perf_counter() - t0)

    sweeps_per_s = args.steps / dt
    units = world if weak else 1
    value = sweeps_per_s * units

    # algorithmic bytes: the local slab of the linear statistic once per accumulation launch
    # (8 B/cell complete data; + a byte of replicate count with missing data; Binomial: f64 weights + the pseudo-data
    #  as one byte when the counts are integers, else as f64) - asked of the context, which knows what it streams
    cells_local = plan.nl * M * T
    cells_local_v = N * plan.ml * T
    form = model.likelihood_form()     # "curve_counts": held-out whole curves run the complete-data stream (no counts read)
    import ctypes as _C
    _b = _C.c_double()
    model._ctx.call("btf_get_accum_bytes_per_cell", _C.byref(_b))
    bpc = _b.value                     # 8: statistic alone; 9: + byte counts, or byte pseudo-data + f64 weights; 16: f64 + f64
    # (BTF_OPT_FUSED_STEP, the default on complete data at whole-column tiles: the V accumulation launch carries the spectral
    #  sampler as its tail - no launch of its own - and its duration is stream + sampler; the roofline of the streaming kernel
    #  is then taken from the W accumulation launch alone, the same kernel without a tail)
    v_fused = model.v_sampler() == "spectral" and kt.get("v_banded", (0.0, 0))[1] == 0 and kt["v_accum"][1] > 0
    w_fused = kt.get("w_solve", (0.0, 0))[1] == 0 and kt["w_accum"][1] > 0
    plain = [k for k, f in (("w_accum", w_fused), ("v_accum", v_fused)) if not f] or ["w_accum", "v_accum"]
    acc_ms = sum(kt[k][0] for k in plain)
    acc_n = sum(kt[k][1] for k in plain)
    acc_us = 1e3 * acc_ms / max(acc_n, 1)
    alg_bytes = bpc * sum({"w_accum": cells_local, "v_accum": cells_local_v}[k] for k in plain) / len(plain)
    achieved = alg_bytes / (acc_us * 1e-6) / 1e9 if acc_us > 0 else 0.0
    # the same W+V step with the reference-reproducible V sampler (`sampler="banded"`: P'L^-T of a declared ordering,
    # the mode that can walk a seeded reference chain) beside the spectral one the headline uses
    banded_per_s = None
    if model.v_sampler() == "spectral" and world == 1 and not as_rank and args.sampler == "auto" and not args.lean:
        model._ctx.call("btf_set_option", _native.OPT_SAMPLER, _native.SAMPLERS["banded"])
        nb = max(20, min(args.steps, 200))
        for _ in range(10):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(nb):
            step()
        fence()
        banded_per_s = nb / (time.perf_counter() - t0)
        model._ctx.call("btf_set_option", _native.OPT_SAMPLER, _native.SAMPLERS["spectral"])
    # ... and the mode that meets north_star's bar (1) bit for bit - rng="host", sampler="banded", compat="reference": the
    # legacy numpy stream feeds every draw, a fixed seed walks the reference's chain - timed in a leg of its own
    host_rng = None
    if world == 1 and not as_rank and not args.lean and args.variant == "complete" and args.sampler == "auto" and legacy_data:
        try:
            host_rng = host_rng_leg(Y, N, M, T, K, local_rank, stream, fence)
        except Exception as e:         # pragma: no cover  (never let the extra leg cost the headline line)
            host_rng = {"error": repr(e)}
    b_wv = bpc * (cells_local + cells_local_v)                     # B_WV of SURVEY 8(d), per GPU
    if args.variant in ("binomial", "negbinom"):                   # + the PG draw: trials in, omega out (48 B/cell in all)
        b_wv += 16.0 * cells_local
    ms_step = 1e3 * dt / args.steps
    kernels_us = {k: round(1e3 * v[0] / max(v[1], 1), 2) for k, v in kt.items() if v[1] > 0}
    sampler = model.v_sampler()
    out = {
        "metric": metric_for(args.config, world),
        "value": round(value, 2),
        "unit": "sweeps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_step, 4),
        "higher_is_better": True,
        "scaling": "weak" if (weak or world == 1) else "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic" + (" (SURVEY 8(d): legacy np.random.seed(1) stream)" if legacy_data else " (per-row / per-block streams: rank slabs)"),
        "config": {"workload": "%s_btf %s (%d,%d,%d,%d) nembeds=%d tf_order=2 %s data, W+V update, rng=device%s"
                               % ({"binomial": "binomial", "negbinom": "negbinom"}.get(args.variant, "gaussian"), args.config, N, M, T,
                                  {"binomial": 1}.get(args.variant, R), K, args.variant + (" (4 trials per cell)" if args.variant == "binomial" else ""),
                                  "; %d-way row/column shards, %s all-gather of W and V" % (
                                      world, {"rccl": "RCCL", "peer": "peer-window", "host": "host-staged"}.get(model._exchange.transport, "?"))
                                  if world > 1 else ""),
                   "global_sweeps_per_s": round(sweeps_per_s, 2), "units_per_sweep": units,
                   "median_ms_per_step": round(float(np.median(per_step_ms)), 4), "median_over_steps": int(nmed),
                   "burn_in_sweeps": args.burn,
                   "full_resample_sweeps_per_s": None if full_per_s is None else round(full_per_s, 2),
                   "step_loop": "C side (btf_wv_steps: one call queues the K steps)" if c_driven else "Python (one _resample_W / _resample_V call pair per step)",
                   "python_driven_sweeps_per_s": None if py_per_s is None else round(py_per_s, 2),
                   "v_sampler": sampler, "banded_sweeps_per_s": None if banded_per_s is None else round(banded_per_s, 2),
                   "likelihood_form": form,
                   "launches_per_step": 4 - int(v_fused) - int(w_fused),
                   # what pins the mode this line times: the device-RNG chain is checked statistically, the
                   # reference-reproducible chain (rng="host", legacy numpy stream) bit-level against the reference's fixtures
                   "parity_of_this_mode": "rng=device (Philox normals, %s square root): statistical - whitening, KS / Anderson-Darling, "
                                          "moment and conditional-mean tests at this size (tests/test_gpu_fullsize.py); the same kernels under "
                                          "rng=host reproduce the reference's fixtures to 1e-10 (W) / 1e-6 (V) (tests/golden, G1-G9)" % sampler,
                   "parallelism": "rows(W)/cols(V) x%d" % world},
        "roofline": {"bound": "hbm", "kernel": "accum_kernel (%s)" % (" + ".join(plain) + " launches" if len(plain) == 2 else
                                                                       plain[0] + " launch; the other accumulation launch carries its "
                                                                       "sampler / solve as a fused tail: kernels_us." +
                                                                       ("v_accum" if v_fused else "w_accum") + " = stream + tail"),
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": None, "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_us": round(acc_us, 2),
                     "timing": "hipExtLaunchKernelGGL start/stop events of each accumulation dispatch, %d steps (median of five block averages)" % nprof,
                     "whole_step_bytes": b_wv,
                     "whole_step_frac": round(b_wv / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "copy_ceiling_GBs": copy_ceiling(torch), "read_ceiling_GBs": read_ceiling(local_rank)},
        "kernels_us": kernels_us,
    }
    if host_rng is not None:
        out["host_rng"] = host_rng
    # ---- the roofline object against the committed profiles of THIS workload (profiles/README.md) --------------------------
    #  traffic: HBM bytes per accumulation launch from the PMC passes (never another workload's figure);
    #  rocprof_avg_us: rocprofv3 --kernel-trace --stats AverageNs of the accumulation kernels in a `--lean` run of this command
    #  (nothing but W+V steps): when it disagrees with the live HIP events, `achieved` / `frac` report the LOWER of the two.
    rl = out["roofline"]
    sfx = "_rank%dof%d" % as_rank if as_rank else ""
    if world == 1:
        tr = pmc_traffic(args.config, args.variant, sfx)
        if tr:
            rl["traffic"], rl["traffic_source"] = tr
        rp = rocprof_accum_avg(args.config, args.variant, sfx)
        if rp:
            rl["rocprof_avg_us"], rl["rocprof_source"] = round(rp[0], 2), rp[1]
            rl["events_frac"] = rl["frac"]
            ach_rp = alg_bytes / (rp[0] * 1e-6) / 1e9
            if ach_rp < achieved:
                rl["achieved"], rl["frac"] = round(ach_rp, 1), round(ach_rp / HBM_PEAK_GBS, 4)
                rl["frac_basis"] = "rocprofv3 AverageNs (lower than the live events' %.4f)" % rl["events_frac"]
            else:
                rl["frac_basis"] = "live HIP events (rocprofv3 AverageNs of the committed run gives %.4f)" % (ach_rp / HBM_PEAK_GBS)
        if args.config == "c3":
            # C3's 67 MB statistic sits inside the 256 MiB Infinity Cache between sweeps, so its byte counters are not HBM
            # traffic proper: the launch that is - the whole C5 tensor on one GPU, 2.1 GB per launch - from its committed run
            big = hbm_resident_reference()
            if big:
                rl["hbm_resident"] = big
    if v_fused and world == 1 and not as_rank and "v_accum" in kernels_us:
        # the LARGEST launch of the step is not the plain stream but the V accumulation launch that carries its sampler as a
        # tail: the same algorithmic bytes over stream + tail (latency-bound chains behind the stream: DESIGN.md 4.3)
        va = kernels_us["v_accum"]
        rl["largest_launch"] = {"kernel": "accum_kernel FUSE_V%s (V accumulation + spectral sampler tail)" % ("DF" if os.environ.get("BTF_VF_DATAFLOW", "1") != "0" else ""),
                                "avg_launch_us": va, "algorithmic_bytes": bpc * cells_local_v,
                                "frac": round(bpc * cells_local_v / (va * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                "timing": "live HIP events"}
    if args.variant in ("binomial", "negbinom"):
        out["config"]["pg_sampler"] = getattr(model, "pg_sampler", "series")
    if world > 1 or exercise:
        # the data-path collectives are the library's own RCCL calls (btf_allgather_W / _V on the context's communicator);
        # torch.distributed carried the communicator id, the barrier and the max over ranks of the clock
        out["config"].update({"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(),
                              "exchange_transport": model._exchange.transport,
                              "communicator": model._exchange.comm_info() if model._exchange.transport in ("rccl", "peer") else None,
                              "collective_us": coll})
    if as_rank:
        # What ONE rank of a P-GPU run does per step, measured on one GPU: its kernels on its real slabs and the RCCL call
        # sequence (one-rank group: launch + local copy of the full message, no wire time).  The projection adds nothing for
        # the wire (4.2 MB of V per step over 7 xGMI links of ~153 GB/s: a few us, overlapped with the own-block chunks)
        # and assumes the ranks stay in step.  A rehearsal, not a measurement of P GPUs.
        ref = same_config_one_gpu(args.config, "complete")
        step_us = 1e3 * ms_step
        out["config"]["workload"] += "; REHEARSAL of rank %d of %d on one GPU (row slab %d x %d, column slab %d x %d)" % (
            as_rank[0], as_rank[1], plan.nl, M, N, plan.ml)
        out["projected"] = {"what": "rank %d of %d: W+V step with both all-gathers (btf_allgather_W / _V on a one-rank RCCL communicator of the context, full message sizes)" % as_rank,
                            "communicator": model._exchange.comm_info(),
                            "ranks": as_rank[1], "per_rank_step_us": round(step_us, 1),
                            "projected_sweeps_per_s": round(1e6 / step_us, 1),
                            "exchange": "overlapped (own-block chunks ahead of the all-gather)" if args.overlap else "in line",
                            "collective_us": coll,
                            "one_gpu_same_workload": ref,
                            "projected_speedup_vs_one_gpu": round((1e6 / step_us) / ref["value"], 2) if ref else None,
                            "not_measured": "wire time and skew between %d real ranks" % as_rank[1]}
    if world > 1:
        # the one-GPU figure of THIS workload (a `--gpus 1` run reports the headline C3 instead): strong scaling - the
        # whole fixed tensor on one GPU; weak - one rank's slab.  From the committed profile of that run, with its file.
        ref = same_config_one_gpu(args.config, "complete")      # (weak: args.config names one rank's slab)
        if ref is not None:
            ref["note"] = "builder-run committed figure of the SAME workload on one MI355X; the driver's --gpus 1 run is the C3 headline, another workload"
            out["config"]["one_gpu_same_workload"] = ref
            out["config"]["speedup_vs_one_gpu_same_workload"] = round(sweeps_per_s / ref["value"], 3) if not weak else None
    if world == 1 and not as_rank and not args.no_cpu and not args.lean and args.variant not in ("binomial", "negbinom"):
        out["cpu_baseline"] = cpu_baseline(Y, model, cfg)
    if world == 1 and not as_rank and not args.no_cpu and not args.lean and args.variant == "binomial":
        out["cpu_baseline"] = cpu_baseline_binomial(Y, model, cfg)
    if args.variant == "binomial" and world == 1 and "pg_draw" in kernels_us:
        # the step's dominant kernel is the exact Polya-Gamma draw - bound by VALU issue, not by bytes: its instruction count
        # from the committed PMC pass of this workload over the live kernel time, against one wave64 VALU instruction per
        # 4 cycles and SIMD (1024 SIMDs, 2.4 GHz peak clock)
        vi = valu_insts("c3" if args.config == "c3" else args.config, "binomial", "pgx_tile_kernel")
        if vi:
            peak = 1024 * 2.4e9 / 4 / 1e9
            ach = vi[0] / (kernels_us["pg_draw"] * 1e-6) / 1e9
            out["roofline_dominant"] = {"kernel": "pgx_tile_kernel (exact Polya-Gamma draw)", "bound": "valu", "achieved": round(ach, 1), "peak": round(peak, 1),
                                        "unit": "G wave-instructions/s", "frac": round(ach / peak, 4), "avg_launch_us": kernels_us["pg_draw"],
                                        "valu_insts_per_launch": vi[0], "source": vi[1]}

    # BASELINE config 4 beside the headline (the default single-GPU run only): the Binomial model at the same (512,256,64),
    # 4 trials per cell, the device Polya-Gamma draw inside the step - a short leg of its own, so that the driver's run sees
    # a C4 number too (`python bench.py --variant binomial` is the full line of that workload)
    if world == 1 and not as_rank and not args.lean and args.config == "c3" and args.variant == "complete" and not args.no_c4:
        try:
            del model
            out["config"]["c4_binomial"] = c4_leg(N, M, T, K, Vt, local_rank, stream, fence, torch)
        except Exception as e:         # pragma: no cover  (never let the extra leg cost the headline line)
            out["config"]["c4_binomial"] = {"error": repr(e)}
    if rank == 0:
        print(json.dumps(out))
    if world > 1 or exercise:
        dist.barrier()
        dist.destroy_process_group()


def host_rng_leg(Y, N, M, T, K, device, stream, fence, steps=40, sweeps=20):
    """The reference-reproducible mode beside `value`: rng="host" (every normal and gamma from the legacy global numpy stream,
    in the reference's order: SURVEY 8(a) Q4), sampler="banded" (P'L^-T z of a declared elimination order, fast_mvn.py:38-47),
    compat="reference".  This is the instantiation the golden fixtures pin bit for bit (G1-G6); it is host-bound by construction -
    82 k legacy polar normals per step and their upload - and is the default a user of INTEGRATION.md section A gets."""
    from functionalmf_amd.factor import GaussianBayesianTensorFiltering
    np.random.seed(1)
    m = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, sigma2_init=0.5, lam2_init=0.1, nu2_init=1.0, rng="host",
                                        compat="reference", sampler="banded", device=device, stream=stream)
    for _ in range(3):
        m.resample(Y)

    def step():
        m._resample_W(Y)
        m._resample_V(Y)
    for _ in range(5):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    m.sync()
    fence()
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(sweeps):
        m.resample(Y)
    m.sync()
    fence()
    dts = time.perf_counter() - t0
    nz = K * (K + 1) // 2 + (N - K) * K + M * K * T
    t0 = time.perf_counter()
    for _ in range(10):
        np.random.normal(size=nz)
    tz = (time.perf_counter() - t0) / 10
    return {"mode": "rng=host, sampler=banded, compat=reference: the fixed-seed chain of the reference (bit-level fixtures G1-G6)",
            "wv_steps_per_s": round(steps / dt, 1), "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps,
            "full_sweeps_per_s": round(sweeps / dts, 1), "sweeps": sweeps, "v_sampler": m.v_sampler(),
            "host_normals_per_step": nz, "legacy_normals_ms_per_step": round(1e3 * tz, 3),
            "bound": "host: the legacy MT19937 / polar-method stream the reference draws from, then one upload per half-sweep"}


def c4_leg(N, M, T, K, Vt, device, stream, fence, torch, steps=200):
    """BASELINE config 4: Binomial BTF (512,256,64), 4 trials per cell, nembeds 5; a step = device Polya-Gamma draw of all
    cells (exact sampler, factor.py:459) + W half-sweep + V half-sweep (factor.py:425-460), rng="device"."""
    from functionalmf_amd.factor import BinomialBayesianTensorFiltering
    rs = np.random.RandomState(7)
    Mu = np.einsum("nk,mtk->nmt", synth_rows(1, range(N), M, T, 1, K, Vt, noise=0.0)[1], Vt)
    data = (rs.binomial(4, 1 / (1 + np.exp(-Mu))).astype(float), np.full((N, M, T), 4.0))
    np.random.seed(1)
    m = BinomialBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, sigma2_init=0.5, lam2_init=0.1, rng="device", compat="reference",
                                        device=device, stream=stream, device_seed=1)
    for _ in range(5):
        m.resample(data)

    def step():
        m._resample_nu2(data)
        m._resample_W(data)
        m._resample_V(data)
    for _ in range(10):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    m.sync()
    m._ctx.call("btf_set_profiling", 1)
    m._ctx.kernel_times()
    for _ in range(50):
        step()
    fence()
    kt = m._ctx.kernel_times()
    m._ctx.call("btf_set_profiling", 0)
    cells = N * M * T
    return {"workload": "binomial_btf c3 (%d,%d,%d) 4 trials per cell nembeds=%d: Polya-Gamma draw + W + V per step, rng=device" % (N, M, T, K),
            "steps_per_s": round(steps / dt, 1), "ms_per_step": round(1e3 * dt / steps, 4), "steps": steps,
            "pg_sampler": getattr(m, "pg_sampler", None),
            "kernels_us": {k: round(1e3 * v[0] / max(v[1], 1), 2) for k, v in kt.items() if v[1] > 0},
            "whole_step_frac_48B_per_cell": round(48.0 * cells / (dt / steps) / 1e9 / HBM_PEAK_GBS, 4)}


def dry_run(args, world, rank, dist, backend):
    """BTF_BENCH_DRY=1: no GPU.  Exercises what the driver depends on - rank launch, barriers, max-over-ranks
    timing, one JSON line from rank 0 - with a host all-gather standing in for a step (tests/test_host_logic.py)."""
    import torch
    x = torch.zeros(64, dtype=torch.float64)
    def step():
        if world > 1:
            outs = [torch.empty_like(x) for _ in range(world)]
            dist.all_gather(outs, x + rank)
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    out = {"metric": metric_for(args.config, world), "value": round(args.steps / dt, 2), "unit": "sweeps/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True,
           "scaling": "weak" if (args.weak or world == 1) else "strong", "vs_baseline": None, "dtype": "f64", "data": "dry-run (no GPU work)",
           "config": {"workload": "DRY RUN %s x%d" % (args.config, world), "rccl_ranks": dist.get_world_size() if world > 1 else 1,
                      "backend": backend}}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def read_ceiling(device):
    """Measured rate of a plain streaming read on this GPU (1 GiB, past the Infinity Cache; btf_read_probe): what a
    read-only kernel like the accumulation can reach at best, next to the 8 TB/s spec peak."""
    try:
        import ctypes as C
        from functionalmf_amd import _native
        out = C.c_double()
        rc = _native.load().btf_read_probe(int(device), 1 << 30, 10, C.byref(out))
        return round(out.value, 1) if rc == 0 else None
    except Exception:
        return None


def copy_ceiling(torch):
    """Measured device-to-device copy rate on this GPU (read + write bytes per second, 1 GiB buffers: past the
    256 MiB Infinity Cache) - the achievable ceiling next to the 8 TB/s spec peak."""
    try:
        n = 1 << 27
        a = torch.empty(n, dtype=torch.float64, device="cuda")
        b = torch.empty_like(a)
        a.fill_(1.0)
        for _ in range(3):
            b.copy_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        del a, b
        return round(2 * n * 8 / (ms * 1e-3) / 1e9, 1)
    except Exception:
        return None


def same_config_one_gpu(config, variant):
    """sweeps/s of `python bench.py --gpus 1 --config <config>` as committed under profiles/ (the bench line kept beside
    the rocprofv3 run of that workload), for the denominator of a scaling figure; None if no such file."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_%s_bench.json" % (config, variant))))
    if not files:
        return None
    try:
        d = json.loads(open(files[-1]).read().strip().splitlines()[-1])
        return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "n_gpus": d["n_gpus"],
                "source": os.path.relpath(files[-1], ROOT)}
    except Exception:
        return None


def pmc_traffic(config, variant, suffix=""):
    """HBM bytes per accumulation launch from a committed rocprofv3 PMC pass OF THIS WORKLOAD
    (profiles/r*_pmc_<config>_<variant><suffix>.json: 2*FETCH_SIZE + WRITE_SIZE, the gfx950 correction of
    MI355X_MICROARCH.md applied; suffix "_rank0of8" for a rehearsed rank's slabs); None if no such profile is committed
    (never another workload's figure)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_%s_%s%s.json" % (config, variant, suffix))))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        # (the instances of THIS variant's likelihood form: the default run's short C4 leg leaves Binomial accumulation
        #  launches in a complete-data profile)
        unweighted = variant in ("complete", "heldout", "curves5")
        vals = [(k, v["hbm_bytes_per_launch_corrected"]) for k, v in d.items()
                if (mode := _plain_accum(k)) is not None and ((mode == "0") == unweighted)]
        # (the timed W+V steps launch the FUSE_LEAN instance "..., 4>"; the full instance "..., 0>" beside it in the same
        #  profile is the burn-in sweeps' launch, whose side tasks read the Tau2 chain as well)
        lean = [x for x in vals if x[0].split("(")[0].rstrip().endswith(", 4>")]
        vals = [x[1] for x in (lean or vals)]
        return (round(max(vals), 1), os.path.relpath(files[-1], ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; 2 x FETCH + WRITE)") if vals else None
    except Exception:
        return None


def valu_insts(config, variant, kernel):
    """(SQ_INSTS_VALU per launch, file) of `kernel` from the committed PMC pass of this workload
    (profiles/r*_pmc_valu_<config>_<variant>.json), or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_valu_%s_%s.json" % (config, variant))))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        for k, v in d.items():
            if kernel in k and isinstance(v, dict) and "SQ_INSTS_VALU" in v:
                return float(v["SQ_INSTS_VALU"]), os.path.relpath(files[-1], ROOT)
    except Exception:
        pass
    return None


def _plain_accum(name):
    """Is this kernel name a plain accumulation launch - accum_kernel<K, MODE, WAVES, TX, TC, UNR, OPL, FUSE> with FUSE == 0
    or 4 (FUSE_LEAN: the same stream compiled without the gamma-drawing side tasks; FUSE 1 / 2 / 3: the launch carries the W
    solves / the spectral sampler as its tail, csrc/btf_fused.h - its duration is not the stream's)?  Returns the
    likelihood MODE as a string, or None."""
    import re
    m = re.search(r"accum_kernel<\d+, (\d+), \d+, [^,]+, [^,]+, \d+, \d+, (\d+)>", name)
    if m:
        return m.group(1) if m.group(2) in ("0", "4") else None
    m = re.search(r"accum_kernel<\d+, (\d+), \d+, [^,]+, [^,]+, \d+, \d+>", name)      # (profiles of rounds 2-3: no FUSE argument)
    return m.group(1) if m else None


def rocprof_accum_avg(config, variant, suffix=""):
    """(average us, file) of the accumulation kernels in the newest committed `rocprofv3 --kernel-trace --stats` summary of a
    --lean run of this workload (profiles/r*_<config>_<variant><suffix>_lean_kernel_stats.csv: calls-weighted AverageNs
    of every accum_kernel instance); None if no such file is committed."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_%s%s_lean_kernel_stats.csv" % (config, variant, suffix))))
    if not files:
        return None
    try:
        tot = calls = 0.0
        for r in csv.DictReader(open(files[-1])):
            if _plain_accum(r["Name"]) is not None:      # (not the launches that carry a fused tail)
                tot += float(r["TotalDurationNs"]); calls += float(r["Calls"])
        return (1e-3 * tot / calls, os.path.relpath(files[-1], ROOT)) if calls else None
    except Exception:
        return None


def hbm_resident_reference():
    """The accumulation launch of the whole C5 tensor on one GPU (2.147 GB per launch, 8 x the Infinity Cache) from its
    committed run - builder-run, not this run: bench line + PMC traffic + rocprofv3 average where committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_c5_complete_bench.json")))
    if not files:
        return None
    try:
        d = json.loads(open(files[-1]).read().strip().splitlines()[-1])
        r = d["roofline"]
        out = {"workload": d["config"]["workload"], "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"],
               "avg_launch_us": r["avg_launch_us"], "achieved": r["achieved"], "frac": r["frac"], "traffic": r.get("traffic"),
               "source": os.path.relpath(files[-1], ROOT) + " (builder-run on one MI355X, not this run)"}
        rp = rocprof_accum_avg("c5", "complete")
        if rp:
            out["rocprof_avg_us"], out["rocprof_source"] = round(rp[0], 2), rp[1]
        return out
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


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(Y, model, cfg):
    """The oracle's reference-faithful W+V update (per-row / per-column loops, the sufficient statistics
    re-reduced from the 4-D tensor on every half-sweep as factor.py:329-330/:374-375 do, dense LAPACK in
    place of CHOLMOD) on the host cores, from the GPU chain's current state: with all BLAS threads and with
    one thread; bounded to roughly 30 s in all."""
    from oracle import btf_oracle as orc
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threadpool_limits = None
        cores = os.cpu_count() or 1
    st = dict(W=model.W.copy(), V=model.V.copy(), Tau2=np.array(model.Tau2, dtype=float).copy(),
              lam2=float(model.lam2), sigma2=float(model.sigma2), nu2=float(np.asarray(model.nu2).reshape(-1)[0]))
    Delta = orc.trend_penalty(cfg["T"], 2)

    def faithful(budget, nmax):
        np.random.seed(123)
        n, t0 = 0, time.perf_counter()
        while True:
            orc.w_step(st, Y)
            orc.v_step(st, Y, Delta)
            n += 1
            el = time.perf_counter() - t0
            if el > budget or n >= nmax:
                return n, el
    n, el = faithful(10.0, 50)
    out = {"value": round(n / el, 4), "unit": "sweeps/s", "cores": int(cores), "kind": "port", "cpu_model": cpu_model(),
           "sample": "%d full W+V updates of the same (%d,%d,%d,%d) K=%d tensor by oracle/btf_oracle.py "
                     "(numpy/LAPACK, BLAS threads=%d)" % (n, cfg["N"], cfg["M"], cfg["T"], cfg["R"], cfg["K"], cores)}
    if threadpool_limits is not None:
        with threadpool_limits(limits=1):
            n1, el1 = faithful(8.0, 20)
        out["single_thread_value"] = round(n1 / el1, 4)
        out["single_thread_sample"] = "%d W+V updates, same code, BLAS / OpenMP limited to 1 thread" % n1
    # second CPU number: statistics hoisted, BLAS + banded LAPACK ("strong CPU")
    try:
        Rr, ybar = orc.hoisted_stats(Y)
        n2, t0 = 0, time.perf_counter()
        while True:
            orc.w_step_strong(st, Rr, ybar)
            orc.v_step_strong(st, Rr, ybar, Delta)
            n2 += 1
            el2 = time.perf_counter() - t0
            if el2 > 6.0 or n2 >= 200:
                break
        out["strong_cpu_value"] = round(n2 / el2, 3)
        out["strong_cpu_sample"] = "%d W+V updates, hoisted statistics + BLAS + scipy banded Cholesky" % n2
    except Exception as e:        # pragma: no cover
        out["strong_cpu_value"] = None
        out["strong_cpu_sample"] = "failed: %r" % (e,)
    return out


def cpu_baseline_binomial(data, model, cfg):
    """Config C4 on the host cores: the oracle's Binomial step - Polya-Gamma weights omega ~ PG(N, w.v) for every cell
    (factor.py:447-460), then the weighted W and V half-sweeps on kappa = (Y - N/2)/omega (factor.py:437-445 with :313-409).
    pypolyagamma is absent: the draws are the oracle's definition-based series sampler
    (pg_draw_series_cells, 200 gamma terms + the tail's mean - what PyPolyaGamma's own truncated sampler does), timed on a
    SAMPLE of the cells and scaled to all of them; the two half-sweeps are timed on the whole tensor, once."""
    from oracle import btf_oracle as orc
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    Ys, Ntr = data
    W, V = model.W.copy(), model.V.copy()
    st = dict(W=W, V=V, Tau2=np.array(model.Tau2, dtype=float).copy(), lam2=float(model.lam2), sigma2=float(model.sigma2))
    psi = np.einsum("nk,mtk->nmt", W, V)
    cells = psi.size
    ns = min(cells, 1 << 19)
    rng = np.random.RandomState(5)
    idx = rng.choice(cells, ns, replace=False)
    t0 = time.perf_counter()
    om_s = orc.pg_draw_series_cells(float(np.nanmax(Ntr)), psi.reshape(-1)[idx], rng)
    t_pg = (time.perf_counter() - t0) * cells / ns
    omega = np.full(cells, float(np.mean(om_s)))
    omega[idx] = om_s
    with np.errstate(divide="ignore"):
        st["nu2"] = np.where(np.isnan(Ys), np.inf, 1.0 / omega.reshape(psi.shape))
    Delta = orc.trend_penalty(cfg["T"], 2)
    np.random.seed(123)
    t0 = time.perf_counter()
    orc.binomial_w_step(st, Ys, Ntr)
    orc.binomial_v_step(st, Ys, Ntr, Delta)
    t_wv = time.perf_counter() - t0
    return {"value": round(1.0 / (t_pg + t_wv), 4), "unit": "sweeps/s", "cores": int(cores), "kind": "port", "cpu_model": cpu_model(),
            "sample": "Polya-Gamma draws of %d of the %d cells by oracle.pg_draw_series_cells (%.2f s, scaled to %.1f s for all cells; one "
                      "thread) + one weighted W and one weighted V half-sweep of the whole (%d,%d,%d) K=%d tensor by oracle.binomial_w_step / "
                      "binomial_v_step (%.1f s; numpy/LAPACK, BLAS threads=%d)" % (ns, cells, t_pg * ns / cells, t_pg, cfg["N"], cfg["M"], cfg["T"],
                                                                                   cfg["K"], t_wv, cores)}


if __name__ == "__main__":
    main()
