"""Worker of test_two_ranks_share_one_gpu: two processes (gloo control plane, exchange staged
through the host because RCCL refuses two ranks on one device) run the SHARDED kernels - row
and column offsets, local slabs, per-rank partial SSE - on cuda:0 and must reproduce the
unsharded oracle.
Also the worker of test_rccl_exchange_with_one_rank_reproduces_the_plain_chain (BTF_DIST_BACKEND=nccl,
BTF_EXERCISE_EXCHANGE=1, one rank): the same checks with the DEVICE collectives - all_gather_into_tensor on the
context's W / V buffers and the 8-byte all-reduce of the residual sum of squares, issued under the context's own
stream - in the call sequence of an N-rank RCCL run."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, state_from  # noqa: E402
from oracle import btf_oracle as orc  # noqa: E402
from functionalmf_amd.factor import GaussianBayesianTensorFiltering  # noqa: E402


def main():
    backend = os.environ.get("BTF_DIST_BACKEND", "gloo")
    exercise = os.environ.get("BTF_EXERCISE_EXCHANGE", "0")
    if backend == "nccl":
        import torch
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    for name in ("g2_c2_complete.npz", "g1_c1_heldout.npz"):
        g = load_golden(name)
        N, M, T, R, K, tf = [int(x) for x in g["dims"]]
        st = state_from(g, "s0_")
        model = GaussianBayesianTensorFiltering(
            N, M, T, nembeds=K, tf_order=tf, sigma2_init=st["sigma2"], lam2_init=st["lam2"], nu2_init=st["nu2"],
            W_init=st["W"], V_init=st["V"], Tau2_init=st["Tau2"], compat="exact", shard=(rank, world), device=0)
        assert model._exchange.active and model._plan.world == world
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
                W_init=st["W"], V_init=st["V"], compat="exact", shard=shard, device=0, rng="device", device_seed=9,
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
                                            W_init=st["W"], V_init=st["V"], compat="exact", shard=shard, device=0, rng="device",
                                            device_seed=9)
        for _ in range(2):
            m.resample((g["Ysucc"], g["Ntrials"]))
        chains.append((m.W.copy(), m.V.copy(), np.array(m.nu2).copy()))
        del m
    a, b = chains
    assert np.abs(a[2] - b[2]).max() / np.abs(b[2]).max() < 1e-12          # the same Polya-Gamma draws, cell by cell
    assert np.abs(a[0] - b[0]).max() / np.abs(b[0]).max() < 1e-7 and np.abs(a[1] - b[1]).max() / np.abs(b[1]).max() < 1e-5
    print("SHARD_GPU_OK rank", rank, flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
