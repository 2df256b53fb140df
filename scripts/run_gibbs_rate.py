"""End-to-end run_gibbs throughput (every sweep kept) at (512,256,64,4) K=5, rng="device"."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functionalmf_amd.factor import GaussianBayesianTensorFiltering
from bench import synth_rows, synth_V
N, M, T, R, K = 512, 256, 64, 4, 5
Vt = synth_V(1, M, T, K)
Y, _ = synth_rows(1, range(N), M, T, R, K, Vt)
np.random.seed(1)
m = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, sigma2_init=0.5, lam2_init=0.1, nu2_init=1.0, rng="device")
m.run_gibbs(Y, nburn=20, nthin=1, nsamples=5, verbose=False)
for nthin in (1, 10):
    S = 300
    t0 = time.perf_counter()
    res = m.run_gibbs(Y, nburn=0, nthin=nthin, nsamples=S, verbose=False)
    dt = time.perf_counter() - t0
    print("run_gibbs nthin=%d: %d sweeps, %d kept, %.3f s -> %.0f sweeps/s; result keys %s W%s V%s" % (
        nthin, S * nthin, S, dt, S * nthin / dt, sorted(res), res["W"].shape, res["V"].shape), flush=True)
