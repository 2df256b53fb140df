"""Host-side cost of queueing W+V steps (Python + ctypes + launches) against the GPU time of the same steps (C3)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from functionalmf_amd.factor import GaussianBayesianTensorFiltering
N, M, T, R, K = 512, 256, 64, 4, 5
Vt = bench.synth_V(1, M, T, K)
Y, _ = bench.synth_rows(1, range(N), M, T, R, K, Vt)
np.random.seed(1)
m = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, sigma2_init=0.5, lam2_init=0.1, nu2_init=1.0, rng="device", device_seed=1)
for _ in range(10):
    m.resample(Y)
m.sync()
for n in (20, 20, 200, 2000):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        m._resample_W(Y); m._resample_V(Y)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("steps %5d: submit %.1f us/step, total %.1f us/step, sync wait %.1f us" % (n, 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n, 1e6 * (t2 - t1)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(300):
    m._resample_W(Y); m._resample_V(Y)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
