"""Per-launch times of FULL device sweeps (nu2, sigma2, Tau2, lam2, W, V: btf_gibbs_sweeps) against bare W+V steps (C3)."""
import sys, os, time
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
m.resample_sweeps(Y, 20)
m.sync()
for name, fn in (("full sweeps", lambda n: m.resample_sweeps(Y, n)), ("W+V steps", lambda n: [(m._resample_W(Y), m._resample_V(Y)) for _ in range(n)])):
    fn(50); m.sync()
    t0 = time.perf_counter(); fn(1000); m.sync(); dt = time.perf_counter() - t0
    m._ctx.call("btf_set_profiling", 1); m._ctx.kernel_times()
    fn(200); m.sync()
    kt = m._ctx.kernel_times(); m._ctx.call("btf_set_profiling", 0)
    print("%-12s %.1f us each; launches (us, count per 200): %s" % (name, 1e6 * dt / 1000, {k: (round(1e3 * v[0] / v[1], 2), v[1]) for k, v in kt.items() if v[1]}))
