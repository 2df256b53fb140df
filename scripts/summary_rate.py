"""Posterior summary (mean + 5/95 percentiles) of S kept samples at (512,256,64) K=5 on the GPU."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functionalmf_amd.utils import posterior_summary
N, M, T, K = 512, 256, 64, 5
for S in (100, 1000):
    rs = np.random.RandomState(0)
    Ws = rs.normal(size=(S, N, K)); Vs = rs.normal(size=(S, M, T, K))
    posterior_summary(Ws[:2], Vs[:2])          # warm up
    t0 = time.perf_counter()
    mean, quant = posterior_summary(Ws, Vs, q=(5, 95))
    dt = time.perf_counter() - t0
    i, j, t = 7, 11, 13
    mu = np.einsum("zk,zk->z", Ws[:, i], Vs[:, j, t])
    assert abs(mean[i, j, t] - mu.mean()) < 1e-12 and np.allclose(quant[:, i, j, t], np.percentile(mu, (5, 95)), atol=1e-12)
    print("S=%d: %.3f s wall (upload %.0f MB + kernel + download %.0f MB); the (S,N,M,T) tensor would be %.1f GB"
          % (S, dt, (Ws.nbytes + Vs.nbytes) / 1e6, 3 * mean.nbytes / 1e6, S * N * M * T * 8 / 1e9), flush=True)
