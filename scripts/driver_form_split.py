"""Where the 20-step form of the bench loses its ~2 us per step against the 2000-step form (C3, default build): host
timestamps around the timed region, HIP events on the context's stream at its two ends, and the per-step event gaps."""
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
st = torch.cuda.ExternalStream(m._ctx.stream_handle, device=0) if m._ctx.stream_handle else torch.cuda.current_stream()
rows = []
for rep in range(8):
    for _ in range(5):
        m._resample_W(Y); m._resample_V(Y)
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    t0 = time.perf_counter()
    evs[0].record(st)
    for i in range(20):
        m._resample_W(Y); m._resample_V(Y)
        evs[i + 1].record(st)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    gaps = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(20)]) * 1e3
    rows.append((1e6 * (t2 - t0), 1e6 * (t1 - t0), gaps.sum(), gaps[0], np.median(gaps[2:]), gaps[-1]))
r = np.array(rows)
print("wall %.0f us, submit done at %.0f us, GPU (first event to last) %.0f us; first step %.1f us, median step %.1f us, last step %.1f us"
      % tuple(np.median(r, axis=0)))
print("=> outside the events (launch of the first event + return of the sync): %.0f us" % np.median(r[:, 0] - r[:, 2]))
