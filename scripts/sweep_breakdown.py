"""Where a full rng="device" Gibbs sweep spends its wall time (host-side view)."""
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
for _ in range(5):
    m.resample(Y)
m.sync()
steps = [("nu2", lambda: m._resample_nu2(Y)), ("sigma2", m._resample_sigma2), ("Tau2", m._resample_Tau2),
         ("lam2", m._resample_lam2), ("W", lambda: m._resample_W(Y)), ("V", lambda: m._resample_V(Y))]
acc = {k: 0.0 for k, _ in steps}
n = 200
t_all = time.perf_counter()
for _ in range(n):
    for k, f in steps:
        t0 = time.perf_counter(); f(); acc[k] += time.perf_counter() - t0
m.sync()
t_all = time.perf_counter() - t_all
print("full sweep %.1f us" % (1e6 * t_all / n))
for k in acc:
    print("  %-7s %.1f us host time" % (k, 1e6 * acc[k] / n))
# device-side view: per-kernel event times of the same full sweeps
m._ctx.call("btf_set_profiling", 1)
try:
    kt0 = m._ctx.kernel_times()
    for _ in range(100):
        m.resample(Y)
    m.sync()
    kt = m._ctx.kernel_times()
    print("per-kernel us per sweep (events):", {k: round(1e3 * (kt[k][0] - kt0[k][0]) / 100, 2) for k in kt if kt[k][1] > kt0[k][1]},
          "launches per sweep:", {k: (kt[k][1] - kt0[k][1]) / 100 for k in kt if kt[k][1] > kt0[k][1]})
except Exception as e:
    print("kernel times unavailable:", e)
