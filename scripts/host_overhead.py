"""Diagnostic: is the W+V step bound by the GPU or by the host path (Python + ctypes + HIP launches)?
Times (a) the enqueue loop alone, (b) enqueue + drain, (c) a null-kernel-free Python path cost."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from functionalmf_amd.factor import GaussianBayesianTensorFiltering
from bench import synth_rows, synth_V

N, M, T, R, K = 512, 256, 64, 4, 5
Vt = synth_V(1, M, T, K)
Y, _ = synth_rows(1, range(N), M, T, R, K, Vt)
np.random.seed(1)
m = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, sigma2_init=0.5, lam2_init=0.1, nu2_init=1.0, rng="device",
                                    sampler=os.environ.get("BTF_SAMPLER", "auto"))
for _ in range(3):
    m.resample(Y)
m.sync()
for n in (200, 1000):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        m._resample_W(Y); m._resample_V(Y)
    t1 = time.perf_counter()
    m.sync()
    t2 = time.perf_counter()
    print("n=%d: enqueue %.1f us/step, total %.1f us/step" % (n, 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n))
lib = m._ctx.lib
t0 = time.perf_counter()
for _ in range(2000):
    m._bind_data(Y); m._set_noise(); m._push_state(); m._next_seed()
print("python bookkeeping per half-step: %.2f us" % (1e6 * (time.perf_counter() - t0) / 2000))
out = np.zeros(K + K * K + 8)
print("sweeps of the last eigen-solve:", end=" ")
import ctypes as C
print(m.v_sampler())
