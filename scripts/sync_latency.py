"""Wake-up cost of the fence that closes a short timed region: torch.cuda.synchronize() alone against a host spin on
stream.query() in front of it (C3, 20 W+V steps per region - the driver's bench form)."""
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
def fence(spin):
    if spin:
        while not st.query():
            pass
    torch.cuda.synchronize()
res = {False: [], True: []}
for rep in range(12):
    for spin in (False, True):
        for _ in range(5):
            m._resample_W(Y); m._resample_V(Y)
        fence(spin)
        t0 = time.perf_counter()
        for _ in range(20):
            m._resample_W(Y); m._resample_V(Y)
        fence(spin)
        res[spin].append(1e6 * (time.perf_counter() - t0) / 20)
for spin in (False, True):
    a = np.array(res[spin])
    print("spin" if spin else "sync", "us/step: median %.2f min %.2f max %.2f" % (np.median(a), a.min(), a.max()))
