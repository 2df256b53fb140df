"""Time of the Negative-Binomial rate update with one rate per row (rdims=(1,2), examples/negbinom_tensor_filtering.py) at
(512,256,64), 30 MH steps, rng="device": one launch (default) against BTF_NB_MH_STEPWISE=1 (two launches per step)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functionalmf_amd.factor import NegativeBinomialBayesianTensorFiltering

rs = np.random.RandomState(1)
N, M, T, K = 512, 256, 64, 5
W = 0.5 * rs.normal(size=(N, K)); V = 0.2 * np.cumsum(rs.normal(size=(M, T, K)), axis=1)
P = 1 / (1 + np.exp(-np.clip(np.einsum("nk,mtk->nmt", W, V), -4, 4)))
data = rs.negative_binomial(4.0, 1 - P).astype(float)
np.random.seed(2)
m = NegativeBinomialBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, sigma2_init=0.5, lam2_init=0.1, rdims=(1, 2),
                                            nmetropolis=30, rng="device", device_seed=3)
m._bind_data(data)
for _ in range(5):
    m._resample_R(data)
m.sync()
t0 = time.perf_counter()
n = 50
for _ in range(n):
    m._resample_R(data)
m.sync()
print("rate update, one rate per row, 30 MH steps: %.1f us (BTF_NB_MH_STEPWISE=%s)" % (1e6 * (time.perf_counter() - t0) / n, os.environ.get("BTF_NB_MH_STEPWISE", "0")))
