"""Rate of whole constrained (GASS) sweeps at a given size, rng="device".  python scripts/gass_rate.py [N M T K ngrid]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functionalmf_amd.factor import ConstrainedNonconjugateBayesianTensorFiltering

N, M, T, K, ngrid = [int(a) for a in sys.argv[1:6]] if len(sys.argv) >= 6 else (512, 256, 64, 5, 100)
rs = np.random.RandomState(1)
Wt = rs.gamma(2.0, 0.5, size=(N, K)); Wt[np.triu_indices(K, 1)] = 0
Vt = np.zeros((M, T, K))
for j in range(M):
    Vt[j, -1] = rs.gamma(2.0, 0.5, size=K)
    for t in range(T - 2, -1, -1):
        Vt[j, t] = Vt[j, t + 1] + (rs.gamma(1.0, 0.6, size=K) if rs.rand() < 0.3 else 0.0)
Y = rs.poisson(np.einsum("nk,mtk->nmt", Wt, Vt)).astype(float)
Cons = np.concatenate([np.eye(T), np.zeros((T, 1))], axis=1)
mono = np.array([np.concatenate([np.zeros(t), [1, -1], np.zeros(T - t - 2), [-1e-2]]) for t in range(T - 1)])
Cons = np.concatenate([Cons, mono], axis=0)
np.random.seed(2)
m = ConstrainedNonconjugateBayesianTensorFiltering(N, M, T, "poisson_identity", Cons, gass_ngrid=ngrid, nembeds=K, tf_order=0,
                                                   sigma2_init=1.0, lam2_init=0.5, W_init=Wt, V_init=Vt, rng="device", device_seed=1)
for _ in range(3):
    m.resample(Y)
m.sync()
m._ctx.call("btf_set_profiling", 1)
m._ctx.kernel_times()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    m._resample_W(Y)
m.sync(); tw = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for _ in range(n):
    m._resample_V(Y)
m.sync(); tv = (time.perf_counter() - t0) / n
kt = m._ctx.kernel_times()
print("(%d,%d,%d) K=%d J=%d ngrid=%d: W step %.2f ms, V step %.2f ms; ess/gass kernels %.2f ms per W+V" % (
    N, M, T, K, Cons.shape[0], ngrid, 1e3 * tw, 1e3 * tv, kt["ess"][0] / n))
tau = np.einsum("nk,mtk->nmt", m.W, m.V)
print("feasible:", bool((np.einsum("qt,nmt->nmq", Cons[:, :-1], tau) >= Cons[:, -1] - 1e-9).all()), "ll", m.log_likelihood(Y))
