"""Per-kernel breakdown of full Gibbs sweeps at C3 (nu2, sigma2, Tau2 chain, lam2, W, V by btf_gibbs_sweeps): sweeps/s and the
event time of every launch kind.  BTF_LAM_IN_WSOLVE / BTF_BAND_IN_WSOLVE = 0 give the forms this round replaced."""
import sys, os, numpy as np, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from functionalmf_amd.factor import GaussianBayesianTensorFiltering
N, M, T, R, K = 512, 256, 64, 4, 5
Y, Wt, Vt = bench.synth_legacy(1, N, M, T, R, K)
np.random.seed(1)
m = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, sigma2_init=0.5, lam2_init=0.1, nu2_init=1.0, rng="device", compat="reference", device_seed=1)
for _ in range(10):
    m.resample(Y)
m.sync()
import torch
m.resample_sweeps(Y, 50); m.sync()
torch.cuda.synchronize(); t0 = time.perf_counter()
m.resample_sweeps(Y, 400); m.sync(); torch.cuda.synchronize()
print("full sweeps/s %.0f  us/sweep %.2f" % (400 / (time.perf_counter() - t0), 1e6 * (time.perf_counter() - t0) / 400))
m._ctx.call("btf_set_profiling", 1)
m._ctx.kernel_times()
m.resample_sweeps(Y, 200); m.sync()
kt = m._ctx.kernel_times()
print({k: (round(1e3 * v[0] / v[1], 2), v[1]) for k, v in kt.items() if v[1]})
