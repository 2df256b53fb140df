"""Per-kernel HIP-event times of bare W + V steps (no hyper-parameter sweeps at all): A/B aid for builds whose side tasks
are compiled out.  BTF_SAMPLER=banded|spectral, CFG=c3|c3k8|c5."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functionalmf_amd.factor import GaussianBayesianTensorFiltering
from bench import synth_rows, synth_V, CONFIGS

cfg = CONFIGS[os.environ.get("CFG", "c3")]
N, M, T, R, K = cfg["N"], cfg["M"], cfg["T"], cfg["R"], cfg["K"]
Vt = synth_V(1, M, T, K)
Y, _ = synth_rows(1, range(N), M, T, R, K, Vt)
np.random.seed(1)
m = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, sigma2_init=0.5, lam2_init=0.1, nu2_init=1.0, rng="device",
                                    sampler=os.environ.get("BTF_SAMPLER", "banded"))
for _ in range(20):
    m._resample_W(Y); m._resample_V(Y)
m.sync()
m._ctx.call("btf_set_profiling", 1)
m._ctx.kernel_times()
for _ in range(int(os.environ.get("STEPS", "300"))):
    m._resample_W(Y); m._resample_V(Y)
m.sync()
kt = m._ctx.kernel_times()
print({k: round(1e3 * v[0] / max(v[1], 1), 2) for k, v in kt.items() if v[1] > 0})
