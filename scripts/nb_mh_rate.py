"""Time of the Negative-Binomial rate update (30 MH steps, one shared rate) at (512,256,64,4): the one-launch loop
against the per-step launches (BTF_NB_MH_STEPWISE=1), on counts without / with values beyond the 1024-entry table."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functionalmf_amd.factor import NegativeBinomialBayesianTensorFiltering

N, M, T, R, K = 512, 256, 64, 4, 5
rs = np.random.RandomState(3)
W = 0.5 * rs.normal(size=(N, K))
V = 0.1 * np.cumsum(rs.normal(size=(M, T, K)), axis=1)
for scale, tag in ((1.0, "counts < 1024"), (4.0, "with outliers")):
    Mu = scale * np.einsum("nk,mtk->nmt", W, V)
    P = 1 / (1 + np.exp(-np.clip(Mu, -8, 8)))
    Y = rs.negative_binomial(4.0, 1 - np.repeat(P[..., None], R, axis=-1)).astype(float)
    print(tag, "max count", Y.max(), "outliers", int((Y >= 1024).sum()), flush=True)
    for mode in ("0", "1"):
        os.environ["BTF_NB_MH_STEPWISE"] = mode
        np.random.seed(1)
        m = NegativeBinomialBayesianTensorFiltering(N, M, T, nembeds=K, rng="device", device_seed=1)
        for _ in range(3):
            m.resample(Y)
        m.sync()
        t0 = time.perf_counter()
        for _ in range(50):
            m._resample_R(Y)
        m.sync()
        print("  stepwise=%s  %.1f us per rate update, R = %.4f" % (mode, (time.perf_counter() - t0) / 50 * 1e6, float(np.asarray(m.R).reshape(-1)[0])), flush=True)
