"""Soak: long device-mode chains at several shapes; everything must stay finite, no failed
factorisation, W+V throughput stable."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functionalmf_amd.factor import GaussianBayesianTensorFiltering
for (N, M, T, R, K, tf, sweeps) in ((64, 32, 16, 2, 3, 2, 20000), (512, 256, 64, 4, 5, 2, 3000), (200, 100, 40, 2, 8, 2, 3000), (100, 50, 30, 1, 4, 1, 5000)):
    rs = np.random.RandomState(0)
    Vt = 0.2 * np.cumsum(rs.normal(size=(M, T, K)), axis=1); Wt = rs.normal(size=(N, K))
    Y = np.einsum("nk,mtk->nmt", Wt, Vt)[..., None] + rs.normal(0, 0.5, size=(N, M, T, R))
    if K == 4:
        Y[rs.rand(N, M, T, R) < 0.1] = np.nan
    if K == 8:
        Y[rs.rand(N, M) < 0.05] = np.nan             # whole curves missing: the curve-counts form, 200 per-column eigen-solves
    np.random.seed(1)
    m = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=tf, sigma2_init=0.5, lam2_init=0.1, nu2_init=1.0, rng="device")
    t0 = time.time()
    m.resample_sweeps(Y, sweeps)
    m.sync()
    dt = time.time() - t0
    ok = np.isfinite(m.W).all() and np.isfinite(m.V).all() and np.isfinite(m.Tau2).all()
    print("(%d,%d,%d,%d) K=%d tf=%d %s/%s: %d full sweeps in %.1fs (%.0f/s) finite=%s nu2=%.3f lam2=%.2g sigma2=%.3g" % (
        N, M, T, R, K, tf, m.likelihood_form(), m.v_sampler(), sweeps, dt, sweeps / dt, ok, m.nu2, m.lam2, m.sigma2), flush=True)
    assert ok
print("SOAK_OK")
