"""End-to-end sanity run of the V samplers behind large bands: run_gibbs on noisy rank-3 data with 15 % of the replicates missing,
nembeds 8 (twisted sampler without staged likelihood blocks) and 10 (chunked chain), each against the any-size kernel on the same
chain seeds; the posterior mean must beat the cell means and the samplers must agree statistically."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functionalmf_amd.factor import GaussianBayesianTensorFiltering

for K in (8, 10):
    rs = np.random.RandomState(K)
    N, M, T, R = 60, 24, 64, 3
    Wt = rs.normal(size=(N, 3)); Vt = 0.3 * np.cumsum(rs.normal(size=(M, T, 3)), axis=1)
    mu = np.einsum("nk,mtk->nmt", Wt, Vt)
    Y = mu[..., None] + rs.normal(0, 0.5, size=(N, M, T, R))
    Y[rs.rand(N, M, T, R) < 0.15] = np.nan
    ybar = np.nanmean(Y, axis=-1)
    for sampler in ("auto", "generic"):
        np.random.seed(1)
        m = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, sigma2_init=0.5, lam2_init=0.1, nu2_init=1.0, rng="device",
                                            device_seed=4, sampler=sampler)
        res = m.run_gibbs(Y, nburn=400, nthin=1, nsamples=200, print_freq=10**9)
        fit = np.einsum("snk,smtk->snmt", res["W"], res["V"]).mean(0)
        print("K=%d sampler=%-8s rmse(fit,truth)=%.4f  rmse(cell means,truth)=%.4f  nu2=%.3f" % (
            K, m.v_sampler(), np.sqrt(np.mean((fit - mu) ** 2)), np.sqrt(np.nanmean((ybar - mu) ** 2)), float(np.mean(res["nu2"]))))
