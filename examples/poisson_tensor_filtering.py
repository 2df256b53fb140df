#!/usr/bin/env python3
"""Poisson functional matrix factorisation by elliptical slice sampling on the MI355X core: the non-conjugate model
of the reference (NonconjugateBayesianTensorFiltering, factor.py:567-612) with the Poisson likelihood of
examples/poisson_tensor_filtering.py:26-37 - here with the log link, so that no positivity constraints are needed.

The reference passes a Python callback and evaluates it on the host for every proposal; this build evaluates the
likelihood on the GPU, so the likelihood argument is the NAME of a device likelihood.  No plotting."""
import sys
import os
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functionalmf_amd.factor import NonconjugateBayesianTensorFiltering   # was: functionalmf.factor
from functionalmf_amd.utils import posterior_summary


def main(seed=1, nburn=1500, nsamples=500, nthin=2):
    nrows, ncols, ndepth, nreps, nembeds = 11, 12, 20, 2, 3
    rs = np.random.RandomState(seed)
    np.random.seed(seed)
    W_true = 0.5 * rs.normal(size=(nrows, nembeds))
    W_true[:, 0] = 1.0
    W_true[np.triu_indices(nembeds, k=1)] = 0
    V_true = 0.15 * np.cumsum(rs.normal(size=(ncols, ndepth, nembeds)), axis=1)
    V_true[:, :, 0] += 1.0
    log_rate = np.einsum('nk,mtk->nmt', W_true, V_true)
    Y = rs.poisson(np.repeat(np.exp(log_rate)[..., None], nreps, axis=-1)).astype(float)
    Y_missing = Y.copy()
    Y_missing[:3, :3] = np.nan                                  # hold out nine curves

    # ess="rows": one slice per row of W / per column of V, the shrink loops on the GPU (rng="device")
    model = NonconjugateBayesianTensorFiltering(nrows, ncols, ndepth, "poisson_log", nembeds=nembeds, tf_order=1,
                                                sigma2_init=0.5, lam2_init=0.1, rng="device", ess="rows", device_seed=seed)
    results = model.run_gibbs(Y_missing, nburn=nburn, nthin=nthin, nsamples=nsamples, verbose=False)
    mean, (lo, hi) = posterior_summary(results['W'], results['V'], q=(5, 95))
    held = np.zeros(log_rate.shape, dtype=bool)
    held[:3, :3] = True
    rmse_in = float(np.sqrt(np.mean((mean[~held] - log_rate[~held]) ** 2)))
    rmse_out = float(np.sqrt(np.mean((mean[held] - log_rate[held]) ** 2)))
    cover = float(np.mean((log_rate >= lo) & (log_rate <= hi)))
    print("log-rate RMSE observed %.3f held-out %.3f; 90%% band coverage %.2f; final log-likelihood %.1f"
          % (rmse_in, rmse_out, cover, model.log_likelihood(Y_missing)))
    return rmse_in, rmse_out, cover


if __name__ == "__main__":
    main()
