#!/usr/bin/env python3
"""Negative-Binomial functional matrix on the MI355X core: (11,12,20) count curves with one
dispersion R per row (rdims=(1,2), as the functionalmf example of the same name), a 3x3 block of
curves held out.  Same calls as a functionalmf user's script; only the import root differs."""
import sys
import os
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functionalmf_amd.factor import NegativeBinomialBayesianTensorFiltering   # was: functionalmf.factor
from functionalmf_amd.utils import ilogit


def main(seed=1, nburn=1500, nthin=2, nsamples=400):
    nrows, ncols, ndepth, nembeds, nreps = 11, 12, 20, 3, 2
    rs = np.random.RandomState(seed)
    np.random.seed(seed)
    W_true = 0.8 * rs.normal(size=(nrows, nembeds))
    W_true[np.triu_indices(nembeds, k=1)] = 0
    V_true = 0.25 * np.cumsum(rs.normal(size=(ncols, ndepth, nembeds)), axis=1)
    P = ilogit(np.einsum('nk,mtk->nmt', W_true, V_true))
    R_true = 2 + 6 * rs.rand(nrows, 1, 1)
    Mu = R_true * P / (1 - P)
    Y = rs.poisson(rs.gamma(np.repeat(R_true[..., None], nreps, -1) * np.ones(P.shape + (nreps,)),
                            scale=(P / (1 - P))[..., None])).astype(float)
    Y_missing = Y.copy()
    Y_missing[:3, :3] = np.nan

    model = NegativeBinomialBayesianTensorFiltering(nrows, ncols, ndepth, nembeds=nembeds, tf_order=2,
                                                    sigma2_init=0.5, nthreads=1, lam2_init=0.1, rdims=(1, 2))
    results = model.run_gibbs(Y_missing, nburn=nburn, nthin=nthin, nsamples=nsamples, verbose=False)
    Ps = ilogit(np.einsum('znk,zmtk->znmt', results['W'], results['V']).clip(-10, 10))
    Mu_hat = (results['R'] * Ps / (1 - Ps)).mean(0)
    held = np.isnan(Y_missing[..., 0])
    out = dict(corr_observed=float(np.corrcoef(Mu_hat[~held], Mu[~held])[0, 1]),
               rel_mae_observed=float((np.abs(Mu_hat - Mu)[~held] / (1 + Mu[~held])).mean()),
               R_corr=float(np.corrcoef(results['R'].mean(0).reshape(-1), R_true.reshape(-1))[0, 1]))
    print("Negative-Binomial BTF (11,12,20) K=3, per-row dispersion, on MI355X:", out)
    return out


if __name__ == '__main__':
    main(seed=int(sys.argv[1]) if len(sys.argv) > 1 else 1)
