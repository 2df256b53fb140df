#!/usr/bin/env python3
"""Binomial functional matrix through Polya-Gamma augmentation on the MI355X core:
(11,12,20) cells with 10 trials each, a 3x3 block of curves held out.  Same calls as a
functionalmf user's script; only the import root differs.  No plotting."""
import sys
import os
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functionalmf_amd.factor import BinomialBayesianTensorFiltering   # was: functionalmf.factor
from functionalmf_amd.utils import ilogit, posterior_summary


def main(seed=1, nburn=2000, nthin=2, nsamples=500):
    nrows, ncols, ndepth, nembeds, ntrials = 11, 12, 20, 3, 10
    rs = np.random.RandomState(seed)
    np.random.seed(seed)
    W_true = rs.normal(size=(nrows, nembeds))
    W_true[np.triu_indices(nembeds, k=1)] = 0
    V_true = 0.25 * np.cumsum(rs.normal(size=(ncols, ndepth, nembeds)), axis=1)
    P = ilogit(np.einsum('nk,mtk->nmt', W_true, V_true))
    N = np.full((nrows, ncols, ndepth), float(ntrials))
    Y = rs.binomial(ntrials, P).astype(float)
    Y_missing, N_missing = Y.copy(), N.copy()
    Y_missing[:3, :3] = np.nan
    N_missing[np.isnan(Y_missing)] = np.nan

    model = BinomialBayesianTensorFiltering(nrows, ncols, ndepth, nembeds=nembeds, tf_order=2,
                                            sigma2_init=0.5, nthreads=1, lam2_init=0.1)
    results = model.run_gibbs((Y_missing, N_missing), nburn=nburn, nthin=nthin, nsamples=nsamples, verbose=False)
    P_hat, _ = posterior_summary(results['W'], results['V'], q=(), transform="ilogit")   # mean of ilogit(W V') over samples
    held = np.isnan(Y_missing)
    out = dict(mae_observed=float(np.abs(P_hat - P)[~held].mean()), mae_heldout=float(np.abs(P_hat - P)[held].mean()),
               corr=float(np.corrcoef(P_hat.reshape(-1), P.reshape(-1))[0, 1]))
    print("Binomial BTF (11,12,20) K=3 on MI355X:", out)
    return out


if __name__ == '__main__':
    main(seed=int(sys.argv[1]) if len(sys.argv) > 1 else 1)
