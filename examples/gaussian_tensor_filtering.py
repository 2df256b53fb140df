#!/usr/bin/env python3
"""Config C1 of BASELINE.json on the MI355X core: a (10,11,12,3) Gaussian functional matrix,
nembeds=3, 3x3 block of curves held out, full Gibbs sampler, posterior mean vs truth.

The model/driver calls are the ones a functionalmf user writes (README.md:15-40 of the
reference); only the import root differs.  No plotting."""
import sys
import os
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functionalmf_amd.factor import GaussianBayesianTensorFiltering   # was: functionalmf.factor
from functionalmf_amd.utils import posterior_summary


def smooth_truth(nrows, ncols, ndepth, nembeds, rs):
    """Lower-triangular-headed loadings and random-walk curves (SURVEY 8d generator)."""
    W = rs.normal(size=(nrows, nembeds))
    W[np.triu_indices(nembeds, k=1)] = 0
    V = 0.3 * np.cumsum(rs.normal(size=(ncols, ndepth, nembeds)), axis=1)
    return W, V


def main(seed=1, nburn=1000, nsamples=1000, rng="host"):
    nrows, ncols, ndepth, nreps, nembeds = 10, 11, 12, 3, 3
    rs = np.random.RandomState(seed)
    np.random.seed(seed)
    W_true, V_true = smooth_truth(nrows, ncols, ndepth, nembeds, rs)
    Mu = np.einsum('nk,mtk->nmt', W_true, V_true)
    Y = Mu[..., None] + rs.normal(0, 0.5, size=(nrows, ncols, ndepth, nreps))
    Y_missing = Y.copy()
    Y_missing[:3, :3] = np.nan                                  # hold out nine curves

    model = GaussianBayesianTensorFiltering(nrows, ncols, ndepth, nembeds=nembeds, tf_order=2,
                                            sigma2_init=0.5, nthreads=1, lam2_init=0.1, nu2_init=1, rng=rng)
    results = model.run_gibbs(Y_missing, nburn=nburn, nthin=1, nsamples=nsamples, print_freq=100, verbose=False)
    # the reference script builds the (S,N,M,T) tensor on the host: einsum + mean + two np.percentile calls
    # (examples/gaussian_tensor_filtering.py:82-85); same numbers from the GPU without that tensor:
    mean, (lo, hi) = posterior_summary(results['W'], results['V'], q=(5, 95))
    held = np.zeros(Mu.shape, dtype=bool)
    held[:3, :3] = True
    out = dict(rmse_observed=float(np.sqrt(((mean - Mu)[~held] ** 2).mean())),
               rmse_heldout=float(np.sqrt(((mean - Mu)[held] ** 2).mean())),
               coverage90=float(((lo <= Mu) & (Mu <= hi)).mean()),
               nu2=float(results['nu2'].mean()))
    print("Gaussian BTF (10,11,12,3) K=3 on MI355X:", out)
    return out


if __name__ == '__main__':
    main(seed=int(sys.argv[1]) if len(sys.argv) > 1 else 1)
