#!/usr/bin/env python3
"""Poisson functional matrix factorisation with positivity and monotonicity constraints on the MI355X core: the
constrained non-conjugate model of the reference (ConstrainedNonconjugateBayesianTensorFiltering, factor.py:893-1010)
as examples/poisson_tensor_filtering.py sets it up - identity link, every curve w_i . v_j(t) >= 0 and, with
monotone=True, decreasing in t up to a slack of 1e-2 - updated row by row and column by column with generalized
analytic slice sampling (gass.py).

The reference evaluates a Python likelihood callback in a pool of worker processes over shared memory; here all rows
(then all columns) are updated together on the GPU, the likelihood being the device likelihood "poisson_identity".
No plotting; the reference's NMF initialisation (utils.tensor_nmf) is replaced by a crude feasible start."""
import sys
import os
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functionalmf_amd.factor import ConstrainedNonconjugateBayesianTensorFiltering   # was: functionalmf.factor
from functionalmf_amd.utils import posterior_summary


def create_piecewise_constant(rs, nrows, ncols, ndepth, ndims, break_prob=0.2):
    """Non-negative embeddings with piecewise-constant, decreasing curves (the generator of the reference example)."""
    W = rs.gamma(1, 1, size=(nrows, ndims))
    W[np.triu_indices(ndims, k=1)] = 0
    V = np.zeros((ncols, ndepth, ndims))
    for j in range(ncols):
        V[j, -1] = rs.gamma(1, 1, size=ndims)
        for k in range(ndepth - 2, -1, -1):
            V[j, k] = V[j, k + 1]
            if rs.rand() < break_prob:
                V[j, k] += rs.gamma(1, 1, size=ndims)
    return W, V


def main(seed=1, nburn=600, nsamples=300, nthin=1, monotone=True):
    nrows, ncols, ndepth, nreps, nembeds = 11, 12, 20, 1, 3
    rs = np.random.RandomState(seed)
    np.random.seed(seed)
    W_true, V_true = create_piecewise_constant(rs, nrows, ncols, ndepth, nembeds)
    rate = np.einsum('nk,mtk->nmt', W_true, V_true)
    Y = rs.poisson(np.repeat(rate[..., None], nreps, axis=-1)).astype(float)
    Y_missing = Y.copy()
    Y_missing[:3, :3] = np.nan

    Constraints = np.concatenate([np.eye(ndepth), np.zeros((ndepth, 1))], axis=1)              # positive means
    if monotone:                                                                                 # decreasing in t
        C_mono = np.array([np.concatenate([np.zeros(i), [1, -1], np.zeros(ndepth - i - 2), [-1e-2]]) for i in range(ndepth - 1)])
        Constraints = np.concatenate([Constraints, C_mono], axis=0)

    # a feasible start: constant positive rows, flat positive curves at the scale of the data
    W0 = np.full((nrows, nembeds), 1.0)
    W0[np.triu_indices(nembeds, k=1)] = 0
    V0 = np.full((ncols, ndepth, nembeds), max(np.nanmean(Y_missing), 0.1) / nembeds)
    model = ConstrainedNonconjugateBayesianTensorFiltering(nrows, ncols, ndepth, "poisson_identity", Constraints,
                                                           nembeds=nembeds, tf_order=0, sigma2_init=0.5, lam2_init=0.1,
                                                           W_init=W0, V_init=V0, rng="device", device_seed=seed)
    results = model.run_gibbs(Y_missing, nburn=nburn, nthin=nthin, nsamples=nsamples, verbose=False)
    mean, (lo, hi) = posterior_summary(results['W'], results['V'], q=(5, 95))
    A, c = Constraints[:, :-1], Constraints[:, -1]
    feasible = bool(np.all(np.einsum('qt,snmt->snmq', A, np.einsum('snk,smtk->snmt', results['W'][-20:], results['V'][-20:])) >= c - 1e-9))
    held = np.zeros(rate.shape, dtype=bool)
    held[:3, :3] = True
    rel_in = float(np.mean(np.abs(mean[~held] - rate[~held])) / np.mean(rate[~held]))
    rel_out = float(np.mean(np.abs(mean[held] - rate[held])) / np.mean(rate[held]))
    corr = float(np.corrcoef(mean.ravel(), rate.ravel())[0, 1])
    print("rate: relative MAE observed %.3f held-out %.3f, correlation %.3f; kept samples feasible: %s; final log-likelihood %.1f"
          % (rel_in, rel_out, corr, feasible, model.log_likelihood(Y_missing)))
    model.shutdown()
    return rel_in, rel_out, corr, feasible


if __name__ == "__main__":
    main()
