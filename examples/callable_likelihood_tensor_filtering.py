#!/usr/bin/env python3
"""The reference's own calling convention for the non-conjugate model: a PYTHON FUNCTION as the likelihood
(NonconjugateBayesianTensorFiltering(nrows, ncols, ndepth, loglikelihood, ...), factor.py:567-570; its caller
examples/poisson_tensor_filtering.py:60-70 passes a function of (W, V, data)).

This build evaluates named likelihoods on the GPU (examples/poisson_tensor_filtering.py here); a callable takes the
labelled slow path instead - the reference's joint elliptical slice sampler walked on the host, the function evaluated on
W / V read back for every proposal - so that user scripts with their own likelihood run unchanged.  Here: a Student-t
noise model (3 degrees of freedom) that no device likelihood covers.  Small and short on purpose (the reference's scheme -
one slice over all of W, one over all of V - mixes slowly: the fit keeps improving for tens of thousands of sweeps).
No plotting."""
import sys
import os
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functionalmf_amd.factor import NonconjugateBayesianTensorFiltering   # was: functionalmf.factor


def student_t_loglik(W, V, data, dof=3.0, scale=0.3):
    """log p(data | W, V) up to a constant: independent Student-t residuals about the low-rank mean; NaN = missing."""
    mu = np.einsum('nk,mtk->nmt', W, V)[..., None]
    r = (data - mu) / scale
    return float(np.nansum(-0.5 * (dof + 1.0) * np.log1p(r * r / dof)))


def main(seed=2, nburn=2000, nsamples=200, nthin=2):
    nrows, ncols, ndepth, nreps, nembeds = 8, 9, 12, 2, 2
    rs = np.random.RandomState(seed)
    np.random.seed(seed)
    W_true = rs.normal(size=(nrows, nembeds))
    W_true[np.triu_indices(nembeds, k=1)] = 0
    V_true = 0.3 * np.cumsum(rs.normal(size=(ncols, ndepth, nembeds)), axis=1)
    mean = np.einsum('nk,mtk->nmt', W_true, V_true)
    Y = mean[..., None] + 0.3 * rs.standard_t(3.0, size=(nrows, ncols, ndepth, nreps))
    Y[:2, :2] = np.nan                                          # hold out four curves

    model = NonconjugateBayesianTensorFiltering(nrows, ncols, ndepth, student_t_loglik, nembeds=nembeds, tf_order=1,
                                                sigma2_init=1.0, lam2_init=0.1)
    results = model.run_gibbs(Y, nburn=nburn, nthin=nthin, nsamples=nsamples, verbose=False)
    fit = np.einsum('snk,smtk->snmt', results['W'], results['V']).mean(axis=0)
    held = np.zeros(mean.shape, dtype=bool)
    held[:2, :2] = True
    rmse_in = float(np.sqrt(np.mean((fit[~held] - mean[~held]) ** 2)))
    rmse_out = float(np.sqrt(np.mean((fit[held] - mean[held]) ** 2)))
    print("Student-t callable likelihood (%d,%d,%d) K=%d: mean RMSE observed %.3f held-out %.3f; likelihood evaluations of the last "
          "slice %d; final log-likelihood %.1f" % (nrows, ncols, ndepth, nembeds, rmse_in, rmse_out, model.ess_evaluations,
                                                   student_t_loglik(model.W, model.V, Y)))
    return rmse_in, rmse_out


if __name__ == "__main__":
    main()
