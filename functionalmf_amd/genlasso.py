"""MCMC driver and the conjugate inverse-gamma update: counterpart of
functionalmf/genlasso.py:1-66 and :139-171 (same class and method names)."""
import numpy as np


class _BayesianModel(object):
    def __init__(self, **kwargs):
        # the reference swallows unknown kwargs here (genlasso.py:8), e.g. nthreads=1
        pass

    def resample(self, data, **kwargs):
        raise NotImplementedError

    def _inferred_variables(self, var_map):
        raise NotImplementedError

    def inferred_variables(self):
        """All non-nuisance parameters inferred by calling resample."""
        out = {}
        self._inferred_variables(out)
        return out

    def run_gibbs(self, data, nburn=1000, nthin=1, nsamples=1000, verbose=True, print_freq=100,
                  callback=None, **kwargs):
        """Burn in, then keep every `nthin`-th state.  Result layout as the reference
        (genlasso.py:51-65): dict of arrays [nsamples]+shape; scalars stored as [nsamples,1]."""
        results = None
        for step in range(nburn + nthin * nsamples):
            if verbose and step % print_freq == 0:
                print('\tStep {}'.format(step))
            self.resample(data, **kwargs)
            if callback is not None:
                callback(self, data, step, **kwargs)
            kept, rem = divmod(step - nburn, nthin)
            if step >= nburn and rem == 0:
                state = self.inferred_variables()
                if kept == 0:
                    results = {k: np.zeros([nsamples] + ([1] if np.isscalar(v) else list(v.shape)))
                               for k, v in state.items()}
                for k, v in state.items():
                    results[k][kept] = v
        return results


class ConjugateInverseGammaPrior(object):
    """Gamma(shape, rate) prior on a shared precision of Gaussian observations."""

    def __init__(self, N, shape=0.1, rate=0.1):
        self.N = N
        self.shape = shape
        self.rate = rate

    def resample_from_stats(self, sqerr, nobs):
        """Posterior precision draw given the two sufficient statistics
        (genlasso.py:157-164): one legacy-RNG gamma, scale parameterisation."""
        prec = np.random.gamma(self.shape + nobs / 2, 1 / (self.rate + sqerr / 2))
        return prec if self.N == 1 else np.full(self.N, prec)

    def resample(self, data, **kwargs):
        means, obs = data
        means = np.atleast_1d(means)
        obs = np.atleast_1d(obs)
        seen = ~np.isnan(obs)
        return self.resample_from_stats(np.nansum((means - obs) ** 2), np.sum(seen))

    def draw_from_prior(self, size=1):
        return np.random.gamma(self.shape, 1 / self.rate, size=size)
