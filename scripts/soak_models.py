"""Soak of the other model families in rng="device" mode: Binomial (PG series + byte pseudo-data), Negative-Binomial,
non-conjugate Poisson (ESS per row / column) and the constrained Poisson model (GASS): everything finite, constraints
kept, throughput printed."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functionalmf_amd.factor import (BinomialBayesianTensorFiltering, NegativeBinomialBayesianTensorFiltering,
                                     NonconjugateBayesianTensorFiltering, ConstrainedNonconjugateBayesianTensorFiltering)
rs = np.random.RandomState(0)
N, M, T, K = 96, 48, 24, 4
Wt = rs.normal(size=(N, K)); Wt[np.triu_indices(K, 1)] = 0
Vt = 0.3 * np.cumsum(rs.normal(size=(M, T, K)), axis=1)
eta = np.einsum("nk,mtk->nmt", Wt, Vt)
P = 1 / (1 + np.exp(-eta))


def finite(m):
    return bool(np.isfinite(m.W).all() and np.isfinite(m.V).all() and np.isfinite(np.asarray(m.Tau2)).all())


np.random.seed(1)
Ntr = np.full((N, M, T), 6.0); Ys = rs.binomial(6, P).astype(float); Ys[:3, :3] = np.nan; Ntr[:3, :3] = np.nan
m = BinomialBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, sigma2_init=0.5, lam2_init=0.1, rng="device")
t0 = time.time()
for _ in range(2000):
    m.resample((Ys, Ntr))
m.sync(); print("binomial: 2000 sweeps %.1fs finite=%s corr=%.3f" % (time.time() - t0, finite(m), np.corrcoef(np.einsum("nk,mtk->nmt", m.W, m.V).ravel(), eta.ravel())[0, 1]), flush=True)
assert finite(m)

Yc = rs.negative_binomial(4.0, 1 - np.repeat(P[..., None], 2, axis=-1)).astype(float)
m = NegativeBinomialBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, sigma2_init=0.5, lam2_init=0.1, rng="device")
t0 = time.time()
for _ in range(300):
    m.resample(Yc)
m.sync(); print("negbinom: 300 sweeps %.1fs finite=%s R=%.3f" % (time.time() - t0, finite(m), float(np.mean(m.R))), flush=True)
assert finite(m)

Yp = rs.poisson(np.exp(0.5 * eta)).astype(float)
m = NonconjugateBayesianTensorFiltering(N, M, T, "poisson_log", nembeds=K, tf_order=1, sigma2_init=0.5, lam2_init=0.1, rng="device", ess="rows")
t0 = time.time()
for _ in range(1500):
    m.resample(Yp)
m.sync(); print("poisson ESS: 1500 sweeps %.1fs finite=%s unfinished=%d" % (time.time() - t0, finite(m), m.ess_unfinished()), flush=True)
assert finite(m)

Wp = np.abs(Wt) + 0.1; Wp[np.triu_indices(K, 1)] = 0
Vp = np.maximum.accumulate((np.abs(Vt) + 0.1)[:, ::-1], axis=1)[:, ::-1]
Yq = rs.poisson(np.einsum("nk,mtk->nmt", Wp, Vp)).astype(float)
Cons = np.concatenate([np.eye(T), np.zeros((T, 1))], axis=1)
mono = np.array([np.concatenate([np.zeros(t), [1, -1], np.zeros(T - t - 2), [-1e-2]]) for t in range(T - 1)])
Cons = np.concatenate([Cons, mono], axis=0)
m = ConstrainedNonconjugateBayesianTensorFiltering(N, M, T, "poisson_identity", Cons, nembeds=K, tf_order=0, sigma2_init=1.0, lam2_init=0.5,
                                                   W_init=Wp, V_init=Vp, rng="device")
t0 = time.time()
for it in range(600):
    m.resample(Yq)
tau = np.einsum("nk,mtk->nmt", m.W, m.V)
ok = bool((np.einsum("qt,nmt->nmq", Cons[:, :-1], tau) >= Cons[:, -1] - 1e-9).all())
print("constrained GASS: 600 sweeps %.1fs finite=%s feasible=%s" % (time.time() - t0, finite(m), ok), flush=True)
assert finite(m) and ok
print("SOAK_MODELS_OK")
