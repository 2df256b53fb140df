"""A few rng="host" sweeps of the constrained model (host-driven GASS with default per-chain streams): must stay feasible."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functionalmf_amd.factor import ConstrainedNonconjugateBayesianTensorFiltering
rs = np.random.RandomState(3)
N, M, T, K = 9, 6, 8, 3
Wt = rs.gamma(2.0, 0.5, size=(N, K)); Wt[np.triu_indices(K, 1)] = 0
Vt = np.maximum.accumulate(rs.gamma(2.0, 0.5, size=(M, T, K))[:, ::-1], axis=1)[:, ::-1]
Y = rs.poisson(np.einsum("nk,mtk->nmt", Wt, Vt)).astype(float)
Cons = np.concatenate([np.eye(T), np.zeros((T, 1))], axis=1)
np.random.seed(1)
for link in ("poisson_identity", "poisson_log"):
    m = ConstrainedNonconjugateBayesianTensorFiltering(N, M, T, link, Cons, gass_ngrid=30, nembeds=K, tf_order=1, sigma2_init=1.0,
                                                       lam2_init=0.5, W_init=Wt, V_init=Vt)
    ll0 = m.log_likelihood(Y)
    for _ in range(5):
        m.resample(Y)
    tau = np.einsum("nk,mtk->nmt", m.W, m.V)
    print(link, "feasible", bool((tau >= -1e-9).all()), "accepted per chain (last V step)", m.gass_info["accepted"][:6], "ll", ll0, "->", m.log_likelihood(Y))
    assert (tau >= -1e-9).all()
print("HOST_MODE_OK")
