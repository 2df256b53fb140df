import numpy as np, sys
sys.path.insert(0, '.')
from functionalmf_amd.factor import GaussianBayesianTensorFiltering
rs = np.random.RandomState(7)
N, M, T, R, K = 40, 16, 16, 3, 3
Wt = rs.normal(size=(N, K))
Vt = 0.4 * np.cumsum(rs.normal(size=(M, T, K)), axis=1)
Y = np.einsum("nk,mtk->nmt", Wt, Vt)[..., None] + rs.normal(0, 0.5, size=(N, M, T, R))
for mode, devsc in (("device", True), ("device", False), ("host", False)):
    for seed in (8, 9):
        np.random.seed(seed)
        model = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, lam2_true=0.05, sigma2_init=1.0,
                                                nu2_init=1.0, rng=mode, compat="exact", device_seed=seed)
        if not devsc and model._dev_scalars:
            model._ctx.call("btf_device_scalars", 0)
            model._dev_scalars = False
        res = model.run_gibbs(Y, nburn=500, nthin=1, nsamples=3000, verbose=False)
        nu2 = res["nu2"][:, 0]
        print(mode, devsc, seed, "nu2 mean by thirds", nu2[:1000].mean(), nu2[1000:2000].mean(), nu2[2000:].mean(),
              "sd", nu2.std(), "log sigma2", np.log(res["sigma2"][:, 0])[[0, 1000, 2000, 2999]], flush=True)
