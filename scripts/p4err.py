import numpy as np, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from conftest import relerr
from functionalmf_amd.factor import GaussianBayesianTensorFiltering
from oracle import btf_oracle as orc
N, M, T, R, K = 512, 256, 64, 4, 5
rs = np.random.RandomState(1)
Wt = rs.normal(size=(N, K)); Wt[np.triu_indices(K, 1)] = 0
Vt = 0.1 * np.cumsum(rs.normal(size=(M, T, K)), axis=1)
Y = np.einsum("nk,mtk->nmt", Wt, Vt)[..., None] + rs.normal(0, 0.5, size=(N, M, T, R))
np.random.seed(2)
model = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, sigma2_init=0.5, lam2_init=0.1, nu2_init=1.0)
for _ in range(2):
    model.resample(Y)
Delta = orc.trend_penalty(T, 2)
Rr, ybar = orc.hoisted_stats(Y)
st0 = dict(W=model.W.copy(), V=model.V.copy(), Tau2=np.array(model.Tau2, float).copy(), lam2=float(model.lam2), sigma2=float(model.sigma2), nu2=float(model.nu2))
model._v_normals = lambda: np.zeros((M, K * T))
outs = {}
from functionalmf_amd import _native
for var in ("banded", "banded_nopanel", "chain", "generic", "spectral"):
    model._ctx.call("btf_set_option", _native.OPT_SAMPLER, _native.SAMPLERS[var])
    model.V = st0["V"].copy()
    model._resample_V(Y)
    outs[var] = model.V.copy()
st = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in st0.items()}
orc.v_step_strong(st, Rr, ybar, Delta, z=np.zeros((M, K * T)))
for var in outs:
    print("sampler %-15s vs cpu: %.3e" % (var, relerr(outs[var], st["V"])))
print("p4 vs no-p4 %.3e ; twist-p4 vs generic %.3e ; spectral vs generic %.3e ; cpu vs generic %.3e" % (
    relerr(outs["banded"], outs["banded_nopanel"]), relerr(outs["banded"], outs["generic"]), relerr(outs["spectral"], outs["generic"]),
    relerr(st["V"], outs["generic"])))
# per-column worst
e = np.abs(outs["banded"] - st["V"]).reshape(M, -1).max(1) / np.abs(st["V"]).max()
print("worst columns", np.argsort(e)[-3:], e[np.argsort(e)[-3:]], "lam2", st0["lam2"], "tau2 range", st0["Tau2"].min(), st0["Tau2"].max())
