"""1-GPU rehearsal of the sharded path: 1-rank RCCL group, collectives issued for real
(BTF_EXERCISE_EXCHANGE=1), result compared with the oracle."""
import os, sys
os.environ["BTF_EXERCISE_EXCHANGE"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from functionalmf_amd.factor import GaussianBayesianTensorFiltering
from oracle import btf_oracle as orc
N, M, T, R, K = 40, 12, 10, 2, 4
rs = np.random.RandomState(0)
Y = rs.normal(size=(N, M, T, R))
np.random.seed(1)
m = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, sigma2_init=0.5, lam2_init=0.1, nu2_init=1.0,
                                    shard=(0, 1), compat="exact")
assert m._exchange.active
st = dict(W=m.W.copy(), V=m.V.copy(), Tau2=m.Tau2.copy(), lam2=0.1, sigma2=0.5, nu2=1.0)
D = orc.trend_penalty(T, 2)
rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
worst = 0.0
for it in range(3):
    np.random.seed(10 + it); m._resample_W(Y); Wg = m.W.copy()
    np.random.seed(10 + it); orc.w_step(st, Y)
    ew = rel(Wg, st["W"]); m.W = st["W"]
    np.random.seed(20 + it); m._resample_V(Y); Vg = m.V.copy()
    np.random.seed(20 + it); orc.v_step(st, Y, D, compat="exact", perm=orc.perm_from_order(m.v_order(), K, T))
    ev = rel(Vg, st["V"]); m.V = st["V"]
    print("sweep %d: rel err W %.2e V %.2e" % (it, ew, ev), flush=True)
    worst = max(worst, ew, ev * 1e-3)   # V: cond-limited (prior-drawn Tau2), 1e-6 allowed
np.random.seed(3); m._resample_nu2(Y)
sse, nobs = orc.sse_and_count(st, Y)
np.random.seed(3); ref = 1.0 / np.random.gamma(0.1 + nobs / 2.0, 1.0 / (0.1 + sse / 2.0))
print("nu2 rel err %.2e" % (abs(m.nu2 - ref) / ref))
assert worst < 1e-9 and abs(m.nu2 - ref) / ref < 1e-10
dist.destroy_process_group()
print("EXCHANGE_OK")
