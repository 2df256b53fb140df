"""Time btf_nb_loglik at (512,256,64,4) K=5 for count distributions that exercise one path each."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functionalmf_amd.factor import NegativeBinomialBayesianTensorFiltering
from functionalmf_amd import _native
N, M, T, R, K = 512, 256, 64, 4, 5
rs = np.random.RandomState(0)
for name, gen in (("all small (y<=8)", lambda: rs.randint(0, 9, size=(N, M, T, R)).astype(float)),
                  ("all large (y in 40..200)", lambda: rs.randint(40, 200, size=(N, M, T, R)).astype(float)),
                  ("mixed 70/30", lambda: np.where(rs.rand(N, M, T, R) < 0.7, rs.randint(0, 20, size=(N, M, T, R)), rs.randint(40, 200, size=(N, M, T, R))).astype(float)),
                  ("non-integer", lambda: rs.rand(N, M, T, R) * 20)):
    for rdims in ((0, 1, 2), (1, 2), ()):
        data = gen()
        np.random.seed(1)
        m = NegativeBinomialBayesianTensorFiltering(N, M, T, nembeds=K, rdims=rdims, sigma2_init=0.5, lam2_init=0.1)
        m._bind_data(data)
        m._push_state()
        shp = m._rate_shape()
        Rv = 1 + rs.gamma(2, 1, size=shp)
        cand = Rv * 1.05
        ll = np.zeros(shp)
        flags = m._shared_flags().ctypes.data_as(_native._c_ip)
        for _ in range(3):
            m._ctx.call("btf_nb_loglik", _native.dptr(Rv), _native.dptr(cand), flags, _native.dptr(ll))
        m._ctx.call("btf_set_profiling", 1)
        m._ctx.kernel_times()
        t0 = time.perf_counter()
        for _ in range(10):
            m._ctx.call("btf_nb_loglik", _native.dptr(Rv), _native.dptr(cand), flags, _native.dptr(ll))
        wall = (time.perf_counter() - t0) / 10
        kt = m._ctx.kernel_times()
        print("%-26s rdims=%-9s kernel %.0f us, reduce %.0f us, call wall %.0f us" % (
            name, rdims, 1e3 * kt["nb_loglik"][0] / kt["nb_loglik"][1], 1e3 * kt["products"][0] / max(kt["products"][1], 1), 1e6 * wall), flush=True)
        del m
