#!/usr/bin/env python3
"""Golden fixtures for the Negative-Binomial rate update, made by running
the REAL reference class NegativeBinomialBayesianTensorFiltering (factor.py:462-563) in the
build container with the same stand-in modules as make_golden.py (see its header).

Run:  python tests/golden/make_golden_nb.py

Writes g7_negbinom_<tag>.npz: counts, injected W / V / R, the seed, and what the reference
produced - R and the Binomial trial counts N after `_resample_R` (30 random-walk MH steps), and
the whole state after one full `resample` given injected Polya-Gamma draws.  Data only.
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def main():
    mg._install_shims()
    sys.path.insert(0, mg.REF)
    import functionalmf.factor as rfactor
    N, M, T, Rr, K = 7, 6, 8, 3, 3
    rs = np.random.RandomState(71)
    Wt = rs.normal(size=(N, K))
    Vt = 0.3 * np.cumsum(rs.normal(size=(M, T, K)), axis=1)
    P = 1 / (1 + np.exp(-np.einsum("nk,mtk->nmt", Wt, Vt)))
    data = rs.negative_binomial(4.0, 1 - P[..., None].repeat(Rr, -1)).astype(float)
    data[1, 2, 3, 0] = 57.0                      # beyond the small-count fast path of the kernel
    data[0, 0, 0, :] = [0.0, 41.0, 3.0]
    data[:2, :2] = np.nan                        # held-out curves
    data[4, 3, 5, 1] = np.nan                    # a single missing replicate
    data[5, 1, 2, :2] = np.nan
    for tag, rdims in (("scalar", (0, 1, 2)), ("rows", (1, 2)), ("cells", ()), ("cols_depth", (0,))):
        _ = mg._cfg.update(perm=mg._twist_perm_factory(K, T))
        np.random.seed(700)
        model = rfactor.NegativeBinomialBayesianTensorFiltering(
            N, M, T, nembeds=K, tf_order=2, sigma2_init=0.5, lam2_init=0.1, nthreads=1, rdims=rdims)
        model.W[:] = Wt + 0.1 * rs.normal(size=Wt.shape)
        model.W[np.triu_indices(K, k=1)] = 0
        model.V[:] = Vt + 0.1 * rs.normal(size=Vt.shape)
        g = {"dims": np.array([N, M, T, Rr, K, 2]), "rdims": np.array(rdims, dtype=int), "data": data,
             "R_before": np.array(model.R, dtype=float).copy(), "seed_R": 701}
        g.update(mg.pack("s0_", mg.snapshot(model)))
        np.random.seed(701)
        model._resample_R(data.copy())
        g["R_after"] = np.array(model.R, dtype=float).copy()
        g["N_after"] = np.array(model.N, dtype=float).copy()
        # one full sweep (R, then the Binomial sweep on (Y, N)) given injected PG draws
        omega = rs.gamma(2.0, 0.5, size=(N, M, T))
        mg._cfg["omega"] = omega
        model.R = g["R_before"].copy()
        np.random.seed(702)
        model.resample(data.copy())
        g["omega"] = omega
        g["seed_full"] = 702
        g.update(mg.pack("full_", mg.snapshot(model)))
        g["full_R"] = np.array(model.R, dtype=float).copy()
        g["full_N"] = np.array(model.N, dtype=float).copy()
        np.savez_compressed(os.path.join(HERE, "g7_negbinom_%s.npz" % tag), **g)
        print(tag, "R", g["R_before"].reshape(-1)[:3], "->", g["R_after"].reshape(-1)[:3],
              "accepted/changed:", int((g["R_after"] != g["R_before"]).sum()), "of", g["R_after"].size)


if __name__ == "__main__":
    main()
