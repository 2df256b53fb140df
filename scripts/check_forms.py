"""Which likelihood form / sampler the golden fixtures run through (diagnostic)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, state_from
from functionalmf_amd.factor import GaussianBayesianTensorFiltering
for name in ("g1_c1_heldout.npz", "g2_c2_complete.npz", "g3_partial_reps.npz"):
    g = load_golden(name)
    N, M, T, R, K, tf = [int(x) for x in g["dims"]]
    st = state_from(g, "s0_")
    for kw in (dict(), dict(rng="device"), dict(compat="exact")):
        m = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=tf, sigma2_init=st["sigma2"], lam2_init=st["lam2"],
                                            nu2_init=st["nu2"], W_init=st["W"], V_init=st["V"], Tau2_init=st["Tau2"], **kw)
        m.set_data(g["Y"])
        print(name, kw, m.likelihood_form(), m.v_sampler())
