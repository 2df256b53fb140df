"""Diagnostic: time of the stand-alone eigen-solve kernel (cold start) for K = 3..10, by HIP events through torch."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from functionalmf_amd import _native
lib = _native.load()
for K in (3, 5, 8, 10):
    rs = np.random.RandomState(K)
    X = rs.normal(size=(512, K))
    tril = np.tril_indices(K)
    parts = np.ascontiguousarray(np.stack([(X[b::8].T @ X[b::8])[tril] for b in range(8)]))
    out = np.zeros(K + K * K + 1)
    lib.btf_sym_eig(0, K, 8, _native.dptr(parts), _native.dptr(out), None)
    t0 = time.perf_counter()
    for _ in range(20):
        lib.btf_sym_eig(0, K, 8, _native.dptr(parts), _native.dptr(out), None)
    print("K=%d sweeps=%d  (host wall per call incl. malloc/copies: %.1f us)" % (K, out[-1], 1e6 * (time.perf_counter() - t0) / 20))
