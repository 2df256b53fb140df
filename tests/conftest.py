import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def state_from(g, prefix):
    """Oracle/product state dict from fixture keys '<prefix>W', ... (scalars unboxed)."""
    st = {}
    for k in ("W", "V", "Tau2", "Tau2_a", "Tau2_b", "Tau2_c", "lam2", "lam2_a", "sigma2", "nu2"):
        v = np.array(g[prefix + k], dtype=float)
        st[k] = float(v) if v.ndim == 0 else (float(v[0]) if v.shape == (1,) and k in ("lam2", "lam2_a", "sigma2", "nu2") else v.copy())
    return st


def relerr(a, b):
    a = np.asarray(a, float)
    b = np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get
