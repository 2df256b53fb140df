"""Diagnostic: where does the fast banded kernel spend its cycles (phase stamps).
BTF_SAMPLER=banded|spectral|chain, BTF_HELDOUT=1 (whole curves held out), BTF_MISSING5=1 (weighted data)."""
import ctypes as C
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functionalmf_amd.factor import GaussianBayesianTensorFiltering
from bench import synth_rows, synth_V

from bench import CONFIGS
_cfg = CONFIGS[os.environ.get("CFG", "c3")]      # CFG=c3 (default) | c3k10 | flu | ...
N, M, T, R, K = _cfg["N"], _cfg["M"], _cfg["T"], _cfg["R"], _cfg["K"]
Vt = synth_V(1, M, T, K)
Y, _ = synth_rows(1, range(N), M, T, R, K, Vt)
if os.environ.get("BTF_HELDOUT"):
    Y[:3, :3] = np.nan
if os.environ.get("BTF_MISSING5"):
    rs = np.random.RandomState(7)
    Y[rs.rand(N, M) < 0.05] = np.nan
    Y[rs.rand(N, M, T, R) < 0.05] = np.nan
np.random.seed(1)
m = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, sigma2_init=0.5, lam2_init=0.1, nu2_init=1.0, rng="device")
import os
from functionalmf_amd import _native
m._ctx.call("btf_set_option", _native.OPT_SAMPLER, _native.SAMPLERS[os.environ.get("BTF_SAMPLER", "spectral")])
for _ in range(2):
    m.resample(Y)
lib = m._ctx.lib
lib.btf_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
out = np.zeros((M, 6), dtype=np.int64)
lib.btf_debug_stamps(m._ctx.h, out.ctypes.data_as(C.POINTER(C.c_longlong)))   # arm
for _ in range(3):
    m._resample_W(Y); m._resample_V(Y)
m.sync()
lib.btf_debug_stamps(m._ctx.h, out.ctypes.data_as(C.POINTER(C.c_longlong)))
d = np.diff(out, axis=1)
import os
print("sampler", os.environ.get("BTF_SAMPLER", "spectral"), m.v_sampler())
names = ["setup(m0,gram,P)", "assemble", "factor(+z gen)", "w init", "backward"]
print("median cycles per phase (shader clock):")
for i, nme in enumerate(names):
    print("  %-18s %8.0f   (min %d max %d)" % (nme, np.median(d[:, i]), d[:, i].min(), d[:, i].max()))
print("  total %.0f cycles; start spread %.0f" % (np.median(out[:, 5] - out[:, 0]), out[:, 0].max() - out[:, 0].min()))
