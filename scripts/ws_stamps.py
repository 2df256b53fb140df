"""Phase timeline of w_solve_kernel's wave 0 (diagnostic build: scripts/ab_build.sh wsst "-DBTF_WS_STAMPS"):
BTF_LIB_PATH=functionalmf_amd/libbtf_wsst.so python scripts/ws_stamps.py [variant]"""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from functionalmf_amd.factor import GaussianBayesianTensorFiltering
variant = sys.argv[1] if len(sys.argv) > 1 else "missing5"
N, M, T, R, K = 512, 256, 64, 4, 5
Vt = bench.synth_V(1, M, T, K)
Y, _ = bench.synth_rows(1, range(N), M, T, R, K, Vt)
if variant == "missing5":
    Y = Y.copy(); Y[np.random.RandomState(3).rand(*Y.shape) < 0.05] = np.nan
np.random.seed(1)
m = GaussianBayesianTensorFiltering(N, M, T, nembeds=K, tf_order=2, sigma2_init=0.5, lam2_init=0.1, nu2_init=1.0, rng="device", device_seed=1)
for _ in range(5):
    m.resample(Y)
m.sync()
fn = m._ctx.lib.btf_debug_stamps
fn.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
buf = np.zeros((M, 6), dtype=np.int64)
fn(m._ctx.h, buf.ctypes.data_as(C.POINTER(C.c_longlong)))          # allocates the stamp buffer
for rep in range(3):
    m._resample_W(Y)
    m.sync()
    fn(m._ctx.h, buf.ctypes.data_as(C.POINTER(C.c_longlong)))
    st = buf[:64].astype(float)
    rel = st - st[:, :1]                                            # s_memtime: shader clocks, every workgroup from its own entry
    print("rep %d  (shader clocks from the workgroup's entry; median / max over 64 workgroups)" % rep)
    for i, name in enumerate(("entry", "sums in LDS", "barrier", "wave 0 alone", "Cholesky done", "end")):
        print("  %-14s %7.0f %7.0f" % (name, np.median(rel[:, i]), rel[:, i].max()))
    m._resample_V(Y)
