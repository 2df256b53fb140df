"""ctypes binding of the C ABI in include/btf.h (libbtf_hip.so, built in-tree).

There is no CPU fallback: if the shared library is missing or no GPU is
visible, every compute entry point raises.  ``load()`` is lazy so that the pure
host logic (and the symbol-export test) can be imported on a CPU-only box.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.environ.get("BTF_LIB_PATH") or os.path.join(HERE, "libbtf_hip.so")   # override: A/B builds
SOURCES = [os.path.join(CSRC, "btf_abi.hip")]
HEADERS = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + \
    [os.path.join(ROOT, "include", "btf.h")]

BTF_OK, BTF_EINVAL, BTF_EHIP, BTF_ENOTPD, BTF_ESTATE = 0, 1, 2, 3, 4
COMPAT = {"reference": 0, "exact": 1}
KERNEL_NAMES = ["stats", "w_accum", "w_solve", "v_accum", "v_banded", "gram", "products", "sse", "pg_draw", "nb_loglik",
                "prior_band", "gram_eig", "hyper", "ess"]
COMM_ID_BYTES = 128               # BTF_COMM_ID_BYTES of include/btf.h
PEER_DESC_BYTES = 256             # BTF_PEER_DESC_BYTES
OPT_SAMPLER, OPT_NB_HISTOGRAMS, OPT_FUSE_GRAM, OPT_PG_EXACT, OPT_CURVE_COUNTS, OPT_SPLIT_ACCUM, OPT_FUSED_SWEEP, OPT_FUSED_STEP, OPT_FUSED_DATAFLOW = 0, 1, 2, 3, 4, 5, 6, 7, 8
ESS_HOST_LIKELIHOOD = -1          # BTF_ESS_HOST_LIKELIHOOD of include/btf.h
SAMPLERS = {"banded": 0, "spectral": 1, "chain": 2, "generic": 3, "banded_nopanel": 4}

# every symbol include/btf.h declares: (name, restype, argtypes)
_c_dp = C.POINTER(C.c_double)
_c_ip = C.POINTER(C.c_int32)
_ctx = C.c_void_p
SIGNATURES = {
    "btf_create": (C.c_int, [C.POINTER(_ctx), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "btf_destroy": (None, [_ctx]),
    "btf_last_error": (C.c_char_p, [_ctx]),
    "btf_fail_index": (C.c_int, [_ctx]),
    "btf_set_shard": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, C.c_int]),
    "btf_stream": (C.c_void_p, [_ctx]),
    "btf_dev_W": (C.c_void_p, [_ctx]),
    "btf_dev_V": (C.c_void_p, [_ctx]),
    "btf_set_data_gaussian": (C.c_int, [_ctx, _c_dp, _c_dp, C.c_int]),
    "btf_set_data_binomial": (C.c_int, [_ctx, _c_dp, _c_dp, _c_dp, _c_dp]),
    "btf_set_stale_sources": (C.c_int, [_ctx, _c_ip, _c_ip]),
    "btf_set_shard_halo": (C.c_int, [_ctx, C.c_int, C.c_int]),
    "btf_set_W": (C.c_int, [_ctx, _c_dp]),
    "btf_get_W": (C.c_int, [_ctx, _c_dp]),
    "btf_set_V": (C.c_int, [_ctx, _c_dp]),
    "btf_get_V": (C.c_int, [_ctx, _c_dp]),
    "btf_set_hyper": (C.c_int, [_ctx, _c_dp, C.c_double, C.c_double]),
    "btf_set_tau_chain": (C.c_int, [_ctx, _c_dp, _c_dp, _c_dp]),
    "btf_get_tau": (C.c_int, [_ctx, _c_dp, _c_dp, _c_dp, _c_dp]),
    "btf_resample_Tau2": (C.c_int, [_ctx, C.c_uint64, C.c_double, C.c_double, _c_dp]),
    "btf_set_nu2": (C.c_int, [_ctx, C.c_double]),
    "btf_set_omega": (C.c_int, [_ctx, _c_dp, _c_dp]),
    "btf_get_omega": (C.c_int, [_ctx, _c_dp]),
    "btf_w_accum": (C.c_int, [_ctx, C.c_int]),
    "btf_resample_W": (C.c_int, [_ctx, _c_dp, C.c_uint64, C.c_int]),
    "btf_resample_V": (C.c_int, [_ctx, _c_dp, C.c_uint64, C.c_int, C.c_double, C.c_int]),
    "btf_get_V_attempts": (C.c_int, [_ctx, _c_ip]),
    "btf_get_V_order": (C.c_int, [_ctx, _c_ip]),
    "btf_sse": (C.c_int, [_ctx, _c_dp, _c_dp]),
    "btf_sse_begin": (C.c_int, [_ctx]),
    "btf_set_data_counts": (C.c_int, [_ctx, _c_dp, C.c_int]),
    "btf_nb_loglik": (C.c_int, [_ctx, _c_dp, _c_dp, _c_ip, _c_dp]),
    "btf_nb_set_rate": (C.c_int, [_ctx, _c_dp, _c_ip]),
    "btf_nb_mh": (C.c_int, [_ctx, C.c_uint64, C.c_int, C.c_double, C.c_double, _c_ip, _c_dp]),
    "btf_nb_get_rate": (C.c_int, [_ctx, _c_dp, _c_ip]),
    "btf_dev_hyp": (C.c_void_p, [_ctx]),
    "btf_set_global_nobs": (C.c_int, [_ctx, C.c_double]),
    "btf_set_scalar_slot": (C.c_int, [_ctx, C.c_int, C.c_double]),
    "btf_device_scalars": (C.c_int, [_ctx, C.c_int]),
    "btf_set_scalars": (C.c_int, [_ctx, C.c_double, C.c_double, C.c_double, C.c_double]),
    "btf_get_scalars": (C.c_int, [_ctx, _c_dp]),
    "btf_draw_scalars": (C.c_int, [_ctx, C.c_uint64, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]),
    "btf_draw_lam2": (C.c_int, [_ctx, C.c_uint64, C.c_int]),
    "btf_sse_end": (C.c_int, [_ctx, _c_dp, _c_dp, _c_dp]),
    "btf_pg_draw": (C.c_int, [_ctx, C.c_uint64]),
    "btf_pg_batch": (C.c_int, [C.c_int, C.c_int64, _c_dp, _c_dp, C.c_uint64, _c_dp]),
    "btf_pg_batch_mode": (C.c_int, [C.c_int, C.c_int64, _c_dp, _c_dp, C.c_uint64, C.c_int, _c_dp]),
    "btf_posterior_summary": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _c_dp, _c_dp, C.c_int, _c_dp,
                                        C.c_int, _c_dp, _c_dp]),
    "btf_collect_begin": (C.c_int, [_ctx, C.c_int]),
    "btf_collect": (C.c_int, [_ctx, C.c_int]),
    "btf_collect_schedule": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int]),
    "btf_collect_end": (C.c_int, [_ctx, C.c_int, _c_dp, _c_dp, _c_dp, _c_dp]),
    "btf_collect_summary": (C.c_int, [_ctx, C.c_int, C.c_int, _c_dp, C.c_int, _c_dp, _c_dp]),
    "btf_sync": (C.c_int, [_ctx]),
    "btf_ess_begin": (C.c_int, [_ctx, C.c_int, _c_dp, C.c_uint64, C.c_double, C.c_int]),
    "btf_ess_eval": (C.c_int, [_ctx, C.c_int, C.c_double, C.c_int, C.c_int, _c_dp]),
    "btf_ess_run": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, _c_dp, C.c_uint64, C.c_int, C.c_double, C.c_int]),
    "btf_ess_info": (C.c_int, [_ctx, _c_ip, _c_dp]),
    "btf_mvn_banded": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _c_dp, _c_dp, _c_dp, C.c_uint64,
                                 C.c_double, C.c_int, _c_dp, _c_ip]),
    "btf_set_profiling": (C.c_int, [_ctx, C.c_int]),
    "btf_kernel_times": (C.c_int, [_ctx, _c_dp, C.POINTER(C.c_int64)]),
    "btf_set_tuning": (C.c_int, [_ctx, C.c_int, C.c_int]),
    "btf_set_option": (C.c_int, [_ctx, C.c_int, C.c_int]),
    "btf_sym_eig": (C.c_int, [C.c_int, C.c_int, C.c_int, _c_dp, _c_dp, _c_dp]),
    "btf_read_probe": (C.c_int, [C.c_int, C.c_size_t, C.c_int, _c_dp]),
    "btf_queue_Tau2": (C.c_int, [_ctx, C.c_uint64, C.c_double]),
    "btf_queue_lam2": (C.c_int, [_ctx, C.c_uint64, C.c_int]),
    "btf_gibbs_sweeps": (C.c_int, [_ctx, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_double, C.c_double, C.c_double,
                                   C.c_double, C.c_double, C.c_double, C.c_int]),
    "btf_wv_steps": (C.c_int, [_ctx, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_double, C.c_int]),
    "btf_gass_set_constraints": (C.c_int, [_ctx, _c_dp, C.c_int, _c_dp, C.c_int]),
    "btf_gass_begin": (C.c_int, [_ctx, C.c_int, C.c_int, _c_dp, _c_dp, C.c_uint64, C.c_double, C.c_int, C.c_int]),
    "btf_gass_grid": (C.c_int, [_ctx, C.c_int, _c_ip, C.POINTER(C.c_uint8), _c_dp, _c_dp]),
    "btf_gass_eval": (C.c_int, [_ctx, C.c_int, _c_dp, _c_ip, _c_dp]),
    "btf_gass_commit": (C.c_int, [_ctx, C.c_int, _c_dp, _c_ip]),
    "btf_gass_select": (C.c_int, [_ctx, C.c_int, C.c_uint64, _c_ip]),
    "btf_gass_run": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_double, C.c_int]),
    "btf_mvn_dense": (C.c_int, [C.c_int, C.c_int, C.c_int, _c_dp, C.c_int, _c_dp, _c_dp, _c_dp, C.c_uint64, C.c_double, C.c_int,
                                _c_dp, _c_ip]),
    "btf_get_likelihood_form": (C.c_int, [_ctx, _c_ip]),
    "btf_get_draw_counters": (C.c_int, [_ctx, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "btf_set_draw_counters": (C.c_int, [_ctx, C.c_uint64, C.c_uint64]),
    "btf_get_accum_bytes_per_cell": (C.c_int, [_ctx, _c_dp]),
    "btf_get_V_sampler": (C.c_int, [_ctx, _c_ip]),
    "btf_queue_scalars": (C.c_int, [_ctx, C.c_uint64, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, _c_ip]),
    "btf_host_selftest": (C.c_int, []),
    "btf_set_likelihood_param": (C.c_int, [_ctx, C.c_int, C.c_double]),
    "btf_comm_fork": (C.c_int, [_ctx, C.c_void_p]),
    "btf_comm_join": (C.c_int, [_ctx, C.c_void_p]),
    "btf_comm_unique_id": (C.c_int, [C.POINTER(C.c_ubyte), C.c_int]),
    "btf_comm_block": (C.c_int, [C.c_int, C.c_int, C.c_int, _c_ip, _c_ip]),
    "btf_comm_init": (C.c_int, [_ctx, C.c_int, C.c_int, C.POINTER(C.c_ubyte), C.c_int]),
    "btf_comm_rehearse": (C.c_int, [_ctx, C.c_int, C.c_int]),
    "btf_peer_export": (C.c_int, [_ctx, C.POINTER(C.c_ubyte), C.c_int]),
    "btf_peer_init": (C.c_int, [_ctx, C.c_int, C.c_int, C.POINTER(C.c_ubyte), C.c_int]),
    "btf_comm_destroy": (C.c_int, [_ctx]),
    "btf_comm_info": (C.c_int, [_ctx, _c_ip]),
    "btf_allgather_W": (C.c_int, [_ctx]),
    "btf_allgather_V": (C.c_int, [_ctx]),
    "btf_allreduce_sse": (C.c_int, [_ctx]),
    "btf_allreduce_sum": (C.c_int, [_ctx, _c_dp, C.c_int]),
    "btf_set_gathered_W": (C.c_int, [_ctx, _c_dp]),
    "btf_set_gathered_V": (C.c_int, [_ctx, _c_dp]),
}


class BTFError(RuntimeError):
    def __init__(self, code, msg, index=-1):
        super().__init__("btf error %d: %s" % (code, msg))
        self.code = code
        self.index = index


class NotPositiveDefiniteError(BTFError, np.linalg.LinAlgError):
    """Raised where the reference raises LinAlgError (W step, factor.py:357) and
    where it would loop forever after the jitter retries (fast_mvn.py:69-72)."""


INST_SOURCE = os.path.join(CSRC, "btf_instances.hip")
INST_PARTS = 9                      # = BTF_INST_PARTS of csrc/btf_instances.h
OBJ_DIR = os.path.join(ROOT, "build", "obj")          # git- and gpurun-ignored


def build(force=False, verbose=False, jobs=None):
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU): the C-ABI unit and the
    parts of btf_instances.hip (the large kernel families, one compilation each) in parallel, then one link."""
    deps = SOURCES + [INST_SOURCE] + HEADERS
    if not force and os.path.exists(LIB_PATH) and \
            os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(p) for p in deps):
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ_DIR, exist_ok=True)
    tag = os.path.splitext(os.path.basename(LIB_PATH))[0]
    # -amdgpu-kernarg-preload-count: gfx950's command processor hands the first 16 dwords of the kernel arguments over in
    # SGPRs - the accumulation waves issue their first loads without a scalar round trip for the pointers (10.98 -> 10.63 us
    # and 10.66 -> 10.33 us per launch at C3; the code object keeps the s_load prologue for firmware without the feature)
    base = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function",
            "-mllvm", "-amdgpu-kernarg-preload-count=16",
            "-I", os.path.join(ROOT, "include")] + os.environ.get("BTF_BUILD_DEFS", "").split()   # A/B builds: -DBTF_... tuning macros
    units = [(SOURCES[0], os.path.join(OBJ_DIR, tag + "_abi.o"), [])]
    units += [(INST_SOURCE, os.path.join(OBJ_DIR, "%s_inst%d.o" % (tag, p)), ["-DBTF_INST_PART=%d" % p])
              for p in range(INST_PARTS)]

    def compile_unit(u):
        src, obj, defs = u
        cmd = base + defs + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True, cwd=CSRC)
        return obj

    jobs = jobs or max(1, min(len(units), os.cpu_count() or 1))
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        objs = list(pool.map(compile_unit, units))
    cmd = ["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB_PATH] + objs + os.environ.get("BTF_LINK_FLAGS", "").split()
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB_PATH


_lib = None


def load():
    """Load libbtf_hip.so and declare every prototype.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("functionalmf_amd: %s is missing - run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)" % LIB_PATH)
    # One HIP runtime per process: torch bundles its own libamdhip64 (same SONAME).  If
    # torch is (or will be) used in this process it has to be imported first so that
    # our library binds to the runtime torch initialised.
    if "torch" not in sys.modules and os.environ.get("BTF_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def dptr(a):
    return a.ctypes.data_as(_c_dp) if a is not None else None


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Context:
    """Thin RAII wrapper of btf_ctx."""

    def __init__(self, nrows, ncols, ndepth, nembeds, tf_order, device=0, stream=None):
        self.lib = load()
        self.h = _ctx()
        rc = self.lib.btf_create(C.byref(self.h), nrows, ncols, ndepth, nembeds, tf_order, device,
                                 C.c_void_p(stream) if stream else None)
        if rc != BTF_OK:
            raise BTFError(rc, self.lib.btf_last_error(None).decode())
        self.dims = (nrows, ncols, ndepth, nembeds, tf_order)
        self.device = device
        self.stream_handle = self.lib.btf_stream(self.h)      # hipStream_t the step functions enqueue on

    def close(self):
        if getattr(self, "h", None):
            self.lib.btf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc == BTF_OK:
            return
        msg = self.lib.btf_last_error(self.h).decode()
        if rc == BTF_ENOTPD:
            raise NotPositiveDefiniteError(rc, msg, self.lib.btf_fail_index(self.h))
        raise BTFError(rc, msg)

    def call(self, name, *args):
        self.check(getattr(self.lib, name)(self.h, *args))

    def kernel_times(self):
        ms = np.zeros(len(KERNEL_NAMES))
        n = np.zeros(len(KERNEL_NAMES), dtype=np.int64)
        self.check(self.lib.btf_kernel_times(self.h, dptr(ms), n.ctypes.data_as(C.POINTER(C.c_int64))))
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(KERNEL_NAMES)}
