"""Probe: can two processes on ONE GPU share hipMalloc'ed and fine-grained memory through hipIpc handles, and see each
other's writes?  (scripts/ipc_probe.py; run on the GPU box)"""
import ctypes as C, os, sys, multiprocessing as mp

class Handle(C.Structure):
    _fields_ = [("reserved", C.c_ubyte * 64)]


def hip():
    import torch  # noqa: F401  (one HIP runtime per process)
    lib = C.CDLL("libamdhip64.so")
    lib.hipIpcOpenMemHandle.argtypes = [C.POINTER(C.c_void_p), Handle, C.c_uint]      # the handle travels BY VALUE
    lib.hipIpcGetMemHandle.argtypes = [C.POINTER(Handle), C.c_void_p]
    lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    return lib

def child(q_in, q_out):
    lib = hip()
    assert lib.hipSetDevice(0) == 0
    kind, handle = q_in.get()
    h = Handle()
    for i, b in enumerate(handle):
        h.reserved[i] = b
    ptr = C.c_void_p()
    rc = lib.hipIpcOpenMemHandle(C.byref(ptr), h, 1)
    q_out.put(("open", rc))
    if rc != 0:
        q_out.put(("read", None)); q_out.put(("wrote", None))
        return
    buf = (C.c_double * 4)()
    lib.hipMemcpy(buf, ptr, 32, 2)
    q_out.put(("read", list(buf)))
    buf[0] = 42.0
    lib.hipMemcpy(ptr, buf, 32, 1)
    lib.hipDeviceSynchronize()
    q_out.put(("wrote", 0))
    q_in.get()
    lib.hipIpcCloseMemHandle(ptr)

if __name__ == "__main__":
    mp.set_start_method("spawn")
    lib = hip()
    assert lib.hipSetDevice(0) == 0
    for kind in ("coarse", "fine"):
        ptr = C.c_void_p()
        if kind == "coarse":
            rc = lib.hipMalloc(C.byref(ptr), 4096)
        else:
            rc = lib.hipExtMallocWithFlags(C.byref(ptr), 4096, 0x1)      # hipDeviceMallocFinegrained
        print(kind, "alloc rc", rc)
        src = (C.c_double * 4)(1.0, 2.0, 3.0, 4.0)
        lib.hipMemcpy(ptr, src, 32, 1)
        h = Handle()
        rc = lib.hipIpcGetMemHandle(C.byref(h), ptr)
        print(kind, "get handle rc", rc)
        if rc != 0:
            continue
        q_in, q_out = mp.Queue(), mp.Queue()
        p = mp.Process(target=child, args=(q_in, q_out))
        p.start()
        q_in.put((kind, list(h.reserved)))
        print(kind, q_out.get(timeout=120))
        r = q_out.get(timeout=60)
        print(kind, r)
        print(kind, q_out.get(timeout=60))
        back = (C.c_double * 4)()
        lib.hipMemcpy(back, ptr, 32, 2)
        print(kind, "parent sees", list(back))
        q_in.put("done")
        p.join(60)
    print("IPC_PROBE_DONE")
