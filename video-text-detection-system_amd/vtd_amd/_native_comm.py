"""ctypes binding of include/vtd_comm.h (libvtd_comm.so): the detections all-gather over RCCL for hosts that do not use
torch.distributed.  The Python product itself gathers through torch.distributed (vtd_amd/shard.py); this module exists so that
the C entry point is exercised by the tests and shown in INTEGRATION.md."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "libvtd_comm.so")
ID_BYTES = 128

SIGNATURES = {
    "vtd_comm_unique_id": (C.c_int, [C.c_void_p]),
    "vtd_comm_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "vtd_comm_rank": (C.c_int, [C.c_void_p]),
    "vtd_comm_world_size": (C.c_int, [C.c_void_p]),
    "vtd_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "vtd_comm_destroy": (None, [C.c_void_p]),
}

_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python __graft_entry__.py` (build) first")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with status {rc}")
