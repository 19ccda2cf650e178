"""ctypes binding of libblockcg_rccl.so (include/blockcg_rccl.h): the bcg_comm callbacks on RCCL in native code.

bench.py uses this for the data path of N > 1 runs (halo faces over xGMI with grouped ncclSend/ncclRecv, m x m Gram
all-reduce); torch.distributed is then only the launcher's control plane (rendezvous of the 128-byte unique id).  A C++
host links the same library directly (examples/multi_gpu_solver.cpp)."""
import ctypes
import os

from . import _lib

# BCG_RCCL_LIB: another build of the transport -- the tests use libblockcg_rccl_mock.so (the same comm_rccl.cpp over a
# host-staged stand-in for RCCL) to run several ranks of the NATIVE callbacks on one GPU
LIB_PATH = os.environ.get("BCG_RCCL_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build",
                                                          "libblockcg_rccl.so")
UNIQUE_ID_BYTES = 128

SIGNATURES = {
    "bcg_rccl_get_unique_id": (ctypes.c_int, [ctypes.c_void_p]),
    "bcg_rccl_unique_id_via_file": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]),
    "bcg_rccl_unique_id_file_done": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p]),
    "bcg_comm_rccl_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                            ctypes.POINTER(ctypes.c_void_p)]),
    "bcg_comm_rccl_callbacks": (ctypes.POINTER(_lib.bcg_comm), [ctypes.c_void_p]),
    "bcg_comm_rccl_communicators": (ctypes.c_int, [ctypes.c_void_p]),
    "bcg_rccl_barrier": (ctypes.c_int, [ctypes.c_void_p]),
    "bcg_rccl_max_double": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]),
    "bcg_rccl_sum_double": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]),
    "bcg_rccl_warm_up": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "bcg_comm_rccl_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "bcg_rccl_last_error": (ctypes.c_char_p, [ctypes.c_void_p]),
}

_rlib = None


def load():
    global _rlib
    if _rlib is None:
        _lib.load()  # libblockcg_hip.so first (and the one-time build on a fresh checkout)
        if _lib.HIP_RUNTIME == "torch-bundle":  # keep RCCL on the same runtime instance as HIP (_lib.single_hip_runtime)
            _lib.single_hip_runtime(extra=("librccl.so",))
        if not os.path.exists(LIB_PATH):
            _lib.build()
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _rlib = lib
    return _rlib


def get_unique_id():
    buf = ctypes.create_string_buffer(UNIQUE_ID_BYTES)
    if load().bcg_rccl_get_unique_id(buf) != 0:
        raise RuntimeError(load().bcg_rccl_last_error(None).decode())
    return buf.raw


def unique_id_via_file(path, rank, timeout_s=120.0):
    buf = ctypes.create_string_buffer(UNIQUE_ID_BYTES)
    if load().bcg_rccl_unique_id_via_file(os.fsencode(path), rank, timeout_s, buf) != 0:
        raise RuntimeError(load().bcg_rccl_last_error(None).decode())
    return buf.raw


class RcclComm:
    """Installs the native RCCL callbacks on a blockcg_amd.Context (collective over all ranks)."""

    def __init__(self, ctx, unique_id, rank, world):
        self.lib = load()
        self.ctx = ctx
        h = ctypes.c_void_p()
        idbuf = ctypes.create_string_buffer(bytes(unique_id), UNIQUE_ID_BYTES)
        if self.lib.bcg_comm_rccl_create(ctx.h, idbuf, rank, world, ctypes.byref(h)) != 0:
            raise RuntimeError(self.lib.bcg_rccl_last_error(None).decode())
        self.h = h
        self.rank, self.world = rank, world
        self.error = None  # interface parity with comm.TorchDistComm

    @property
    def callbacks(self):
        return self.lib.bcg_comm_rccl_callbacks(self.h).contents

    @property
    def communicators(self):
        """2: the split exchange has a communicator of its own (default); 1: BCG_RCCL_SINGLE_COMM=1."""
        return self.lib.bcg_comm_rccl_communicators(self.h)

    def last_error(self):
        return self.lib.bcg_rccl_last_error(self.h).decode()

    def barrier(self):
        if self.lib.bcg_rccl_barrier(self.h) != 0:
            raise RuntimeError(self.last_error())

    def max(self, value):
        v = ctypes.c_double(value)
        if self.lib.bcg_rccl_max_double(self.h, ctypes.byref(v)) != 0:
            raise RuntimeError(self.last_error())
        return v.value

    def warm_up(self):
        """One word to and from every peer of this rank's face exchange on each communicator, then an all-reduce: RCCL's
        per-peer buffers exist afterwards, so the free device memory read next is what the solve can have (collective)."""
        from .comm import halo_plan
        msgs, _ = halo_plan(self.ctx.dims, self.ctx.grid, self.ctx.coords, 48)
        n = len(msgs)
        ps = (ctypes.c_int * max(n, 1))(*[mm[0] for mm in msgs])
        pr = (ctypes.c_int * max(n, 1))(*[mm[1] for mm in msgs])
        if self.lib.bcg_rccl_warm_up(self.h, n, ps, pr) != 0:
            raise RuntimeError(self.last_error())

    def close(self):
        if getattr(self, "h", None) and self.ctx.h:
            self.lib.bcg_comm_rccl_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
