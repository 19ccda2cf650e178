"""Loader for the C-ABI shared library (include/blockcg_hip.h).

There is no CPU fallback: if libblockcg_hip.so is missing this raises, and on a machine without a
gfx950 device bcg_context_create returns BCG_ERR_NO_DEVICE (surfaced as BlockCGError).
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BCG_LIB") or os.path.join(_HERE, "_build", "libblockcg_hip.so")  # BCG_LIB: A/B builds

c_dbl_p = ctypes.POINTER(ctypes.c_double)
c_int_p = ctypes.POINTER(ctypes.c_int)
c_size_p = ctypes.POINTER(ctypes.c_size_t)

HALO_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, c_int_p, c_int_p, c_size_p, c_size_p, c_size_p)
ALLREDUCE_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)
HALO_END_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p)


class bcg_comm(ctypes.Structure):
    _fields_ = [("user", ctypes.c_void_p), ("halo_exchange", HALO_CB), ("allreduce_sum", ALLREDUCE_CB),
                ("halo_exchange_begin", HALO_CB), ("halo_exchange_end", HALO_END_CB)]


class bcg_sbcgrq_trace(ctypes.Structure):
    _fields_ = [("capacity", ctypes.c_int), ("recorded", ctypes.c_int), ("mats", c_dbl_p), ("res", c_dbl_p)]


# name -> (restype, argtypes); exactly the entry points declared in include/blockcg_hip.h
SIGNATURES = {
    "bcg_context_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                          c_int_p, c_int_p, c_int_p]),
    "bcg_context_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "bcg_last_error": (ctypes.c_char_p, [ctypes.c_void_p]),
    "bcg_context_set_comm": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(bcg_comm)]),
    "bcg_local_volume": (ctypes.c_int64, [ctypes.c_void_p]),
    "bcg_local_dims": (ctypes.c_int, [ctypes.c_void_p, c_int_p, c_int_p]),
    "bcg_halo_plan": (ctypes.c_int, [ctypes.c_int, c_int_p, c_int_p, c_int_p, ctypes.c_size_t, c_int_p, c_int_p, c_size_p,
                                     c_size_p, c_size_p, ctypes.POINTER(ctypes.c_int64)]),
    "bcg_halo_buffers": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                        c_size_p]),
    "bcg_synchronize": (ctypes.c_int, [ctypes.c_void_p]),
    "bcg_context_stream": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), c_int_p]),
    "bcg_overlap_tuning": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "bcg_profiling": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "bcg_profile_json": (ctypes.c_char_p, [ctypes.c_void_p]),
    "bcg_profile_reset": (ctypes.c_int, [ctypes.c_void_p]),
    "bcg_force_generic": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "bcg_capacity_mode": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "bcg_sbcgrq_device_bytes": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.POINTER(ctypes.c_size_t)]),
    "bcg_field_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "bcg_field_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "bcg_field_width": (ctypes.c_int, [ctypes.c_void_p]),
    "bcg_field_upload": (ctypes.c_int, [ctypes.c_void_p, c_dbl_p]),
    "bcg_field_download": (ctypes.c_int, [ctypes.c_void_p, c_dbl_p]),
    "bcg_field_download_sites": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64), c_dbl_p]),
    "bcg_field_copy": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "bcg_field_set_zero": (ctypes.c_int, [ctypes.c_void_p]),
    "bcg_field_fill_random": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64]),
    "bcg_field_add_assign": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "bcg_field_sub_assign": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "bcg_field_add_scalar": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double]),
    "bcg_field_add_matrix": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, c_dbl_p]),
    "bcg_field_rescale_add_scalar": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_double]),
    "bcg_field_rescale_add_matrix": (ctypes.c_int, [ctypes.c_void_p, c_dbl_p, ctypes.c_void_p, ctypes.c_double]),
    "bcg_field_hermitian_dot": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, c_dbl_p]),
    "bcg_field_real_dot": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, c_dbl_p]),
    "bcg_field_tri_solve_rhs": (ctypes.c_int, [ctypes.c_void_p, c_dbl_p]),
    "bcg_field_thin_qr": (ctypes.c_int, [ctypes.c_void_p, c_dbl_p]),
    "bcg_gauge_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]),
    "bcg_gauge_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "bcg_gauge_upload": (ctypes.c_int, [ctypes.c_void_p, c_dbl_p]),
    "bcg_gauge_fill_random": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64]),
    "bcg_dirac_hop": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "bcg_dirac_apply": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]),
    "bcg_sbcgrq_solve": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.POINTER(ctypes.c_void_p),
                                        ctypes.c_void_p, ctypes.c_int, c_dbl_p, ctypes.c_double, ctypes.c_double,
                                        ctypes.c_int, ctypes.c_int, c_int_p, c_dbl_p, ctypes.POINTER(bcg_sbcgrq_trace)]),
    "bcg_sbcgrq_begin": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.POINTER(ctypes.c_void_p),
                                        ctypes.c_void_p, ctypes.c_int, c_dbl_p, ctypes.c_double, ctypes.c_double,
                                        ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "bcg_sbcgrq_iterate": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_int_p, c_dbl_p,
                                          ctypes.POINTER(bcg_sbcgrq_trace)]),
    "bcg_sbcgrq_end": (ctypes.c_int, [ctypes.c_void_p]),
    "bcg_true_residuals": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.POINTER(ctypes.c_void_p),
                                          ctypes.c_void_p, ctypes.c_int, c_dbl_p, c_dbl_p]),
    "bcg_cg_solve": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_double, ctypes.c_int, c_int_p]),
    "bcg_scg_solve": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.POINTER(ctypes.c_void_p),
                                     ctypes.c_void_p, ctypes.c_int, c_dbl_p, ctypes.c_double, ctypes.c_double, ctypes.c_int,
                                     c_int_p]),
    "bcg_bcg_solve": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_double, ctypes.c_int, c_int_p]),
    "bcg_bcgrq_solve": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_double, ctypes.c_int, c_int_p]),
    "bcg_sbcgrq_bytes_per_iteration": (ctypes.c_double, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
}

_lib = None


def build(verbose=False):
    """Compile libblockcg_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    Serialised across processes with a file lock: under `torch.distributed.run --nproc-per-node N` on a fresh checkout
    every rank would otherwise run make at once and write the same objects.  The Makefile links to a temporary name and
    renames, so a reader never maps a half-written library."""
    import fcntl
    os.makedirs(os.path.join(_HERE, "_build"), exist_ok=True)
    with open(os.path.join(_HERE, "_build", ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4"]  # a no-op when another rank has just built it
            if not verbose:
                cmd.append("-s")
            subprocess.run(cmd, check=True)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        # a fresh checkout: compile once with hipcc (it cross-compiles gfx950 without a GPU); there is no CPU fallback
        try:
            build()
        except Exception as e:
            raise ImportError(
                f"{LIB_PATH} is missing and `make -C blockcg_amd/csrc` failed ({e}); "
                "blockcg_amd has no CPU fallback") from e
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header and library out of sync
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
