"""Loader for the C-ABI shared library (include/blockcg_hip.h).

There is no CPU fallback: if libblockcg_hip.so is missing this raises, and on a machine without a
gfx950 device bcg_context_create returns BCG_ERR_NO_DEVICE (surfaced as BlockCGError).
"""
import ctypes
import os
import subprocess
import sys

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
    "bcg_sbcgrq_plan_bytes": (ctypes.c_int, [ctypes.c_int, c_int_p, c_int_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_int, c_size_p]),
    "bcg_field_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "bcg_field_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "bcg_field_width": (ctypes.c_int, [ctypes.c_void_p]),
    "bcg_field_upload": (ctypes.c_int, [ctypes.c_void_p, c_dbl_p]),
    "bcg_field_download": (ctypes.c_int, [ctypes.c_void_p, c_dbl_p]),
    "bcg_sbcgrq_device_bytes_half": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_size_p]),
    "bcg_field_create_half": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "bcg_field_parity": (ctypes.c_int, [ctypes.c_void_p]),
    "bcg_field_sites": (ctypes.c_int64, [ctypes.c_void_p]),
    "bcg_field_parity_copy": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    "bcg_dirac_hop_half": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "bcg_host_alloc": (ctypes.c_int, [ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]),
    "bcg_host_free": (ctypes.c_int, [ctypes.c_void_p]),
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


HIP_RUNTIME = None  # which HIP runtime this process ended up with: "torch" | "torch-bundle" | "system"


def single_hip_runtime(extra=()):
    """One ROCr per process.  This image's torch wheel bundles its own libamdhip64.so / libhsa-runtime64.so / librccl.so
    (unversioned SONAMEs) beside the system ROCm this library links (libamdhip64.so.7); whichever of two ROCr instances
    initialises second sees "No HIP GPUs are available".  torch loads its bundle RTLD_GLOBAL, so with torch imported FIRST
    the library's HIP symbols bind to the bundle and there is one runtime.  A host that imports blockcg_amd first would arm
    the trap for a later `import torch`; so, when torch is installed but not imported yet, its bundled runtime is mapped
    here RTLD_GLOBAL (torch itself is NOT imported) and both end up on the same instance, in either import order.
    BCG_HIP_RUNTIME=system opts out (a host that never imports torch and wants /opt/rocm's runtime).  If a system HIP
    runtime is already mapped by something else, it is too late to unify: warn, naming the fix."""
    global HIP_RUNTIME
    import importlib.util
    import warnings
    if "torch" in sys.modules:
        HIP_RUNTIME = "torch"
        return HIP_RUNTIME
    if HIP_RUNTIME == "system" and not extra:
        return HIP_RUNTIME
    if os.environ.get("BCG_HIP_RUNTIME", "") == "system":
        HIP_RUNTIME = "system"
        return HIP_RUNTIME
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    tlib = os.path.join(os.path.dirname(spec.origin), "lib") if spec is not None and spec.origin else None
    if tlib is None or not os.path.exists(os.path.join(tlib, "libamdhip64.so")):
        HIP_RUNTIME = "system"
        return HIP_RUNTIME
    try:
        with open("/proc/self/maps") as f:
            mapped = {ln.split()[-1] for ln in f if "libamdhip64" in ln or "libhsa-runtime64" in ln}
    except OSError:
        mapped = set()
    foreign = sorted(p for p in mapped if not p.startswith(tlib))
    if foreign and HIP_RUNTIME != "torch-bundle":
        warnings.warn(f"blockcg_amd: a HIP runtime is already mapped from {foreign[0]} and torch (with its own bundled ROCm "
                      f"runtime in {tlib}) has not been imported: a later `import torch` in this process will see no GPU. "
                      "Import torch before anything that loads libamdhip64, or set BCG_HIP_RUNTIME=system if torch is never "
                      "used here.", RuntimeWarning, stacklevel=3)
        HIP_RUNTIME = "system"
        return HIP_RUNTIME
    for name in ("libhsa-runtime64.so", "libamdhip64.so") + tuple(extra):
        path = os.path.join(tlib, name)
        if os.path.exists(path):
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    HIP_RUNTIME = "torch-bundle"
    return HIP_RUNTIME


def load():
    global _lib
    if _lib is not None:
        return _lib
    single_hip_runtime()
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
