"""Python mirror of the reference's header-level interface over the C ABI.

Names and argument meaning follow the reference so that parity tests read like its own tests
(test/solvers.cpp): block_fermion_field(V-or-context, N_rhs), dirac_op(context, mass),
SBCGrQ(X, B, D, sigma, eps, eps_shifts, max_iterations) -> number of operator applications.
Host arrays are numpy complex128 in the reference's layout: field [V, m, 3]; gauge
[V, ndim, 3, 3] with [.., k, r] = U(r, k); m x m matrices as ordinary (row, col) numpy arrays.
"""
import ctypes
import json

import numpy as np

from . import _lib
from ._lib import c_dbl_p

SUPPORTED_WIDTHS = tuple(range(1, 33))  # the reference's N_rhs is any int (inc/fields.hpp:19-26)

_STATUS = {1: "BCG_ERR_INVALID", 2: "BCG_ERR_UNSUPPORTED", 3: "BCG_ERR_HIP", 4: "BCG_ERR_NO_DEVICE", 5: "BCG_ERR_COMM",
           6: "BCG_ERR_NUMERIC"}


class BlockCGError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{_STATUS.get(code, code)}: {msg}")
        self.code = code


def _dp(a):
    return a.ctypes.data_as(c_dbl_p)


def _mat_in(M, m):
    M = np.asarray(M, dtype=np.complex128)
    if M.shape != (m, m):
        raise ValueError(f"expected a {m}x{m} matrix")
    return np.ascontiguousarray(M.T)  # column-major buffer


def _ivec(v, n=4, fill=1):
    v = list(v) + [fill] * (n - len(v))
    return (ctypes.c_int * n)(*v)


class Context:
    """One GPU = one rank of the process grid over a periodic lattice of up to 4 dimensions."""

    def __init__(self, dims, device=0, grid=None, coords=None, stream=None):
        self.lib = _lib.load()
        self.dims = [int(d) for d in dims]
        self.ndim = len(self.dims)
        self.grid = [int(g) for g in grid] if grid is not None else [1] * self.ndim
        self.coords = [int(x) for x in coords] if coords is not None else [0] * self.ndim
        h = ctypes.c_void_p()
        rc = self.lib.bcg_context_create(ctypes.byref(h), device, stream, self.ndim, _ivec(self.dims), _ivec(self.grid),
                                         _ivec(self.coords, fill=0))
        if rc != 0:
            raise BlockCGError(rc, self.lib.bcg_last_error(None).decode())
        self.h = h
        self._comm_keepalive = None
        ld = (ctypes.c_int * 4)()
        og = (ctypes.c_int * 4)()
        self.lib.bcg_local_dims(self.h, ld, og)
        self.local_dims = list(ld)[:self.ndim]
        self.origin = list(og)[:self.ndim]
        self.V = int(self.lib.bcg_local_volume(self.h))

    def check(self, rc):
        if rc != 0:
            raise BlockCGError(rc, self.lib.bcg_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.bcg_context_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        self.check(self.lib.bcg_synchronize(self.h))

    def profiling(self, enable=True):
        self.check(self.lib.bcg_profiling(self.h, 1 if enable else 0))

    def profile_reset(self):
        self.check(self.lib.bcg_profile_reset(self.h))

    def profile(self):
        return json.loads(self.lib.bcg_profile_json(self.h).decode())

    def force_generic(self, enable=True):
        self.check(self.lib.bcg_force_generic(self.h, 1 if enable else 0))

    def capacity_mode(self, ring_slices):
        """Keep dirac_op::op's intermediate field as a ring of `ring_slices` x3 slices (0: whole field)."""
        self.check(self.lib.bcg_capacity_mode(self.h, int(ring_slices)))

    def sbcgrq_device_bytes(self, m, n_shifts, consume_B=False):
        n = ctypes.c_size_t()
        self.check(self.lib.bcg_sbcgrq_device_bytes(self.h, m, n_shifts, 1 if consume_B else 0, ctypes.byref(n)))
        return n.value

    def sbcgrq_device_bytes_half(self, m, n_shifts, consume_B=False):
        n = ctypes.c_size_t()
        self.check(self.lib.bcg_sbcgrq_device_bytes_half(self.h, m, n_shifts, 1 if consume_B else 0, ctypes.byref(n)))
        return n.value

    def halo_buffers(self):
        s = ctypes.c_void_p()
        r = ctypes.c_void_p()
        n = ctypes.c_size_t()
        self.check(self.lib.bcg_halo_buffers(self.h, ctypes.byref(s), ctypes.byref(r), ctypes.byref(n)))
        return s.value, r.value, n.value

    def set_comm(self, comm_struct, keepalive):
        self._comm_keepalive = (comm_struct, keepalive)
        self.check(self.lib.bcg_context_set_comm(self.h, ctypes.byref(comm_struct)))

    def bytes_per_iteration(self, m, n_shifts):
        return float(self.lib.bcg_sbcgrq_bytes_per_iteration(self.h, m, n_shifts))


class block_fermion_field:
    """Device block field (inc/fields.hpp:25-147).  `N_rhs` is the reference's template parameter."""

    def __init__(self, ctx, N_rhs, host=None, parity=None):
        """parity = 0 / 1: a half-volume field, the ctx.V / 2 sites of that parity (include/blockcg_hip.h,
        bcg_field_create_half); every operation then takes operands of the same parity."""
        self.ctx = ctx
        self.N_rhs = int(N_rhs)
        self.parity = parity
        self.V = ctx.V if parity is None else ctx.V // 2
        h = ctypes.c_void_p()
        if parity is None:
            ctx.check(ctx.lib.bcg_field_create(ctx.h, self.N_rhs, ctypes.byref(h)))
        else:
            ctx.check(ctx.lib.bcg_field_create_half(ctx.h, self.N_rhs, int(parity), ctypes.byref(h)))
        self.h = h
        if host is not None:
            self.upload(host)

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.ctx.lib.bcg_field_destroy(self.h)
            self.h = None
        except Exception:
            pass

    # full field <-> its parity halves
    def split_parity(self):
        """(even, odd): new half-volume fields holding this field's sites of parity 0 and 1."""
        out = []
        for par in (0, 1):
            half = block_fermion_field(self.ctx, self.N_rhs, parity=par)
            self.ctx.check(self.ctx.lib.bcg_field_parity_copy(self.h, half.h, 1))
            out.append(half)
        return tuple(out)

    def merge_parity(self, even, odd):
        """This (full) field's sites of either parity <- the two half-volume fields."""
        for half in (even, odd):
            self.ctx.check(self.ctx.lib.bcg_field_parity_copy(self.h, half.h, 0))
        return self

    # host <-> device
    def pinned_array(self):
        """A (V, N_rhs, 3) complex128 array in pinned host memory (bcg_host_alloc): transfers to / from it run at the bus
        rate with no staging copy.  Freed when the array (and every view of it) is garbage-collected."""
        import ctypes
        import weakref
        n = self.V * self.N_rhs * 3 * 16
        p = ctypes.c_void_p()
        self.ctx.check(self.ctx.lib.bcg_host_alloc(n, ctypes.byref(p)))
        buf = (ctypes.c_char * n).from_address(p.value)
        a = np.frombuffer(buf, dtype=np.complex128).reshape(self.V, self.N_rhs, 3)
        weakref.finalize(buf, self.ctx.lib.bcg_host_free, p)
        return a

    def upload(self, host):
        a = np.ascontiguousarray(host, dtype=np.complex128)
        if a.shape != (self.V, self.N_rhs, 3):
            raise ValueError(f"expected host array of shape {(self.V, self.N_rhs, 3)}, got {a.shape}")
        self.ctx.check(self.ctx.lib.bcg_field_upload(self.h, _dp(a)))
        return self

    def download(self, out=None):
        """out: an existing (V, N_rhs, 3) complex128 array to fill (e.g. a view of pinned memory from pinned_array)."""
        a = np.empty((self.V, self.N_rhs, 3), dtype=np.complex128) if out is None else out
        if a.shape != (self.V, self.N_rhs, 3) or a.dtype != np.complex128 or not a.flags.c_contiguous:
            raise ValueError("download(out=...): expected a C-contiguous complex128 array of shape (V, N_rhs, 3)")
        self.ctx.check(self.ctx.lib.bcg_field_download(self.h, _dp(a)))
        return a

    def download_sites(self, sites):
        """Tiles of chosen local sites, [len(sites), N_rhs, 3] (operator[] read access, inc/fields.hpp:37-38)."""
        sites = np.ascontiguousarray(sites, dtype=np.int64)
        a = np.empty((len(sites), self.N_rhs, 3), dtype=np.complex128)
        self.ctx.check(self.ctx.lib.bcg_field_download_sites(self.h, len(sites),
                                                             sites.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), _dp(a)))
        return a

    def copy(self):
        f = block_fermion_field(self.ctx, self.N_rhs)
        self.ctx.check(self.ctx.lib.bcg_field_copy(f.h, self.h))
        return f

    def setZero(self):
        self.ctx.check(self.ctx.lib.bcg_field_set_zero(self.h))
        return self

    def setRandom(self, seed=1):
        self.ctx.check(self.ctx.lib.bcg_field_fill_random(self.h, seed))
        return self

    def __iadd__(self, rhs):
        self.ctx.check(self.ctx.lib.bcg_field_add_assign(self.h, rhs.h))
        return self

    def __isub__(self, rhs):
        self.ctx.check(self.ctx.lib.bcg_field_sub_assign(self.h, rhs.h))
        return self

    def add(self, rhs, rhs_multiplier):
        """this += rhs * rhs_multiplier  (scalar or m x m)  inc/fields.hpp:70-77"""
        if np.isscalar(rhs_multiplier):
            self.ctx.check(self.ctx.lib.bcg_field_add_scalar(self.h, rhs.h, float(rhs_multiplier)))
        else:
            M = _mat_in(rhs_multiplier, self.N_rhs)
            self.ctx.check(self.ctx.lib.bcg_field_add_matrix(self.h, rhs.h, _dp(M)))
        return self

    def rescale_add(self, lhs_multiplier, rhs, rhs_multiplier):
        """this = this * lhs_multiplier + rhs * rhs_multiplier  inc/fields.hpp:79-90"""
        if np.isscalar(lhs_multiplier):
            self.ctx.check(self.ctx.lib.bcg_field_rescale_add_scalar(self.h, float(lhs_multiplier), rhs.h,
                                                                     float(rhs_multiplier)))
        else:
            M = _mat_in(lhs_multiplier, self.N_rhs)
            self.ctx.check(self.ctx.lib.bcg_field_rescale_add_matrix(self.h, _dp(M), rhs.h, float(rhs_multiplier)))
        return self

    def hermitian_dot(self, rhs):
        m = self.N_rhs
        out = np.empty((m, m), dtype=np.complex128)
        self.ctx.check(self.ctx.lib.bcg_field_hermitian_dot(self.h, rhs.h, _dp(out)))
        return np.ascontiguousarray(out.T)

    def real_dot(self, rhs):
        out = ctypes.c_double()
        self.ctx.check(self.ctx.lib.bcg_field_real_dot(self.h, rhs.h, ctypes.byref(out)))
        return out.value

    def multiply_upper_triangular_inverse_RHS(self, R):
        M = _mat_in(R, self.N_rhs)
        self.ctx.check(self.ctx.lib.bcg_field_tri_solve_rhs(self.h, _dp(M)))
        return self

    def thinQR(self):
        """In place; returns R (the reference fills its argument, inc/fields.hpp:140-146)."""
        m = self.N_rhs
        out = np.empty((m, m), dtype=np.complex128)
        self.ctx.check(self.ctx.lib.bcg_field_thin_qr(self.h, _dp(out)))
        return np.ascontiguousarray(out.T)


class dirac_op:
    """inc/dirac_op.hpp:8-44 on a device lattice: public V, mass; op(lhs, rhs)."""

    def __init__(self, ctx, mass=0.1, U=None, seed=None):
        self.ctx = ctx
        self.V = ctx.V
        self.mass = float(mass)
        h = ctypes.c_void_p()
        ctx.check(ctx.lib.bcg_gauge_create(ctx.h, ctypes.byref(h)))
        self.h = h
        if U is not None:
            self.set_links(U)
        elif seed is not None:
            ctx.check(ctx.lib.bcg_gauge_fill_random(self.h, seed))

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.ctx.lib.bcg_gauge_destroy(self.h)
            self.h = None
        except Exception:
            pass

    def set_links(self, U):
        a = np.ascontiguousarray(U, dtype=np.complex128)
        if a.shape != (self.V, self.ctx.ndim, 3, 3):
            raise ValueError(f"expected links of shape {(self.V, self.ctx.ndim, 3, 3)}, got {a.shape}")
        self.ctx.check(self.ctx.lib.bcg_gauge_upload(self.h, _dp(a)))

    def op(self, lhs, rhs):
        self.ctx.check(self.ctx.lib.bcg_dirac_apply(self.ctx.h, self.h, self.mass, lhs.h, rhs.h))

    def D(self, lhs, rhs):
        """The reference's private hop (inc/dirac_op.hpp:14-21), exposed for tests.  Half-volume fields: lhs of the
        parity opposite to rhs's."""
        if getattr(rhs, "parity", None) is not None:
            self.ctx.check(self.ctx.lib.bcg_dirac_hop_half(self.ctx.h, self.h, lhs.h, rhs.h))
        else:
            self.ctx.check(self.ctx.lib.bcg_dirac_hop(self.ctx.h, self.h, lhs.h, rhs.h))


def SBCGrQ(X, B, D, sigma, eps=1.e-15, eps_shifts=1.e-15, max_iterations=1000000, trace_limit=0, consume_B=False,
           return_info=False):
    """inc/block_solvers.hpp:91-185.  X: list of fields (overwritten); returns operator applications."""
    ctx = B.ctx
    S = len(X)
    if len(sigma) != S:
        raise ValueError("number of shifts does not match number of solution vectors")  # :97-98
    m = B.N_rhs
    sig = np.ascontiguousarray(sigma, dtype=np.float64)
    Xh = (ctypes.c_void_p * S)(*[x.h for x in X])
    it = ctypes.c_int(0)
    res = ctypes.c_double(0.0)
    tr = None
    tr_p = None
    if trace_limit > 0:
        mats = np.zeros((trace_limit, 3 + 2 * S, m, m), dtype=np.complex128)
        rr = np.zeros((trace_limit, 1 + S), dtype=np.float64)
        tr = _lib.bcg_sbcgrq_trace(trace_limit, 0, _dp(mats), _dp(rr))
        tr_p = ctypes.byref(tr)
    ctx.check(ctx.lib.bcg_sbcgrq_solve(ctx.h, D.h, D.mass, Xh, B.h, S, _dp(sig), eps, eps_shifts, int(max_iterations),
                                       1 if consume_B else 0, ctypes.byref(it), ctypes.byref(res), tr_p))
    if not return_info:
        return it.value
    info = dict(iterations=it.value, residual=res.value, trace=None)
    if tr is not None:
        n = tr.recorded
        mt = np.ascontiguousarray(np.swapaxes(mats[:n], -1, -2))
        info["trace"] = dict(alpha=mt[:, 0], rho=mt[:, 1], delta=mt[:, 2], alpha_s=mt[:, 3:3 + S],
                             beta_s=mt[:, 3 + S:3 + 2 * S], residual=rr[:n, 0], residual_shift=rr[:n, 1:])
    return info


class SBCGrQState:
    """The solver as a resumable state machine (bcg_sbcgrq_begin / iterate / end): lets bench.py run
    W warm-up iterations, then time exactly K iterations of the hot loop."""

    def __init__(self, X, B, D, sigma, eps=0.0, eps_shifts=0.0, consume_B=False):
        self.ctx = B.ctx
        self._keep = (X, B, D)
        S = len(X)
        if len(sigma) != S:
            raise ValueError("number of shifts does not match number of solution vectors")
        sig = np.ascontiguousarray(sigma, dtype=np.float64)
        Xh = (ctypes.c_void_p * S)(*[x.h for x in X])
        st = ctypes.c_void_p()
        self.ctx.check(self.ctx.lib.bcg_sbcgrq_begin(self.ctx.h, D.h, D.mass, Xh, B.h, S, _dp(sig), eps, eps_shifts,
                                                     1 if consume_B else 0, ctypes.byref(st)))
        self.h = st
        self.iterations = 0
        self.residual = 1.0

    def iterate(self, n):
        it = ctypes.c_int(0)
        res = ctypes.c_double(0.0)
        self.ctx.check(self.ctx.lib.bcg_sbcgrq_iterate(self.h, int(n), ctypes.byref(it), ctypes.byref(res), None))
        self.iterations, self.residual = it.value, res.value
        return it.value

    def end(self):
        if self.h:
            self.ctx.lib.bcg_sbcgrq_end(self.h)
            self.h = None

    def __del__(self):
        try:
            self.end()
        except Exception:
            pass


def true_residuals(X, B, D, sigma):
    """Reference acceptance measure (test/solvers.cpp:104-116) on the device; returns [n_shifts, N_rhs]."""
    ctx = B.ctx
    S = len(X)
    sig = np.ascontiguousarray(sigma, dtype=np.float64)
    Xh = (ctypes.c_void_p * S)(*[x.h for x in X])
    res = np.empty((S, B.N_rhs), dtype=np.float64)
    ctx.check(ctx.lib.bcg_true_residuals(ctx.h, D.h, D.mass, Xh, B.h, S, _dp(sig), _dp(res)))
    return res


def SBCGrQ_half_volume(X, B, D, sigma, eps=1.e-15, eps_shifts=1.e-15, max_iterations=1000000):
    """(op + sigma_s) X_s = B as two half-volume solves, one per site parity: dirac_op::D couples opposite parities only
    (inc/dirac_op.hpp:14-21), so op = mass^2 - D^2 (inc/dirac_op.hpp:36-43) is block diagonal in the parity.  X, B: full
    fields; each half solve runs inc/block_solvers.hpp:91-185 unchanged on half-volume fields (half the work fields'
    memory).  Returns the operator applications of the (even, odd) solve.  A caller short of memory keeps half fields
    only and calls SBCGrQ on them directly."""
    its = []
    halves = []
    for par, Bp in enumerate(B.split_parity()):
        Xp = [block_fermion_field(B.ctx, B.N_rhs, parity=par) for _ in X]
        its.append(SBCGrQ(Xp, Bp, D, sigma, eps, eps_shifts, max_iterations, consume_B=True))
        halves.append(Xp)
    for s, x in enumerate(X):
        x.merge_parity(halves[0][s], halves[1][s])
    return tuple(its)


def CG(x, b, D, eps=1.e-15, max_iterations=1000000):
    """src/standard_solvers.cpp:3-32"""
    it = ctypes.c_int(0)
    b.ctx.check(b.ctx.lib.bcg_cg_solve(b.ctx.h, D.h, D.mass, x.h, b.h, eps, int(max_iterations), ctypes.byref(it)))
    return it.value


def SCG(x, b, D, sigma, eps=1.e-15, eps_shifts=1.e-15, max_iterations=1000000):
    """src/standard_solvers.cpp:34-95"""
    S = len(x)
    if len(sigma) != S:
        raise ValueError("number of shifts does not match number of solution vectors")
    sig = np.ascontiguousarray(sigma, dtype=np.float64)
    xh = (ctypes.c_void_p * S)(*[v.h for v in x])
    it = ctypes.c_int(0)
    b.ctx.check(b.ctx.lib.bcg_scg_solve(b.ctx.h, D.h, D.mass, xh, b.h, S, _dp(sig), eps, eps_shifts, int(max_iterations),
                                        ctypes.byref(it)))
    return it.value


def BCG(X, B, D, eps=1.e-15, max_iterations=1000000):
    """inc/block_solvers.hpp:10-45"""
    it = ctypes.c_int(0)
    B.ctx.check(B.ctx.lib.bcg_bcg_solve(B.ctx.h, D.h, D.mass, X.h, B.h, eps, int(max_iterations), ctypes.byref(it)))
    return it.value


def BCGrQ(X, B, D, eps=1.e-15, max_iterations=1000000):
    """inc/block_solvers.hpp:50-86"""
    it = ctypes.c_int(0)
    B.ctx.check(B.ctx.lib.bcg_bcgrq_solve(B.ctx.h, D.h, D.mass, X.h, B.h, eps, int(max_iterations), ctypes.byref(it)))
    return it.value
