"""oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes bindings over
  * oracle/_build/liboracle.so : this repository's CPU restatement of the reference's SBCGrQ hot
    path (oracle/oracle.hpp, each function cites the reference file:line it follows), and
  * oracle/_ref/libref1d.so, oracle/_ref/libref4d.so : the UNMODIFIED reference compiled from
    /root/reference (oracle/ref_harness.cpp); present only where `make -C oracle ref` has run.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package, and
only as the checker.  blockcg_amd (the product) never imports it.

All arrays are numpy complex128 in the reference's host layout:
  field  [V, m, 3]      (site, rhs j, colour c)      inc/fields.hpp:19-20,28-30
  gauge  [V, ndim, 3, 3] with [.., k, r] = U(r, k)   (column-major 3x3)   inc/dirac_op.hpp:10-11
  m x m  numpy [m, m] in the usual (row, col) convention; converted to/from column-major here.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_c_int_p = ctypes.POINTER(ctypes.c_int)
_c_dbl_p = ctypes.POINTER(ctypes.c_double)

SUPPORTED_M = tuple(range(1, 33))
REF_SUPPORTED_M = (1, 2, 3, 4, 5, 6, 7, 8, 12, 16, 32)


def build(ref=None):
    """Compile liboracle.so (always) and the reference harness (when /root/reference exists)."""
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    if ref is None:
        ref = os.path.isdir("/root/reference/inc")
    if ref:
        subprocess.run(["make", "-C", _HERE, "-s", "-j2", "ref"], check=True)
        # the reference's unmodified drivers against the drop-in headers (needs the product library to link)
        if os.path.exists(os.path.join(_HERE, "..", "blockcg_amd", "_build", "libblockcg_hip.so")):
            subprocess.run(["make", "-C", _HERE, "-s", "-j2", "dropin"], check=True)


def _dp(a):
    return a.ctypes.data_as(_c_dbl_p)


def _dims(dims):
    d = list(dims) + [1] * (4 - len(dims))
    return (ctypes.c_int * 4)(*d)


def _mat_in(M):
    """numpy (row, col) -> column-major buffer."""
    return np.ascontiguousarray(np.asarray(M, dtype=np.complex128).T)


def _mat_out(buf):
    return np.ascontiguousarray(buf.T)


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.complex128)
    assert a.ndim == 3 and a.shape[2] == 3, "field must be [V, m, 3]"
    return a


class Oracle:
    """The CPU restatement (liboracle.so)."""

    def __init__(self):
        path = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        self.lib = ctypes.CDLL(path)
        self.lib.orc_fill_field.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_uint64, _c_dbl_p]
        self.lib.orc_fill_gauge.argtypes = [ctypes.c_int, _c_int_p, ctypes.c_uint64, _c_dbl_p]
        self.lib.orc_add_scalar.argtypes = [ctypes.c_int, ctypes.c_int64, _c_dbl_p, _c_dbl_p, ctypes.c_double]
        self.lib.orc_rescale_add_scalar.argtypes = [ctypes.c_int, ctypes.c_int64, _c_dbl_p, ctypes.c_double, _c_dbl_p,
                                                    ctypes.c_double]
        self.lib.orc_rescale_add_matrix.argtypes = [ctypes.c_int, ctypes.c_int64, _c_dbl_p, _c_dbl_p, _c_dbl_p,
                                                    ctypes.c_double]
        self.lib.orc_dirac_apply.argtypes = [ctypes.c_int, ctypes.c_int, _c_int_p, _c_dbl_p, ctypes.c_double, _c_dbl_p,
                                             _c_dbl_p]
        self.lib.orc_sbcgrq.argtypes = [ctypes.c_int, ctypes.c_int, _c_int_p, _c_dbl_p, ctypes.c_double, _c_dbl_p,
                                        ctypes.c_int, _c_dbl_p, ctypes.c_double, ctypes.c_double, ctypes.c_int, _c_dbl_p,
                                        _c_int_p, ctypes.c_int, _c_dbl_p, _c_dbl_p, _c_dbl_p]
        self.lib.orc_true_residuals.argtypes = [ctypes.c_int, ctypes.c_int, _c_int_p, _c_dbl_p, ctypes.c_double, _c_dbl_p,
                                                ctypes.c_int, _c_dbl_p, _c_dbl_p, _c_dbl_p]
        self.lib.orc_bench_sbcgrq.argtypes = [ctypes.c_int, ctypes.c_int, _c_int_p, ctypes.c_double, ctypes.c_int, _c_dbl_p,
                                              ctypes.c_int, ctypes.c_uint64, _c_dbl_p, _c_dbl_p]

    @staticmethod
    def _chk(rc):
        if rc != 0:
            raise ValueError("oracle: unsupported block width m")

    # --- synthetic inputs (counter-based generator, oracle.hpp) ---
    def fill_field(self, m, nsites, seed, first_global_site=0):
        out = np.empty((nsites, m, 3), dtype=np.complex128)
        self.lib.orc_fill_field(m, nsites, first_global_site, seed, _dp(out))
        return out

    def fill_gauge(self, dims, seed):
        nd = len(dims)
        V = int(np.prod(dims))
        out = np.empty((V, nd, 3, 3), dtype=np.complex128)
        self.lib.orc_fill_gauge(nd, _dims(dims), seed, _dp(out))
        return out

    # --- host threads for the site loops (1 = the reference's sequential summation order) ---
    def set_threads(self, n):
        return self.lib.orc_set_threads(int(n))

    def set_gram_arith(self, mode):
        """0: the reference's sequential site sums; 1: pairwise (device-like); 2: long double accumulation."""
        return self.lib.orc_set_gram_arith(int(mode))

    # --- sampled evaluator: (D psi)(x), (A psi)(x) at chosen global sites of a generated lattice ---
    def hop_sampled(self, m, dims, seed_U, seed_psi, sites):
        sites = np.ascontiguousarray(sites, dtype=np.int64)
        out = np.empty((len(sites), m, 3), dtype=np.complex128)
        self.lib.orc_hop_sampled.argtypes = [ctypes.c_int, ctypes.c_int, _c_int_p, ctypes.c_uint64, ctypes.c_uint64,
                                             ctypes.c_int64, ctypes.POINTER(ctypes.c_int64), _c_dbl_p]
        self._chk(self.lib.orc_hop_sampled(m, len(dims), _dims(dims), seed_U, seed_psi, len(sites),
                                           sites.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), _dp(out)))
        return out

    def apply_sampled(self, m, dims, seed_U, seed_psi, mass, sites):
        sites = np.ascontiguousarray(sites, dtype=np.int64)
        out = np.empty((len(sites), m, 3), dtype=np.complex128)
        self.lib.orc_apply_sampled.argtypes = [ctypes.c_int, ctypes.c_int, _c_int_p, ctypes.c_uint64, ctypes.c_uint64,
                                               ctypes.c_double, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64), _c_dbl_p]
        self._chk(self.lib.orc_apply_sampled(m, len(dims), _dims(dims), seed_U, seed_psi, mass, len(sites),
                                             sites.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), _dp(out)))
        return out

    def gram_generated(self, m, nsites, seed_a, seed_b):
        """a^dagger b of two generated fields, chunked sums (no lattice-sized array); mirrored like hermitian_dot."""
        out = np.empty((m, m), dtype=np.complex128)
        self.lib.orc_gram_generated.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint64, _c_dbl_p]
        self._chk(self.lib.orc_gram_generated(m, nsites, seed_a, seed_b, _dp(out)))
        return _mat_out(out)

    def sbcgrq_generated(self, m, dims, seed_U, seed_B, mass, sigma, iterations):
        """SBCGrQ trace of the first `iterations` iterations on generated inputs (no lattice-sized numpy arrays)."""
        S = len(sigma)
        sig = np.ascontiguousarray(sigma, dtype=np.float64)
        tr_m = np.zeros((iterations, 3 + 2 * S, m, m), dtype=np.complex128)
        tr_r = np.zeros((iterations, 1 + S), dtype=np.float64)
        it = ctypes.c_int(0)
        self.lib.orc_sbcgrq_generated.argtypes = [ctypes.c_int, ctypes.c_int, _c_int_p, ctypes.c_uint64, ctypes.c_uint64,
                                                  ctypes.c_double, ctypes.c_int, _c_dbl_p, ctypes.c_int, _c_dbl_p, _c_dbl_p,
                                                  _c_int_p]
        self._chk(self.lib.orc_sbcgrq_generated(m, len(dims), _dims(dims), seed_U, seed_B, mass, S, _dp(sig), iterations,
                                                _dp(tr_m), _dp(tr_r), ctypes.byref(it)))
        mats = np.ascontiguousarray(np.swapaxes(tr_m, -1, -2))
        return dict(iterations=it.value, alpha=mats[:, 0], rho=mats[:, 1], delta=mats[:, 2], alpha_s=mats[:, 3:3 + S],
                    beta_s=mats[:, 3 + S:3 + 2 * S], residual=tr_r[:, 0], residual_shift=tr_r[:, 1:])

    # --- K1 + op ---
    def hop(self, U, dims, x):
        x = _f(x)
        out = np.empty_like(x)
        U = np.ascontiguousarray(U, dtype=np.complex128)
        self._chk(self.lib.orc_hop(x.shape[1], len(dims), _dims(dims), _dp(U), _dp(x), _dp(out)))
        return out

    def dirac_apply(self, U, dims, mass, x):
        x = _f(x)
        out = np.empty_like(x)
        U = np.ascontiguousarray(U, dtype=np.complex128)
        self._chk(self.lib.orc_dirac_apply(x.shape[1], len(dims), _dims(dims), _dp(U), mass, _dp(x), _dp(out)))
        return out

    # --- field primitives; each returns the updated copy of y ---
    def add_scalar(self, y, x, a):
        y = _f(y).copy(); x = _f(x)
        self._chk(self.lib.orc_add_scalar(y.shape[1], y.shape[0], _dp(y), _dp(x), float(a)))
        return y

    def rescale_add_scalar(self, y, a, x, b):
        y = _f(y).copy(); x = _f(x)
        self._chk(self.lib.orc_rescale_add_scalar(y.shape[1], y.shape[0], _dp(y), float(a), _dp(x), float(b)))
        return y

    def add_matrix(self, y, x, M):
        y = _f(y).copy(); x = _f(x); Mc = _mat_in(M)
        self._chk(self.lib.orc_add_matrix(y.shape[1], ctypes.c_int64(y.shape[0]), _dp(y), _dp(x), _dp(Mc)))
        return y

    def rescale_add_matrix(self, y, M, x, b=1.0):
        y = _f(y).copy(); x = _f(x); Mc = _mat_in(M)
        self._chk(self.lib.orc_rescale_add_matrix(y.shape[1], y.shape[0], _dp(y), _dp(Mc), _dp(x), float(b)))
        return y

    def hermitian_dot(self, a, b):
        a = _f(a); b = _f(b); m = a.shape[1]
        out = np.empty((m, m), dtype=np.complex128)
        self._chk(self.lib.orc_hermitian_dot(m, ctypes.c_int64(a.shape[0]), _dp(a), _dp(b), _dp(out)))
        return _mat_out(out)

    def tri_solve_rhs(self, y, R):
        y = _f(y).copy(); Rc = _mat_in(R)
        self._chk(self.lib.orc_tri_solve_rhs(y.shape[1], ctypes.c_int64(y.shape[0]), _dp(y), _dp(Rc)))
        return y

    def thin_qr(self, y):
        y = _f(y).copy(); m = y.shape[1]
        R = np.empty((m, m), dtype=np.complex128)
        self._chk(self.lib.orc_thin_qr(m, ctypes.c_int64(y.shape[0]), _dp(y), _dp(R)))
        return y, _mat_out(R)

    def sub(self, y, x):
        y = _f(y).copy(); x = _f(x)
        self._chk(self.lib.orc_sub(y.shape[1], ctypes.c_int64(y.shape[0]), _dp(y), _dp(x)))
        return y

    def cholesky_upper(self, G):
        Gc = _mat_in(G); R = np.empty_like(Gc)
        self.lib.orc_cholesky_upper(Gc.shape[0], _dp(Gc), _dp(R))
        return _mat_out(R)

    def inverse(self, A):
        Ac = _mat_in(A); R = np.empty_like(Ac)
        self.lib.orc_inverse(Ac.shape[0], _dp(Ac), _dp(R))
        return _mat_out(R)

    # --- solver ---
    def sbcgrq(self, U, dims, mass, B, sigma, eps=1e-15, eps_shifts=1e-15, max_iterations=1000000, trace_limit=0):
        """Returns dict(X=[S,V,m,3], iterations=int, seconds=float, trace=dict or None)."""
        B = _f(B); V, m, _ = B.shape; S = len(sigma)
        U = np.ascontiguousarray(U, dtype=np.complex128)
        sig = np.ascontiguousarray(sigma, dtype=np.float64)
        X = np.empty((S, V, m, 3), dtype=np.complex128)
        it = ctypes.c_int(0); sec = ctypes.c_double(0.0)
        tr_m = np.zeros((max(trace_limit, 1), 3 + 2 * S, m, m), dtype=np.complex128)
        tr_r = np.zeros((max(trace_limit, 1), 1 + S), dtype=np.float64)
        self._chk(self.lib.orc_sbcgrq(m, len(dims), _dims(dims), _dp(U), mass, _dp(B), S, _dp(sig), eps, eps_shifts,
                                      max_iterations, _dp(X), ctypes.byref(it), trace_limit, _dp(tr_m), _dp(tr_r),
                                      ctypes.byref(sec)))
        trace = None
        if trace_limit > 0:
            n = min(trace_limit, it.value)
            mats = np.ascontiguousarray(np.swapaxes(tr_m[:n], -1, -2))  # column-major -> (row, col)
            trace = dict(alpha=mats[:, 0], rho=mats[:, 1], delta=mats[:, 2], alpha_s=mats[:, 3:3 + S],
                         beta_s=mats[:, 3 + S:3 + 2 * S], residual=tr_r[:n, 0], residual_shift=tr_r[:n, 1:])
        return dict(X=X, iterations=it.value, seconds=sec.value, trace=trace)

    def true_residuals(self, U, dims, mass, B, sigma, X):
        B = _f(B); V, m, _ = B.shape; S = len(sigma)
        U = np.ascontiguousarray(U, dtype=np.complex128)
        X = np.ascontiguousarray(X, dtype=np.complex128)
        sig = np.ascontiguousarray(sigma, dtype=np.float64)
        res = np.empty((S, m), dtype=np.float64)
        self._chk(self.lib.orc_true_residuals(m, len(dims), _dims(dims), _dp(U), mass, _dp(B), S, _dp(sig), _dp(X), _dp(res)))
        return res

    # --- the other four solvers ---
    def cg(self, U, dims, mass, b, eps=1e-15, max_iterations=1000000):
        b = _f(b); U = np.ascontiguousarray(U, dtype=np.complex128)
        x = np.empty_like(b); it = ctypes.c_int(0)
        self.lib.orc_cg.argtypes = [ctypes.c_int, _c_int_p, _c_dbl_p, ctypes.c_double, _c_dbl_p, ctypes.c_double, ctypes.c_int,
                                    _c_dbl_p, _c_int_p]
        self.lib.orc_cg(len(dims), _dims(dims), _dp(U), mass, _dp(b), eps, max_iterations, _dp(x), ctypes.byref(it))
        return x, it.value

    def scg(self, U, dims, mass, b, sigma, eps=1e-15, eps_shifts=1e-15, max_iterations=1000000):
        b = _f(b); U = np.ascontiguousarray(U, dtype=np.complex128); S = len(sigma)
        sig = np.ascontiguousarray(sigma, dtype=np.float64)
        x = np.empty((S,) + b.shape, dtype=np.complex128); it = ctypes.c_int(0)
        self.lib.orc_scg.argtypes = [ctypes.c_int, _c_int_p, _c_dbl_p, ctypes.c_double, _c_dbl_p, ctypes.c_int, _c_dbl_p,
                                     ctypes.c_double, ctypes.c_double, ctypes.c_int, _c_dbl_p, _c_int_p]
        self.lib.orc_scg(len(dims), _dims(dims), _dp(U), mass, _dp(b), S, _dp(sig), eps, eps_shifts, max_iterations, _dp(x),
                         ctypes.byref(it))
        return x, it.value

    def bcg(self, U, dims, mass, B, eps=1e-15, max_iterations=1000000, with_qr=False):
        B = _f(B); U = np.ascontiguousarray(U, dtype=np.complex128)
        X = np.empty_like(B); it = ctypes.c_int(0)
        self.lib.orc_bcg.argtypes = [ctypes.c_int, ctypes.c_int, _c_int_p, _c_dbl_p, ctypes.c_double, _c_dbl_p, ctypes.c_double,
                                     ctypes.c_int, ctypes.c_int, _c_dbl_p, _c_int_p]
        self._chk(self.lib.orc_bcg(B.shape[1], len(dims), _dims(dims), _dp(U), mass, _dp(B), eps, max_iterations,
                                   1 if with_qr else 0, _dp(X), ctypes.byref(it)))
        return X, it.value

    def bench_sbcgrq(self, m, dims, mass, sigma, iterations, seed=1):
        """Fixed-work single-thread CPU run on synthetic inputs; returns (seconds_for_iterations, seconds_setup)."""
        sig = np.ascontiguousarray(sigma, dtype=np.float64)
        tot = ctypes.c_double(0.0); setup = ctypes.c_double(0.0)
        self._chk(self.lib.orc_bench_sbcgrq(m, len(dims), _dims(dims), mass, len(sigma), _dp(sig), iterations, seed,
                                            ctypes.byref(tot), ctypes.byref(setup)))
        return tot.value - setup.value, setup.value


def ref_available():
    return all(os.path.exists(os.path.join(_HERE, "_ref", n)) for n in ("libref1d.so", "libref4d.so"))


class Reference:
    """The unmodified reference (libref1d.so) or its solver over the substitute n-D operator (libref4d.so)."""

    def __init__(self, four_d=False):
        self.four_d = four_d
        self.lib = ctypes.CDLL(os.path.join(_HERE, "_ref", "libref4d.so" if four_d else "libref1d.so"))
        self.lib.ref_dirac_create.restype = ctypes.c_void_p
        if four_d:
            self.lib.ref_dirac_create.argtypes = [ctypes.c_int, _c_int_p, ctypes.c_double, _c_dbl_p]
        else:
            self.lib.ref_dirac_create.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_uint, _c_dbl_p]
        self.lib.ref_dirac_destroy.argtypes = [ctypes.c_void_p]
        self.lib.ref_dirac_op.argtypes = [ctypes.c_void_p, ctypes.c_int, _c_dbl_p, _c_dbl_p]
        self.lib.ref_add_scalar.argtypes = [ctypes.c_int, ctypes.c_int, _c_dbl_p, _c_dbl_p, ctypes.c_double]
        self.lib.ref_rescale_add_scalar.argtypes = [ctypes.c_int, ctypes.c_int, _c_dbl_p, ctypes.c_double, _c_dbl_p,
                                                    ctypes.c_double]
        self.lib.ref_rescale_add_matrix.argtypes = [ctypes.c_int, ctypes.c_int, _c_dbl_p, _c_dbl_p, _c_dbl_p,
                                                    ctypes.c_double]
        self.lib.ref_sbcgrq.argtypes = [ctypes.c_void_p, ctypes.c_int, _c_dbl_p, ctypes.c_int, _c_dbl_p, ctypes.c_double,
                                        ctypes.c_double, ctypes.c_int, _c_dbl_p, _c_int_p, _c_dbl_p]
        self.lib.ref_true_residuals.argtypes = [ctypes.c_void_p, ctypes.c_int, _c_dbl_p, ctypes.c_int, _c_dbl_p, _c_dbl_p,
                                                _c_dbl_p]
        self._h = None
        self.V = None

    def srand(self, seed):
        self.lib.ref_srand(ctypes.c_uint(seed))

    def make_dirac_1d(self, V, mass, seed):
        """Reference dirac_op(V, mass) after srand(seed); returns its (private) links by replay, [V,1,3,3]."""
        assert not self.four_d
        self.destroy()
        U = np.empty((V, 1, 3, 3), dtype=np.complex128)
        self._h = self.lib.ref_dirac_create(V, mass, seed, _dp(U))
        self.V = V
        return U

    def make_dirac_nd(self, dims, mass, U):
        assert self.four_d
        self.destroy()
        U = np.ascontiguousarray(U, dtype=np.complex128)
        self._h = self.lib.ref_dirac_create(len(dims), _dims(dims), mass, _dp(U))
        self.V = int(np.prod(dims))

    def destroy(self):
        if self._h:
            self.lib.ref_dirac_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    @staticmethod
    def _chk(rc):
        if rc != 0:
            raise ValueError("reference harness: block width not instantiated")

    def field_random(self, m, V):
        out = np.empty((V, m, 3), dtype=np.complex128)
        self._chk(self.lib.ref_field_random(m, V, _dp(out)))
        return out

    def dirac_op(self, x):
        x = _f(x); out = np.empty_like(x)
        self._chk(self.lib.ref_dirac_op(self._h, x.shape[1], _dp(x), _dp(out)))
        return out

    def add_scalar(self, y, x, a):
        y = _f(y).copy(); x = _f(x)
        self._chk(self.lib.ref_add_scalar(y.shape[1], y.shape[0], _dp(y), _dp(x), float(a)))
        return y

    def rescale_add_scalar(self, y, a, x, b):
        y = _f(y).copy(); x = _f(x)
        self._chk(self.lib.ref_rescale_add_scalar(y.shape[1], y.shape[0], _dp(y), float(a), _dp(x), float(b)))
        return y

    def add_matrix(self, y, x, M):
        y = _f(y).copy(); x = _f(x); Mc = _mat_in(M)
        self._chk(self.lib.ref_add_matrix(y.shape[1], y.shape[0], _dp(y), _dp(x), _dp(Mc)))
        return y

    def rescale_add_matrix(self, y, M, x, b=1.0):
        y = _f(y).copy(); x = _f(x); Mc = _mat_in(M)
        self._chk(self.lib.ref_rescale_add_matrix(y.shape[1], y.shape[0], _dp(y), _dp(Mc), _dp(x), float(b)))
        return y

    def hermitian_dot(self, a, b):
        a = _f(a); b = _f(b); m = a.shape[1]
        out = np.empty((m, m), dtype=np.complex128)
        self._chk(self.lib.ref_hermitian_dot(m, a.shape[0], _dp(a), _dp(b), _dp(out)))
        return _mat_out(out)

    def tri_solve_rhs(self, y, R):
        y = _f(y).copy(); Rc = _mat_in(R)
        self._chk(self.lib.ref_tri_solve_rhs(y.shape[1], y.shape[0], _dp(y), _dp(Rc)))
        return y

    def thin_qr(self, y):
        y = _f(y).copy(); m = y.shape[1]
        R = np.empty((m, m), dtype=np.complex128)
        self._chk(self.lib.ref_thin_qr(m, y.shape[0], _dp(y), _dp(R)))
        return y, _mat_out(R)

    def sub(self, y, x):
        y = _f(y).copy(); x = _f(x)
        self._chk(self.lib.ref_sub(y.shape[1], y.shape[0], _dp(y), _dp(x)))
        return y

    def cholesky_upper(self, G):
        Gc = _mat_in(G); R = np.empty_like(Gc)
        self._chk(self.lib.ref_cholesky_upper(Gc.shape[0], _dp(Gc), _dp(R)))
        return _mat_out(R)

    def inverse(self, A):
        Ac = _mat_in(A); R = np.empty_like(Ac)
        self._chk(self.lib.ref_inverse(Ac.shape[0], _dp(Ac), _dp(R)))
        return _mat_out(R)

    def sbcgrq(self, B, sigma, eps=1e-15, eps_shifts=1e-15, max_iterations=1000000):
        B = _f(B); V, m, _ = B.shape; S = len(sigma)
        sig = np.ascontiguousarray(sigma, dtype=np.float64)
        X = np.empty((S, V, m, 3), dtype=np.complex128)
        it = ctypes.c_int(0); sec = ctypes.c_double(0.0)
        self._chk(self.lib.ref_sbcgrq(self._h, m, _dp(B), S, _dp(sig), eps, eps_shifts, max_iterations, _dp(X),
                                      ctypes.byref(it), ctypes.byref(sec)))
        return dict(X=X, iterations=it.value, seconds=sec.value)

    def true_residuals(self, B, sigma, X):
        B = _f(B); V, m, _ = B.shape; S = len(sigma)
        sig = np.ascontiguousarray(sigma, dtype=np.float64)
        X = np.ascontiguousarray(X, dtype=np.complex128)
        res = np.empty((S, m), dtype=np.float64)
        self._chk(self.lib.ref_true_residuals(self._h, m, _dp(B), S, _dp(sig), _dp(X), _dp(res)))
        return res

    def bcg(self, B, eps=1e-15, max_iterations=1000000, with_qr=False):
        B = _f(B); X = np.empty_like(B); it = ctypes.c_int(0)
        self.lib.ref_bcg.argtypes = [ctypes.c_void_p, ctypes.c_int, _c_dbl_p, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                     _c_dbl_p, _c_int_p]
        self._chk(self.lib.ref_bcg(self._h, B.shape[1], _dp(B), eps, max_iterations, 1 if with_qr else 0, _dp(X),
                                   ctypes.byref(it)))
        return X, it.value

    def cg(self, b, eps=1e-15, max_iterations=1000000):
        assert not self.four_d
        b = _f(b); x = np.empty_like(b); it = ctypes.c_int(0)
        self.lib.ref_cg.argtypes = [ctypes.c_void_p, _c_dbl_p, ctypes.c_double, ctypes.c_int, _c_dbl_p, _c_int_p]
        self.lib.ref_cg(self._h, _dp(b), eps, max_iterations, _dp(x), ctypes.byref(it))
        return x, it.value

    def scg(self, b, sigma, eps=1e-15, eps_shifts=1e-15, max_iterations=1000000):
        assert not self.four_d
        b = _f(b); S = len(sigma); sig = np.ascontiguousarray(sigma, dtype=np.float64)
        x = np.empty((S,) + b.shape, dtype=np.complex128); it = ctypes.c_int(0)
        self.lib.ref_scg.argtypes = [ctypes.c_void_p, _c_dbl_p, ctypes.c_int, _c_dbl_p, ctypes.c_double, ctypes.c_double,
                                     ctypes.c_int, _c_dbl_p, _c_int_p]
        self.lib.ref_scg(self._h, _dp(b), S, _dp(sig), eps, eps_shifts, max_iterations, _dp(x), ctypes.byref(it))
        return x, it.value
