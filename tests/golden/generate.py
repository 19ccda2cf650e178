#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE UNMODIFIED REFERENCE.

Run only in the build container, where /root/reference exists:
    make -C oracle ref && python tests/golden/generate.py
The reference is reached through oracle/_ref/libref1d.so and libref4d.so (oracle/ref_harness.cpp
compiled against /root/reference/inc).  Fixtures hold VALUES (inputs and the reference's outputs),
never seeds: Eigen::setRandom draws from std::rand() in a compiler-dependent order
(SURVEY.md section 8c).

Files written (numpy .npz, complex128 in the reference's host layout, see oracle/__init__.py):
  ref1d_v128_m3.npz    the reference's own test configuration (test/solvers.cpp:8-17):
                       inputs, every field primitive's output, the solve, and the solution after
                       k = 1..5 iterations (max_iterations = k) to pin the early iterates.
  ref1d_v128_m1.npz    same lattice, N_rhs = 1 (fermion_field), 2 shifts.
  ref1d_v1000_m4.npz   BASELINE.json config 0: V=1000, mass=1e-3, tol=1e-10, m=4, 1 shift.
  ref1d_v48_m12.npz    benchmark.cpp's block width (N_rhs = 12) and its nine shifts on a small lattice.
  ref4d_4x4x4x6_m4.npz ref4d_4x4x2x2_m16.npz ref4d_6x4x4x2_m8.npz
                       reference SBCGrQ + reference field arithmetic over the substitute n-D operator
                       (full data).
  ref1d_v128_other_solvers.npz  CG, SCG (N_rhs = 1), BCG, BCGrQ (N_rhs = 3) at the reference's test configuration.
  ref4d_8x8x8x8_m4.npz inputs from the oracle's counter-based generator (seeds stored), summary of
                       the reference solve only (iterations, residuals, column norms of X).
  ref1d_v96_m5.npz ref4d_4x2x4x2_m7.npz   odd block widths (the reference's N_rhs is any int), full data.
`generate.py NAME.npz ...` writes the named fixtures only.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def rand_matrix(rng, m):
    return rng.uniform(-1, 1, (m, m)) + 1j * rng.uniform(-1, 1, (m, m))


def primitives(R, m, V, B, rng):
    """Outputs of every reference field primitive on fixed inputs."""
    Y = R.field_random(m, V)
    M = rand_matrix(rng, m)
    d = dict(Y=Y, M=M)
    d["add_scalar_0p3"] = R.add_scalar(Y, B, 0.3)
    d["rescale_add_scalar_m1_0p25"] = R.rescale_add_scalar(Y, -1.0, B, 0.25)
    d["add_matrix"] = R.add_matrix(Y, B, M)
    d["rescale_add_matrix_1"] = R.rescale_add_matrix(Y, M, B, 1.0)
    d["hermitian_dot_YB"] = R.hermitian_dot(Y, B)
    d["hermitian_dot_YY"] = R.hermitian_dot(Y, Y)
    q, r = R.thin_qr(Y)
    d["thinqr_Q"] = q
    d["thinqr_R"] = r
    d["tri_solve"] = R.tri_solve_rhs(B, r)
    d["sub"] = R.sub(Y, B)
    d["chol_upper"] = R.cholesky_upper(d["hermitian_dot_YY"])
    d["inverse_M"] = R.inverse(M)
    d["op_B"] = R.dirac_op(B)
    return d


def solve(R, B, shifts, eps, eps_shifts, early=0):
    s = R.sbcgrq(B, shifts, eps, eps_shifts)
    d = dict(X=s["X"], iterations=np.int64(s["iterations"]), residuals=R.true_residuals(B, shifts, s["X"]))
    for k in range(1, early + 1):
        d[f"X_after_{k}"] = R.sbcgrq(B, shifts, 0.0, 0.0, max_iterations=k)["X"]
    return d


ONLY = set(sys.argv[1:])  # `generate.py NAME.npz ...`: write the named fixtures only (the others stay as committed)


def gen_1d(name, V, m, mass, shifts, eps, eps_shifts, seed, early, with_primitives=True):
    if ONLY and name not in ONLY:
        return
    R = oracle.Reference(four_d=False)
    U = R.make_dirac_1d(V, mass, seed)  # srand(seed); dirac_op D(V, mass);  rand() state continues into B
    B = R.field_random(m, V)
    rng = np.random.default_rng(1234 + m)
    d = dict(dims=np.array([V]), mass=mass, shifts=np.array(shifts), eps=eps, eps_shifts=eps_shifts, U=U, B=B)
    if with_primitives:
        d.update(primitives(R, m, V, B, rng))
    d.update(solve(R, B, shifts, eps, eps_shifts, early))
    np.savez(os.path.join(OUT, name), **d)
    print(name, "iterations", int(d["iterations"]), "max residual", d["residuals"].max())


def gen_nd(name, dims, m, mass, shifts, eps, eps_shifts, seed, early, full=True):
    if ONLY and name not in ONLY:
        return
    O = oracle.Oracle()
    R = oracle.Reference(four_d=True)
    V = int(np.prod(dims))
    U = O.fill_gauge(dims, seed)
    B = O.fill_field(m, V, seed + 1)
    R.make_dirac_nd(dims, mass, U)
    R.srand(seed)
    rng = np.random.default_rng(99 + m)
    d = dict(dims=np.array(dims), mass=mass, shifts=np.array(shifts), eps=eps, eps_shifts=eps_shifts,
             seed_U=np.uint64(seed), seed_B=np.uint64(seed + 1))
    if full:
        d.update(U=U, B=B)
        d.update(primitives(R, m, V, B, rng))
        d.update(solve(R, B, shifts, eps, eps_shifts, early))
    else:
        s = solve(R, B, shifts, eps, eps_shifts, 0)
        X = s.pop("X")
        d.update(s)
        d["X_colnorm"] = np.sqrt((np.abs(X) ** 2).sum(axis=(1, 3)))  # [S, m]
        d["X_sites"] = X[:, :4].copy()  # first four sites of every shift
        d["op_B_sites"] = R.dirac_op(B)[:4].copy()
    np.savez(os.path.join(OUT, name), **d)
    print(name, "iterations", int(d["iterations"]), "max residual", d["residuals"].max())


def gen_other_solvers(name):
    """CG, SCG, BCG, BCGrQ of the unmodified reference at its own test configuration (test/solvers.cpp:8-91)."""
    if ONLY and name not in ONLY:
        return
    V, mass, eps = 128, 0.5, 1e-10
    shifts = [0.0, 0.01, 0.10, 0.20, 0.9]
    R = oracle.Reference(four_d=False)
    U = R.make_dirac_1d(V, mass, 1)
    b = R.field_random(1, V)
    B = R.field_random(3, V)
    d = dict(dims=np.array([V]), mass=mass, eps=eps, shifts=np.array(shifts), U=U, b=b, B=B)
    d["x_cg"], it = R.cg(b, eps); d["it_cg"] = np.int64(it)
    d["x_scg"], it = R.scg(b, shifts, eps); d["it_scg"] = np.int64(it)
    d["X_bcg"], it = R.bcg(B, eps, with_qr=False); d["it_bcg"] = np.int64(it)
    d["X_bcgrq"], it = R.bcg(B, eps, with_qr=True); d["it_bcgrq"] = np.int64(it)
    np.savez(os.path.join(OUT, name), **d)
    print(name, "iterations CG/SCG/BCG/BCGrQ", int(d["it_cg"]), int(d["it_scg"]), int(d["it_bcg"]), int(d["it_bcgrq"]))


def main():
    if not oracle.ref_available():
        sys.exit("oracle/_ref is missing: run `make -C oracle ref` where /root/reference exists")
    test_shifts = [0.0, 0.01, 0.10, 0.20, 0.9]  # test/solvers.cpp:16
    bench_shifts = [0, 0, 1e-10, 1e-8, 1e-6, 1e-5, 1e-4, 1e-2, 1e-1]  # benchmark.cpp:12-13
    gen_1d("ref1d_v128_m3.npz", 128, 3, 0.5, test_shifts, 1e-10, 1e-15, 1, early=5)
    gen_1d("ref1d_v128_m1.npz", 128, 1, 0.5, [0.0, 0.1], 1e-10, 1e-15, 1, early=2)
    gen_1d("ref1d_v1000_m4.npz", 1000, 4, 1e-3, [0.0], 1e-10, 1e-15, 1, early=0, with_primitives=False)
    gen_1d("ref1d_v48_m12.npz", 48, 12, 0.1, bench_shifts, 1e-10, 1e-15, 1, early=1)
    s4 = [0.0, 1e-6, 1e-4, 1e-2]
    gen_nd("ref4d_4x4x4x6_m4.npz", [4, 4, 4, 6], 4, 0.1, s4, 1e-10, 1e-12, 11, early=2)
    gen_nd("ref4d_4x4x2x2_m16.npz", [4, 4, 2, 2], 16, 0.2, s4, 1e-10, 1e-12, 12, early=2)
    gen_nd("ref4d_6x4x4x2_m8.npz", [6, 4, 4, 2], 8, 0.1, [0.0], 1e-10, 1e-12, 13, early=2)
    gen_nd("ref4d_8x8x8x8_m4.npz", [8, 8, 8, 8], 4, 0.05, s4, 1e-10, 1e-10, 14, early=0, full=False)
    # the widest MFMA path (BASELINE config 4: m = 32, 8 shifts) and the one generic width without a fixture so far
    s8 = sorted([0.0, 1e-6, 1e-4, 1e-2, 1e-5, 1e-3, 1e-1, 1.0])
    gen_nd("ref4d_4x2x2x4_m32.npz", [4, 2, 2, 4], 32, 0.3, s8, 1e-10, 1e-12, 15, early=1)
    gen_1d("ref1d_v64_m6.npz", 64, 6, 0.4, [0.0, 0.05], 1e-10, 1e-15, 2, early=2)
    gen_other_solvers("ref1d_v128_other_solvers.npz")
    # odd block widths: the reference's N_rhs is an arbitrary template int (inc/fields.hpp:19-26)
    gen_1d("ref1d_v96_m5.npz", 96, 5, 0.3, [0.0, 0.02, 0.5], 1e-10, 1e-15, 3, early=2)
    gen_nd("ref4d_4x2x4x2_m7.npz", [4, 2, 4, 2], 7, 0.15, s4, 1e-10, 1e-12, 16, early=2)


if __name__ == "__main__":
    main()
