"""BASELINE config 0 (V = 1000, mass = 1e-3, eps = 1e-10, m = 4: condition number ~1e6): why the GPU needs 1632 iterations
where the reference needs ~1720, and what tolerance on the iteration count that evidence supports.

The count at this conditioning is set by rounding in the two Gram products per iteration (P^dag T and Q^dag Q): CG-type
recurrences lose conjugacy through those errors and pay for it in iterations.  Everything here runs on the CPU."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden


@pytest.fixture(scope="module")
def cfg0():
    g = load_golden("ref1d_v1000_m4.npz")
    return g, dict(U=g["U"], dims=[1000], mass=float(g["mass"]), B=g["B"], sigma=list(g["shifts"]), eps=float(g["eps"]),
                   eps_shifts=float(g["eps_shifts"]))


def _count(orc, c, mode):
    orc.set_gram_arith(mode)
    try:
        s = orc.sbcgrq(c["U"], c["dims"], c["mass"], c["B"], c["sigma"], c["eps"], c["eps_shifts"])
    finally:
        orc.set_gram_arith(0)
    res = orc.true_residuals(c["U"], c["dims"], c["mass"], c["B"], c["sigma"], s["X"]).max()
    return s["iterations"], res


def test_gram_summation_order_explains_the_gpu_iteration_count(orc, cfg0):
    g, c = cfg0
    ref_it = int(g["iterations"])                # the unmodified reference, oracle/Makefile's recipe
    seq, res_seq = _count(orc, c, 0)              # oracle, reference's sequential site sums
    tree, res_tree = _count(orc, c, 1)            # oracle, pairwise (tree) site sums: the GPU's reduction shape
    ld, res_ld = _count(orc, c, 2)                # oracle, sequential sums kept in long double
    # 1. in the reference's order the oracle reproduces the reference's count (+-2 %)
    assert abs(seq - ref_it) <= 0.02 * ref_it, (seq, ref_it)
    # 2. a more accurate Gram matrix -- by order (tree) or by precision (long double) -- needs FEWER iterations, by 4-9 %
    assert 0.91 * seq <= tree <= 0.96 * seq, (tree, seq)
    assert ld <= tree * 1.01 and ld >= 0.90 * seq, (ld, tree, seq)
    # 3. the GPU's measured count at this configuration (1632, tests/test_gpu_parity.py re-measures it) is the tree
    #    count to +-2 %, not the sequential one
    GPU_MEASURED = 1632
    assert abs(GPU_MEASURED - tree) <= 0.02 * tree and abs(GPU_MEASURED - seq) > 0.04 * seq
    # every variant satisfies the reference's acceptance test (true residual < 2 eps)
    assert max(res_seq, res_tree, res_ld) < 2 * c["eps"]


def test_the_reference_itself_moves_with_compiler_flags(cfg0, tmp_path):
    """The unmodified reference, same sources and inputs, counted 1700 / 1717 / 1729 / 1730 iterations when built with
    -ffp-contract=off / the Makefile's flags / -O2 / -march=native here (and 1761 in the build that wrote the round-1
    fixture): +-2 % is the resolution of this number, on either side of the comparison."""
    if not os.path.isdir("/root/reference/inc"):
        pytest.skip("/root/reference is not present on this machine")
    import oracle
    g, c = cfg0
    lib = tmp_path / "libref1d_nocontract.so"
    r = subprocess.run(["g++", "-std=c++11", "-O3", "-march=x86-64-v3", "-ffp-contract=off", "-DEIGEN_NO_DEBUG", "-DNDEBUG",
                        "-fPIC", "-w", "-I/root/reference/inc", "-shared", "-o", str(lib),
                        os.path.join(ROOT, "oracle", "ref_harness.cpp"), "/root/reference/src/standard_solvers.cpp"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    real = os.path.join
    oracle.os.path.join = lambda *a: str(lib) if str(a[-1]) == "libref1d.so" else real(*a)
    try:
        R = oracle.Reference(four_d=False)
    finally:
        oracle.os.path.join = real
    U = R.make_dirac_1d(1000, c["mass"], 1)
    B = R.field_random(4, 1000)
    assert np.array_equal(U, g["U"]) and np.array_equal(B, g["B"])   # same lattice and sources as the fixture
    it = R.sbcgrq(B, c["sigma"], c["eps"], c["eps_shifts"])["iterations"]
    ref_it = int(g["iterations"])
    assert it != ref_it and abs(it - ref_it) <= 0.02 * ref_it, (it, ref_it)
