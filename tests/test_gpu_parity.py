"""GPU parity tests proper: the HIP path (through the C ABI) against
  (1) golden fixtures computed by the unmodified reference (tests/golden/*.npz),
  (2) the CPU oracle on the same seeded inputs at sizes it finishes in seconds,
  (3) size-independent properties at BASELINE.json's full sizes.
Tolerances are the fp64 ones of SURVEY.md Appendix F (tests/conftest.py).
"""
import os

import numpy as np
import pytest

from conftest import TOL_COEFF, TOL_KERNEL, TOL_SOLUTION, golden_files, rel_err

pytestmark = pytest.mark.gpu

FULL = [f for f in golden_files() if "8x8x8x8" not in f and "other_solvers" not in f]
WITH_PRIMS = [f for f in FULL if "v1000" not in f]


@pytest.fixture(scope="module")
def bc():
    import blockcg_amd
    return blockcg_amd


def _dims(g):
    return [int(d) for d in g["dims"]]


def _setup(bc, g):
    ctx = bc.Context(_dims(g))
    D = bc.dirac_op(ctx, float(g["mass"]), U=g["U"])
    return ctx, D


@pytest.mark.parametrize("path", WITH_PRIMS, ids=os.path.basename)
def test_primitives_match_reference_fixture(bc, path):
    g = np.load(path)
    ctx, D = _setup(bc, g)
    m = g["B"].shape[1]
    F = lambda a: bc.block_fermion_field(ctx, m, a)  # noqa: E731
    B, Y, M = g["B"], g["Y"], g["M"]
    # upload / download round trip is exact
    assert np.array_equal(F(B).download(), B)
    out = bc.block_fermion_field(ctx, m)
    D.op(out, F(B))
    assert rel_err(out.download(), g["op_B"]) < TOL_KERNEL
    assert rel_err(F(Y).add(F(B), 0.3).download(), g["add_scalar_0p3"]) < TOL_KERNEL
    assert rel_err(F(Y).rescale_add(-1.0, F(B), 0.25).download(), g["rescale_add_scalar_m1_0p25"]) < TOL_KERNEL
    assert rel_err(F(Y).add(F(B), M).download(), g["add_matrix"]) < TOL_KERNEL
    assert rel_err(F(Y).rescale_add(M, F(B), 1.0).download(), g["rescale_add_matrix_1"]) < TOL_KERNEL
    assert rel_err(F(Y).hermitian_dot(F(B)), g["hermitian_dot_YB"]) < TOL_KERNEL
    fy = F(Y)
    G = fy.hermitian_dot(fy)
    assert rel_err(G, g["hermitian_dot_YY"]) < TOL_KERNEL
    assert np.array_equal(np.triu(G, 1), np.conj(np.tril(G, -1)).T)  # exactly Hermitian, inc/fields.hpp:115-120
    q = F(Y)
    R = q.thinQR()
    assert rel_err(R, g["thinqr_R"]) < TOL_KERNEL
    assert rel_err(q.download(), g["thinqr_Q"]) < 1e-12
    assert rel_err(F(B).multiply_upper_triangular_inverse_RHS(g["thinqr_R"]).download(), g["tri_solve"]) < TOL_KERNEL
    fy = F(Y)
    fy -= F(B)
    assert rel_err(fy.download(), g["sub"]) < TOL_KERNEL
    fy += F(B)
    assert rel_err(fy.download(), Y) < TOL_KERNEL


@pytest.mark.parametrize("path", FULL, ids=os.path.basename)
def test_solver_matches_reference_fixture(bc, orc, path):
    g = np.load(path)
    ctx, D = _setup(bc, g)
    m = g["B"].shape[1]
    dims, mass, shifts = _dims(g), float(g["mass"]), list(g["shifts"])
    eps, eps_s = float(g["eps"]), float(g["eps_shifts"])
    B = bc.block_fermion_field(ctx, m, g["B"])
    X = [bc.block_fermion_field(ctx, m) for _ in shifts]
    info = bc.SBCGrQ(X, B, D, shifts, eps, eps_s, trace_limit=5, return_info=True)
    ref_it = int(g["iterations"])
    # +-1 at well-conditioned configurations.  At mass = 1e-3 (condition number ~1e6, BASELINE config 0) the count depends
    # on the summation order of the two Gram products: the reference's one running sum over the sites loses more digits
    # than the GPU's tree-shaped reductions, and needs ~6 % more iterations (1717 on the reference, 1725 on the oracle in
    # the reference's order, 1625 on the oracle with pairwise site sums, 1632 on the GPU).  Shown on the CPU in
    # tests/test_iteration_count_sensitivity.py; here the GPU must land within SURVEY Appendix F's +-2 % of the oracle
    # run in the GPU's summation shape, and within 8 % of the reference's own count.
    if mass < 0.01:
        orc.set_gram_arith(1)
        try:
            tree_it = orc.sbcgrq(g["U"], dims, mass, g["B"], shifts, eps, eps_s)["iterations"]
        finally:
            orc.set_gram_arith(0)
        assert abs(info["iterations"] - tree_it) <= int(0.02 * tree_it), (info["iterations"], tree_it, ref_it)
        assert abs(info["iterations"] - ref_it) <= int(0.08 * ref_it)
    else:
        assert abs(info["iterations"] - ref_it) <= 1
    Xh = np.stack([x.download() for x in X])
    # the reference's acceptance criterion, recomputed independently on the CPU (test/solvers.cpp:104-116)
    res = orc.true_residuals(g["U"], dims, mass, g["B"], shifts, Xh)
    assert res[0].max() < 2 * eps
    assert np.all(res < np.maximum(2 * eps, 4 * g["residuals"]))
    if info["iterations"] == ref_it and mass >= 0.05:
        assert rel_err(Xh, g["X"]) < TOL_SOLUTION
    # coefficient matrices of the first iterations against the oracle (pinned to the reference)
    o = orc.sbcgrq(g["U"], dims, mass, g["B"], shifts, eps, eps_s, max_iterations=5, trace_limit=5)
    n = min(5, info["iterations"])
    for key in ("alpha", "rho", "delta", "alpha_s", "beta_s"):
        assert rel_err(info["trace"][key][:n], o["trace"][key][:n]) < TOL_COEFF, key
    assert np.allclose(info["trace"]["residual"][:n], o["trace"]["residual"][:n], rtol=1e-9)


def _report(line):
    """Measured parity figures the design documents quote ("report it"): printed, and appended to
    gpurun_out/parity_report.txt where that directory can be written (the GPU box copies it back)."""
    print(line)
    try:
        from conftest import ROOT
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "parity_report.txt"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


@pytest.mark.parametrize("m", [5, 7, 9, 10, 11, 13, 14, 15, 17, 19, 20, 23, 24, 27, 29, 31])
def test_every_block_width(bc, orc, m):
    """The reference's N_rhs is an arbitrary template int (inc/fields.hpp:19-26): every 1 <= m <= 32 exists here (generic
    kernels outside 8, 16, 32).  Per width: every field primitive and the operator against the oracle (pinned to the
    reference at m = 5 and 7 by the fixtures ref1d_v96_m5 / ref4d_4x2x4x2_m7, which the fixture tests above run on the
    GPU too), then four fixed iterations of SBCGrQ, X and every coefficient."""
    dims, mass, shifts = [6, 4, 2, 3], 0.2, [0.0, 1e-3, 0.5]
    ctx = bc.Context(dims)
    V = ctx.V
    D = bc.dirac_op(ctx, mass, seed=5)
    U = orc.fill_gauge(dims, 5)
    Bh, Yh = orc.fill_field(m, V, 6), orc.fill_field(m, V, 7)
    rng = np.random.default_rng(m)
    M = rng.uniform(-1, 1, (m, m)) + 1j * rng.uniform(-1, 1, (m, m))
    F = lambda a: bc.block_fermion_field(ctx, m, a)  # noqa: E731
    assert np.array_equal(F(Bh).download(), Bh)
    out = bc.block_fermion_field(ctx, m)
    D.op(out, F(Bh))
    assert rel_err(out.download(), orc.dirac_apply(U, dims, mass, Bh)) < TOL_KERNEL
    D.D(out, F(Bh))
    assert rel_err(out.download(), orc.hop(U, dims, Bh)) < TOL_KERNEL
    assert rel_err(F(Yh).add(F(Bh), 0.3).download(), orc.add_scalar(Yh, Bh, 0.3)) < TOL_KERNEL
    assert rel_err(F(Yh).add(F(Bh), M).download(), orc.add_matrix(Yh, Bh, M)) < TOL_KERNEL
    assert rel_err(F(Yh).rescale_add(M, F(Bh), 1.0).download(), orc.rescale_add_matrix(Yh, M, Bh, 1.0)) < TOL_KERNEL
    assert rel_err(F(Yh).hermitian_dot(F(Bh)), orc.hermitian_dot(Yh, Bh)) < TOL_KERNEL
    q = F(Yh)
    R = q.thinQR()
    qo, Ro = orc.thin_qr(Yh)
    assert rel_err(R, Ro) < TOL_KERNEL and rel_err(q.download(), qo) < 1e-12
    assert rel_err(F(Bh).multiply_upper_triangular_inverse_RHS(Ro).download(), orc.tri_solve_rhs(Bh, Ro)) < 1e-12
    X = [bc.block_fermion_field(ctx, m) for _ in shifts]
    info = bc.SBCGrQ(X, F(Bh), D, shifts, 0.0, 0.0, max_iterations=4, trace_limit=4, return_info=True)
    o = orc.sbcgrq(U, dims, mass, Bh, shifts, 0.0, 0.0, max_iterations=4, trace_limit=4)
    assert info["iterations"] == 4 and rel_err(np.stack([x.download() for x in X]), o["X"]) < 1e-11
    for key in ("alpha", "rho", "delta", "alpha_s", "beta_s"):
        assert rel_err(info["trace"][key], o["trace"][key]) < TOL_COEFF, key


def test_odd_block_width_beyond_the_range_is_refused(bc):
    ctx = bc.Context([8])
    for m in (0, 33, 64):
        with pytest.raises(bc.BlockCGError) as e:
            bc.block_fermion_field(ctx, m)
        assert e.value.code == 2  # BCG_ERR_UNSUPPORTED


def test_config0_solution_against_the_reference(bc, orc):
    """BASELINE config 0 (V = 1000, mass 1e-3, tol 1e-10, m = 4, 1 shift; fixture = the unmodified reference's run): the
    iteration counts side by side and || X_gpu - X_ref || / || X_ref ||, reported and bounded.  The count differs by the
    Gram summation order (test above, tests/test_iteration_count_sensitivity.py); the SOLUTION does not: both stop at a
    true residual of 1e-10 and the oracle in either summation order is within 7e-13 of the reference's X."""
    g = np.load(golden_files("ref1d_v1000_m4.npz")[0])
    ctx, D = _setup(bc, g)
    m = g["B"].shape[1]
    shifts, eps, eps_s = list(g["shifts"]), float(g["eps"]), float(g["eps_shifts"])
    B = bc.block_fermion_field(ctx, m, g["B"])
    X = [bc.block_fermion_field(ctx, m) for _ in shifts]
    it = bc.SBCGrQ(X, B, D, shifts, eps, eps_s)
    Xh = np.stack([x.download() for x in X])
    err = rel_err(Xh, g["X"])
    res = orc.true_residuals(g["U"], _dims(g), float(g["mass"]), g["B"], shifts, Xh)
    _report(f"config 0 (ref1d_v1000_m4): iterations GPU {it} / reference {int(g['iterations'])}; "
            f"|X_gpu - X_ref| / |X_ref| = {err:.3e}; max true residual {res.max():.3e} (reference {g['residuals'].max():.3e})")
    assert err < TOL_SOLUTION and res.max() < 2 * eps


def test_summary_fixture_8x8x8x8_against_the_reference(bc, orc):
    """The largest reference-run 4-D fixture (8^4, m = 4, 4 shifts: reference SBCGrQ + field arithmetic over the substitute
    operator; inputs from the shared counter generator, summary of the reference's solve stored): iteration count, column
    norms and the first sites of every X_s, and the operator on B, on the GPU."""
    g = np.load(golden_files("ref4d_8x8x8x8_m4.npz")[0])
    dims, mass, shifts = _dims(g), float(g["mass"]), list(g["shifts"])
    ctx = bc.Context(dims)
    D = bc.dirac_op(ctx, mass, seed=int(g["seed_U"]))
    B = bc.block_fermion_field(ctx, 4).setRandom(seed=int(g["seed_B"]))
    out = bc.block_fermion_field(ctx, 4)
    D.op(out, B)
    assert rel_err(out.download_sites(np.arange(4)), g["op_B_sites"]) < TOL_KERNEL
    X = [bc.block_fermion_field(ctx, 4) for _ in shifts]
    it = bc.SBCGrQ(X, B, D, shifts, float(g["eps"]), float(g["eps_shifts"]))
    assert abs(it - int(g["iterations"])) <= 1, (it, int(g["iterations"]))
    Xh = np.stack([x.download() for x in X])
    coln = np.sqrt((np.abs(Xh) ** 2).sum(axis=(1, 3)))
    assert rel_err(coln, g["X_colnorm"]) < 1e-6
    assert rel_err(Xh[:, :4], g["X_sites"]) < 1e-5
    res = bc.true_residuals(X, B, D, shifts)
    assert res[0].max() < 2 * float(g["eps"])
    _report(f"ref4d_8x8x8x8_m4: iterations GPU {it} / reference {int(g['iterations'])}; column norms {rel_err(coln, g['X_colnorm']):.2e}, "
            f"first sites {rel_err(Xh[:, :4], g['X_sites']):.2e}")


@pytest.mark.parametrize("dims,m,generic", [([6, 4, 4, 2], 3, False), ([4, 2, 6], 2, False), ([16, 4, 4, 6], 16, False),
                                             ([16, 4, 4, 6], 16, True), ([32, 4, 2, 4], 8, False), ([8, 4, 4, 4], 32, False)])
def test_nd_operator_against_the_reference_1d_operator_line_by_line(bc, orc, dims, m, generic):
    """D and A on the GPU against an evaluation that shares nothing with the repository's n-D formula: the reference's 1-D
    loop (inc/dirac_op.hpp:17-20, in numpy, pinned to the reference's fixture in tests/test_oracle_golden.py) along every
    lattice line of every direction with the staggered sign (conftest.hop_by_lines) -- generic and specialised stencils."""
    from conftest import hop_by_lines
    ctx = bc.Context(dims)
    ctx.force_generic(generic)
    D = bc.dirac_op(ctx, 0.3, seed=77)
    psi = bc.block_fermion_field(ctx, m).setRandom(seed=78)
    U, ph = orc.fill_gauge(dims, 77), orc.fill_field(m, ctx.V, 78)
    want = hop_by_lines(U, dims, ph)
    out = bc.block_fermion_field(ctx, m)
    D.D(out, psi)
    assert rel_err(out.download(), want) < TOL_KERNEL
    D.op(out, psi)
    assert rel_err(out.download(), 0.09 * ph - hop_by_lines(U, dims, want)) < TOL_KERNEL


@pytest.mark.parametrize("path", WITH_PRIMS, ids=os.path.basename)
def test_early_iterates_match_reference_fixture(bc, path):
    g = np.load(path)
    ctx, D = _setup(bc, g)
    m = g["B"].shape[1]
    shifts = list(g["shifts"])
    k = 1
    while f"X_after_{k}" in g.files:
        B = bc.block_fermion_field(ctx, m, g["B"])
        X = [bc.block_fermion_field(ctx, m) for _ in shifts]
        assert bc.SBCGrQ(X, B, D, shifts, 0.0, 0.0, max_iterations=k) == k
        assert rel_err(np.stack([x.download() for x in X]), g[f"X_after_{k}"]) < 1e-12
        k += 1


def test_generator_matches_oracle_bit_for_bit(bc, orc):
    dims = [6, 4, 2, 3]
    ctx = bc.Context(dims)
    f = bc.block_fermion_field(ctx, 3).setRandom(seed=42)
    assert np.array_equal(f.download(), orc.fill_field(3, ctx.V, 42))
    D = bc.dirac_op(ctx, 0.1, seed=7)
    U = orc.fill_gauge(dims, 7)
    x = orc.fill_field(3, ctx.V, 42)
    out = bc.block_fermion_field(ctx, 3)
    D.op(out, f)
    assert rel_err(out.download(), orc.dirac_apply(U, dims, 0.1, x)) < TOL_KERNEL


@pytest.mark.parametrize("m,dims,S", [(16, [8, 8, 8, 8], 4), (8, [16, 8, 8, 4], 1), (32, [4, 4, 4, 4], 8), (2, [5, 3, 7], 3),
                                      (6, [10, 6], 2)])
def test_fixed_work_against_oracle(bc, orc, m, dims, S):
    """Seeded synthetic inputs, 6 fixed iterations: every X_s and every coefficient against the oracle."""
    mass = 0.05
    shifts = [0.0, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1.0][:S]
    ctx = bc.Context(dims)
    D = bc.dirac_op(ctx, mass, seed=3)
    B = bc.block_fermion_field(ctx, m).setRandom(seed=4)
    X = [bc.block_fermion_field(ctx, m) for _ in shifts]
    info = bc.SBCGrQ(X, B, D, shifts, 0.0, 0.0, max_iterations=6, trace_limit=6, return_info=True)
    U = orc.fill_gauge(dims, 3)
    Bh = orc.fill_field(m, ctx.V, 4)
    o = orc.sbcgrq(U, dims, mass, Bh, shifts, 0.0, 0.0, max_iterations=6, trace_limit=6)
    assert info["iterations"] == o["iterations"] == 6
    assert rel_err(np.stack([x.download() for x in X]), o["X"]) < 1e-11
    for key in ("alpha", "rho", "delta", "alpha_s", "beta_s"):
        assert rel_err(info["trace"][key], o["trace"][key]) < TOL_COEFF, key


def test_config1_32c4_m8_solves_to_tolerance(bc):
    """BASELINE.json config 1 (V=32^4, m=8, 1 shift): the reference's acceptance test computed with
    device primitives (op, add, -=, hermitian_dot), plus operator properties at full size."""
    dims, m, mass, eps = [32, 32, 32, 32], 8, 0.2, 1e-10
    ctx = bc.Context(dims)
    D = bc.dirac_op(ctx, mass, seed=11)
    B = bc.block_fermion_field(ctx, m).setRandom(seed=12)
    X = [bc.block_fermion_field(ctx, m)]
    it = bc.SBCGrQ(X, B, D, [0.0], eps, max_iterations=2000)
    assert 0 < it < 2000
    AX = bc.block_fermion_field(ctx, m)
    D.op(AX, X[0])
    AX -= B
    r2 = np.real(np.diag(AX.hermitian_dot(AX)))
    b2 = np.real(np.diag(B.hermitian_dot(B)))
    assert np.sqrt(r2 / b2).max() < 2 * eps
    # A is Hermitian positive definite: <B, A X> = <A B, X>, and <B, A B> > 0
    AB = bc.block_fermion_field(ctx, m)
    D.op(AB, B)
    lhs = B.hermitian_dot(AX.rescale_add(1.0, B, 1.0))  # AX was A X - B; add B back -> A X
    rhs = AB.hermitian_dot(X[0])
    assert rel_err(lhs, rhs) < 1e-10
    assert np.all(np.real(np.diag(B.hermitian_dot(AB))) > 0)
    # thinQR leaves orthonormal columns
    Q = B.copy()
    R = Q.thinQR()
    assert rel_err(Q.hermitian_dot(Q), np.eye(m)) < 1e-12
    assert np.allclose(np.tril(R, -1), 0)


def test_config2_64c4_m16_4shifts_solves_to_tolerance(bc):
    """BASELINE.json config 2 (V=64^4, m=16, 4 shifts): full-size solve, true residual of every shift."""
    dims, m, mass, eps = [64, 64, 64, 64], 16, 0.3, 1e-9
    shifts = [0.0, 1e-6, 1e-4, 1e-2]
    ctx = bc.Context(dims)
    D = bc.dirac_op(ctx, mass, seed=21)
    B = bc.block_fermion_field(ctx, m).setRandom(seed=22)
    X = [bc.block_fermion_field(ctx, m) for _ in shifts]
    it = bc.SBCGrQ(X, B, D, shifts, eps, eps, max_iterations=400)  # bounded: a wrong stencil must fail, not spin
    assert 0 < it < 400
    b2 = np.real(np.diag(B.hermitian_dot(B)))
    AX = bc.block_fermion_field(ctx, m)
    for s, sig in enumerate(shifts):
        D.op(AX, X[s])
        AX.add(X[s], sig)
        AX -= B
        r2 = np.real(np.diag(AX.hermitian_dot(AX)))
        assert np.sqrt(r2 / b2).max() < 2 * eps, s


def test_full_size_capacity_mode_and_tile_classes_agree_with_default(bc, monkeypatch):
    """At BASELINE's full single-GPU size (64^4, m=16, 4 shifts) the three ways the stencil can be launched -- one
    launch, capacity mode (ring of x3 slices) and the interior/boundary classes of the split exchange (forced on an
    undivided lattice) -- must give the same iteration count and solutions (size-independent property: the operator
    is the same linear map)."""
    dims, m, mass, eps = [64, 64, 64, 64], 16, 0.5, 1e-8
    shifts = [0.0, 1e-2]
    got = {}
    for mode in ("default", "capacity", "classes"):
        monkeypatch.setenv("BCG_FORCE_TILE_CLASSES", "1" if mode == "classes" else "0")
        ctx = bc.Context(dims)
        if mode == "capacity":
            ctx.capacity_mode(8)
        D = bc.dirac_op(ctx, mass, seed=31)
        B = bc.block_fermion_field(ctx, m).setRandom(seed=32)
        X = [bc.block_fermion_field(ctx, m) for _ in shifts]
        it = bc.SBCGrQ(X, B, D, shifts, eps, eps, max_iterations=200)
        assert 0 < it < 200
        res = bc.true_residuals(X, B, D, shifts)
        assert res.max() < 2 * eps
        # a cheap fingerprint of the solutions: their Gram matrices with the source
        got[mode] = (it, [B.hermitian_dot(x) for x in X])
        del X, B, D, ctx
    for mode in ("capacity", "classes"):
        assert got[mode][0] == got["default"][0]
        for a, b in zip(got[mode][1], got["default"][1]):
            assert rel_err(a, b) < 1e-9


def test_error_behaviour(bc):
    ctx = bc.Context([16])
    D = bc.dirac_op(ctx, 0.5, seed=1)
    B = bc.block_fermion_field(ctx, 3).setRandom(seed=2)
    X = [bc.block_fermion_field(ctx, 3) for _ in range(2)]
    with pytest.raises(bc.BlockCGError) as e:  # unsorted shifts, inc/block_solvers.hpp:100-101
        bc.SBCGrQ(X, B, D, [0.1, 0.0], 1e-10)
    assert e.value.code == 1
    with pytest.raises(bc.BlockCGError):  # negative shift, :99
        bc.SBCGrQ(X, B, D, [-0.1, 0.0], 1e-10)
    with pytest.raises(bc.BlockCGError) as e:  # width out of range (1 <= m <= 32)
        bc.block_fermion_field(ctx, 33)
    assert e.value.code == 2
    # CholQR breakdown is reported, not silently NaN (a zero block has a singular Gram matrix)
    Z = bc.block_fermion_field(ctx, 3).setZero()
    with pytest.raises(bc.BlockCGError) as e:
        Z.thinQR()
    assert e.value.code == 6
    # non-convergence = return value equals max_iterations
    assert bc.SBCGrQ(X, B, D, [0.0, 0.1], 1e-300, 1e-300, max_iterations=3) == 3


@pytest.mark.parametrize("m", [16, 32, 8])
@pytest.mark.parametrize("dims", [[5, 3, 7], [16, 4, 4, 6], [37]])
def test_mfma_fast_path_matches_generic_and_oracle(bc, orc, m, dims):
    """Fused MFMA kernels against the generic VALU kernels and the oracle, including volumes whose row
    count is not a multiple of the 16-row MFMA tile and the cache-blocked stencil walk."""
    V = int(np.prod(dims))
    U = orc.fill_gauge(dims, 31)
    Bh = orc.fill_field(m, V, 32)
    Yh = orc.fill_field(m, V, 33)
    rng = np.random.default_rng(m)
    M = rng.uniform(-1, 1, (m, m)) + 1j * rng.uniform(-1, 1, (m, m))
    results = {}
    for mode in ("fast", "generic"):
        ctx = bc.Context(dims)
        ctx.force_generic(mode == "generic")
        D = bc.dirac_op(ctx, 0.1, U=U)
        F = lambda a: bc.block_fermion_field(ctx, m, a)  # noqa: E731
        out = bc.block_fermion_field(ctx, m)
        D.op(out, F(Bh))
        hop = bc.block_fermion_field(ctx, m)
        D.D(hop, F(Bh))
        q = F(Yh)
        R = q.thinQR()
        results[mode] = dict(op=out.download(), hop=hop.download(), add=F(Yh).add(F(Bh), M).download(),
                             xpay=F(Yh).rescale_add(M, F(Bh), 0.7).download(), dot=F(Yh).hermitian_dot(F(Bh)),
                             Q=q.download(), R=R)
    oracle_vals = dict(op=orc.dirac_apply(U, dims, 0.1, Bh), hop=orc.hop(U, dims, Bh), add=orc.add_matrix(Yh, Bh, M),
                       xpay=orc.rescale_add_matrix(Yh, M, Bh, 0.7), dot=orc.hermitian_dot(Yh, Bh))
    oracle_vals["Q"], oracle_vals["R"] = orc.thin_qr(Yh)
    for k, v in oracle_vals.items():
        assert rel_err(results["fast"][k], v) < 2e-13, ("fast", k)
        assert rel_err(results["generic"][k], v) < 2e-13, ("generic", k)


@pytest.mark.parametrize("c2", ["0", "2"])
def test_cache_blocked_stencil_walk(bc, orc, c2, monkeypatch):
    monkeypatch.setenv("BCG_HOP_C2", c2)
    dims, m = [16, 4, 4, 6], 16
    V = int(np.prod(dims))
    U = orc.fill_gauge(dims, 41)
    Bh = orc.fill_field(m, V, 42)
    ctx = bc.Context(dims)
    D = bc.dirac_op(ctx, 0.2, U=U)
    out = bc.block_fermion_field(ctx, m)
    D.op(out, bc.block_fermion_field(ctx, m, Bh))
    assert rel_err(out.download(), orc.dirac_apply(U, dims, 0.2, Bh)) < TOL_KERNEL


@pytest.mark.parametrize("m,dims", [(16, [32, 4, 2, 6]), (16, [16, 2, 2, 2]), (8, [32, 2, 4, 2]), (32, [8, 4, 4, 2]), (32, [16, 2, 2, 4])])
@pytest.mark.parametrize("walk,blocks", [("0", "768"), ("3", "8"), ("0", "3")])
def test_specialised_4d_stencil(bc, orc, m, dims, walk, blocks, monkeypatch):
    """k_hop4 (tile = consecutive x0 sites, double-buffered link staging) in every walk order, with more
    tiles than blocks so that the prefetch pipeline wraps, against the oracle's hop and op."""
    monkeypatch.setenv("BCG_HOP_WALK", walk)
    monkeypatch.setenv("BCG_HOP_BLOCKS", blocks)
    monkeypatch.setenv("BCG_HOP_PATCH", f"{4 * (64 // m)},2,2")
    V = int(np.prod(dims))
    U = orc.fill_gauge(dims, 51)
    Bh = orc.fill_field(m, V, 52)
    ctx = bc.Context(dims)
    D = bc.dirac_op(ctx, 0.3, U=U)
    x = bc.block_fermion_field(ctx, m, Bh)
    out = bc.block_fermion_field(ctx, m)
    D.D(out, x)
    assert rel_err(out.download(), orc.hop(U, dims, Bh)) < TOL_KERNEL
    D.op(out, x)
    assert rel_err(out.download(), orc.dirac_apply(U, dims, 0.3, Bh)) < TOL_KERNEL
    if m == 16:  # fused Gram variant of the second hop (solver path): two iterations against the oracle
        X = [bc.block_fermion_field(ctx, m)]
        info = bc.SBCGrQ(X, x, D, [0.01], 0.0, 0.0, max_iterations=2, trace_limit=2, return_info=True)
        o = orc.sbcgrq(U, dims, 0.3, Bh, [0.01], 0.0, 0.0, max_iterations=2, trace_limit=2)
        assert rel_err(info["trace"]["alpha"], o["trace"]["alpha"]) < TOL_COEFF
        assert rel_err(X[0].download(), o["X"][0]) < 1e-11


@pytest.mark.parametrize("blocks", ["32", "64"])
def test_stencil_x3_carry_path(bc, orc, blocks, monkeypatch):
    """Per-XCD patch walk with as many blocks per class as tiles per patch slice: every block then visits
    (tile, x3), (tile, x3+1), ... and takes its -x3 neighbours from the LDS ring and U_3(x-3) from the previous
    link image (kernels_stencil.hip, k_hop4) instead of global memory."""
    monkeypatch.setenv("BCG_HOP_WALK", "3")
    monkeypatch.setenv("BCG_HOP_BLOCKS", blocks)
    monkeypatch.setenv("BCG_HOP_PATCH", "16,2,2")
    dims, m = [16, 4, 4, 16], 16
    V = int(np.prod(dims))
    U = orc.fill_gauge(dims, 61)
    Bh = orc.fill_field(m, V, 62)
    ctx = bc.Context(dims)
    D = bc.dirac_op(ctx, 0.2, U=U)
    x = bc.block_fermion_field(ctx, m, Bh)
    out = bc.block_fermion_field(ctx, m)
    for _ in range(3):
        D.D(out, x)
        assert rel_err(out.download(), orc.hop(U, dims, Bh)) < TOL_KERNEL
    D.op(out, x)
    assert rel_err(out.download(), orc.dirac_apply(U, dims, 0.2, Bh)) < TOL_KERNEL


def test_other_solvers_match_reference_fixture(bc, orc):
    """CG, SCG, BCG, BCGrQ and the on-device true-residual check, against values the
    unmodified reference computed at its own test configuration (test/solvers.cpp:19-91)."""
    g = np.load(golden_files("ref1d_v128_other_solvers.npz")[0])
    dims, mass, eps, shifts = _dims(g), float(g["mass"]), float(g["eps"]), list(g["shifts"])
    ctx = bc.Context(dims)
    D = bc.dirac_op(ctx, mass, U=g["U"])
    b = bc.block_fermion_field(ctx, 1, g["b"])
    B = bc.block_fermion_field(ctx, 3, g["B"])
    x = bc.block_fermion_field(ctx, 1)
    it = bc.CG(x, b, D, eps)
    assert abs(it - int(g["it_cg"])) <= 1 and rel_err(x.download(), g["x_cg"]) < TOL_SOLUTION
    assert bc.true_residuals([x], b, D, [0.0]).max() < 2 * eps
    xs = [bc.block_fermion_field(ctx, 1) for _ in shifts]
    it = bc.SCG(xs, b, D, shifts, eps)
    assert abs(it - int(g["it_scg"])) <= 1
    assert rel_err(np.stack([v.download() for v in xs]), g["x_scg"]) < TOL_SOLUTION
    res = bc.true_residuals(xs, b, D, shifts)
    assert res.max() < 2 * eps  # REQUIRE(residual < 2 * stopping_criterion), test/solvers.cpp:50
    assert rel_err(res, orc.true_residuals(g["U"], dims, mass, g["b"], shifts, np.stack([v.download() for v in xs]))) < 1e-3
    X = bc.block_fermion_field(ctx, 3)
    it = bc.BCG(X, B, D, eps)
    assert abs(it - int(g["it_bcg"])) <= 1 and rel_err(X.download(), g["X_bcg"]) < TOL_SOLUTION
    assert bc.true_residuals([X], B, D, [0.0]).max() < 2 * eps
    it = bc.BCGrQ(X, B, D, eps)
    assert abs(it - int(g["it_bcgrq"])) <= 1 and rel_err(X.download(), g["X_bcgrq"]) < TOL_SOLUTION
    assert bc.true_residuals([X], B, D, [0.0]).max() < 2 * eps


def test_other_solvers_on_4d_lattice_mfma_width(bc, orc):
    """BCG / BCGrQ at block width 16 on a 4-D lattice against the oracle (seeded inputs)."""
    dims, m, mass, eps = [16, 4, 4, 4], 16, 0.4, 1e-10
    V = int(np.prod(dims))
    U = orc.fill_gauge(dims, 71)
    Bh = orc.fill_field(m, V, 72)
    ctx = bc.Context(dims)
    D = bc.dirac_op(ctx, mass, U=U)
    B = bc.block_fermion_field(ctx, m, Bh)
    X = bc.block_fermion_field(ctx, m)
    for with_qr, fn in ((False, bc.BCG), (True, bc.BCGrQ)):
        it = fn(X, B, D, eps)
        Xo, ito = orc.bcg(U, dims, mass, Bh, eps, with_qr=with_qr)
        assert abs(it - ito) <= 1
        assert bc.true_residuals([X], B, D, [0.0]).max() < 2 * eps
        if it == ito:
            assert rel_err(X.download(), Xo) < TOL_SOLUTION


def test_shift_retirement_matches_oracle(bc, orc):
    """Shifts retire while the base system still iterates (eps_shifts >> eps): the reference decrements a counter
    and stops updating the largest active shift (inc/block_solvers.hpp:161,179-181).  Same retirement iterations,
    same frozen solutions."""
    dims, m, mass = [12, 6, 4], 4, 0.05
    shifts, eps, eps_s = [0.0, 0.3, 2.0, 9.0], 1e-10, 1e-4
    V = int(np.prod(dims))
    U = orc.fill_gauge(dims, 81)
    Bh = orc.fill_field(m, V, 82)
    ctx = bc.Context(dims)
    D = bc.dirac_op(ctx, mass, U=U)
    B = bc.block_fermion_field(ctx, m, Bh)
    X = [bc.block_fermion_field(ctx, m) for _ in shifts]
    info = bc.SBCGrQ(X, B, D, shifts, eps, eps_s, trace_limit=400, return_info=True)
    o = orc.sbcgrq(U, dims, mass, Bh, shifts, eps, eps_s, trace_limit=400)
    assert abs(info["iterations"] - o["iterations"]) <= 1
    n = min(info["iterations"], o["iterations"])
    visited_gpu = info["trace"]["residual_shift"][:n] >= 0
    visited_cpu = o["trace"]["residual_shift"][:n] >= 0
    assert np.array_equal(visited_gpu, visited_cpu)          # every shift retires at the same iteration
    assert not visited_cpu[-1, 1:].all()                     # and at least one did retire before the end
    assert rel_err(np.stack([x.download() for x in X])[1:], o["X"][1:]) < 1e-9
    res = bc.true_residuals(X, B, D, shifts)
    assert res[0].max() < 2 * eps and res[1:].max() < 50 * eps_s


@pytest.mark.parametrize("dims,m", [([1], 1), ([2], 3), ([3, 1, 2], 2), ([1, 1, 1, 1], 16), ([16, 1, 1, 2], 16), ([2, 2, 2, 2], 8)])
def test_degenerate_lattices(bc, orc, dims, m):
    """Extents of 1 and 2 (x+mu and x-mu coincide or are the site itself), single-site lattices, and
    max_iterations = 0."""
    V = int(np.prod(dims))
    U = orc.fill_gauge(dims, 91)
    Bh = orc.fill_field(m, V, 92)
    ctx = bc.Context(dims)
    D = bc.dirac_op(ctx, 0.7, U=U)
    B = bc.block_fermion_field(ctx, m, Bh)
    out = bc.block_fermion_field(ctx, m)
    D.op(out, B)
    assert rel_err(out.download(), orc.dirac_apply(U, dims, 0.7, Bh)) < TOL_KERNEL
    X = [bc.block_fermion_field(ctx, m).setRandom(seed=5)]
    if 3 * V < m:  # more columns than rows: B cannot have full column rank, the initial CholQR (:115) breaks down.
        with pytest.raises(bc.BlockCGError) as e:  # reported here; the reference would carry NaN
            bc.SBCGrQ(X, B, D, [0.25], 1e-10, max_iterations=0)
        assert e.value.code == 6
        return
    assert bc.SBCGrQ(X, B, D, [0.25], 1e-10, max_iterations=0) == 0
    assert not X[0].download().any()                         # X is zeroed even when no iteration runs (:111-113)
    it = bc.SBCGrQ(X, B, D, [0.25], 1e-10, max_iterations=200)
    assert it <= 200 and bc.true_residuals(X, B, D, [0.25]).max() < 2e-10


@pytest.mark.parametrize("m,dims,ring", [(16, [32, 4, 4, 12], 3), (16, [32, 4, 4, 12], 4), (16, [16, 8, 4, 6], 6),
                                         (16, [64, 4, 2, 8], 4), (8, [32, 4, 4, 8], 4), (32, [16, 4, 4, 6], 3)])
@pytest.mark.parametrize("walk,blocks", [("3", "8"), ("0", "768")])
def test_capacity_mode_matches_default_and_oracle(bc, orc, m, dims, ring, walk, blocks, monkeypatch):
    """bcg_capacity_mode: dirac_op::op's intermediate field kept as a ring of x3 slices (the second stencil runs ring-2
    slices behind the first, slices L3-1 and 0 computed twice).  Operator, fused Gram product and a fixed number of
    SBCGrQ iterations against the whole-field mode and the oracle."""
    monkeypatch.setenv("BCG_HOP_WALK", walk)
    monkeypatch.setenv("BCG_HOP_BLOCKS", blocks)
    monkeypatch.setenv("BCG_HOP_PATCH", "16,2,2")
    V = int(np.prod(dims))
    U = orc.fill_gauge(dims, 61)
    Bh = orc.fill_field(m, V, 62)
    shifts, iters = [0.0, 0.02, 0.3], 5
    res = {}
    for mode in (0, ring):
        ctx = bc.Context(dims)
        ctx.capacity_mode(mode)
        D = bc.dirac_op(ctx, 0.2, U=U)
        B = bc.block_fermion_field(ctx, m, Bh)
        out = bc.block_fermion_field(ctx, m)
        D.op(out, B)
        X = [bc.block_fermion_field(ctx, m) for _ in shifts]
        info = bc.SBCGrQ(X, B, D, shifts, 0.0, 0.0, max_iterations=iters, trace_limit=iters, return_info=True)
        res[mode] = (out.download(), [x.download() for x in X], info["trace"], ctx.sbcgrq_device_bytes(m, len(shifts)))
    assert np.array_equal(res[0][0], res[ring][0])  # the operator alone: same arithmetic per site, bit for bit
    assert rel_err(res[ring][0], orc.dirac_apply(U, dims, 0.2, Bh)) < TOL_KERNEL
    o = orc.sbcgrq(U, dims, 0.2, Bh, shifts, 0.0, 0.0, max_iterations=iters, trace_limit=iters)
    for key in ("alpha", "rho", "delta", "alpha_s", "beta_s"):
        assert rel_err(res[ring][2][key], res[0][2][key]) < 1e-11, key
        assert rel_err(res[ring][2][key], o["trace"][key]) < TOL_COEFF, key
    for s in range(len(shifts)):
        assert rel_err(res[ring][1][s], res[0][1][s]) < 1e-11
        assert rel_err(res[ring][1][s], o["X"][s]) < 1e-10
    field = V * 3 * m * 16
    # what the mode is for: the ring instead of the intermediate field, and none of the two further residual buffers of
    # the shift updates grouped over four iterations (pair_shifts_depth: m = 8, 16; capacity mode groups two, for free)
    # nor the spare P_0 of the deferred X_0 update (DeferredX0: one more field outside capacity mode)
    # (at m = 16 the ring sweep keeps the fused product's block partials of all its chunks side by side: 1024 blocks x chunks)
    chunks = -(-dims[3] // (ring - 2))
    more_partials = max(0, 1024 * chunks * m * m * 16 - 2048 * 32 * 32 * 16) if m == 16 else 0
    assert res[0][3] - res[ring][3] == field - field // dims[3] * ring + (3 * field if m in (8, 16) else 0) - more_partials


def test_capacity_mode_arguments(bc):
    ctx = bc.Context([16, 4, 4, 12])
    for bad in (1, 2, 5, 24):
        with pytest.raises(bc.BlockCGError) as e:
            ctx.capacity_mode(bad)
        assert e.value.code == 1
    ctx.capacity_mode(12)
    ctx.capacity_mode(0)
    with pytest.raises(bc.BlockCGError) as e:
        bc.Context([16, 4, 12]).capacity_mode(3)
    assert e.value.code == 2
    # widths outside the specialised stencil keep the whole intermediate field and still solve
    ctx = bc.Context([8, 4, 4, 6])
    ctx.capacity_mode(3)
    D = bc.dirac_op(ctx, 0.5, seed=1)
    B = bc.block_fermion_field(ctx, 4).setRandom(seed=2)
    X = [bc.block_fermion_field(ctx, 4)]
    bc.SBCGrQ(X, B, D, [0.1], 1e-10, max_iterations=300)
    assert bc.true_residuals(X, B, D, [0.1]).max() < 2e-10


@pytest.mark.parametrize("m,dims,patch,blocks", [(16, [32, 8, 8, 6], "16,2,2", "32"), (16, [64, 4, 8, 4], "32,2,2", "64"),
                                                 (8, [64, 8, 8, 4], "32,2,2", "32"), (32, [16, 8, 8, 4], "8,2,2", "32")])
@pytest.mark.parametrize("sync", ["0", "1", "4"])
@pytest.mark.parametrize("bundle", ["0", "1", "2"])
def test_column_sweep_stencil(bc, orc, m, dims, patch, blocks, sync, bundle, monkeypatch):
    """The column sweep in both forms: k_hop4c (one block per row of a patch slice, scalar row pointers; bundle = 0) and
    k_hop4b (one block per 2 x 2 bundle of columns, rows shared through LDS; bundle = 1, the default, uses it at m = 8 for
    the plain hop only, bundle = 2 for every launch), the
    blocks of an XCD class paced by device counters (window 1 forces waits, 0 switches pacing off).  Operator, fused Gram
    product, a fixed number of iterations and capacity mode against the oracle; the profile shows which form ran."""
    monkeypatch.setenv("BCG_HOP_PATCH", patch)
    monkeypatch.setenv("BCG_HOP_BLOCKS", blocks)
    monkeypatch.setenv("BCG_HOP_SYNC", sync)
    monkeypatch.setenv("BCG_HOP_BUNDLE", bundle)
    monkeypatch.setenv("BCG_HOP_BUNDLE_SYNC", sync if sync == "0" else "-" + sync)  # < 0: pace short sweeps too
    V = int(np.prod(dims))
    U = orc.fill_gauge(dims, 71)
    Bh = orc.fill_field(m, V, 72)
    shifts, iters = [0.0, 0.05], 4
    o = orc.sbcgrq(U, dims, 0.3, Bh, shifts, 0.0, 0.0, max_iterations=iters, trace_limit=iters)
    ref_op = orc.dirac_apply(U, dims, 0.3, Bh)
    for ring in (0, 2 if dims[3] == 4 else 3):
        if ring == 2:
            ring = 4  # ring = L3
        ctx = bc.Context(dims)
        ctx.capacity_mode(ring)
        ctx.profiling(True)
        D = bc.dirac_op(ctx, 0.3, U=U)
        B = bc.block_fermion_field(ctx, m, Bh)
        out = bc.block_fermion_field(ctx, m)
        D.op(out, B)
        assert rel_err(out.download(), ref_op) < TOL_KERNEL
        X = [bc.block_fermion_field(ctx, m) for _ in shifts]
        info = bc.SBCGrQ(X, B, D, shifts, 0.0, 0.0, max_iterations=iters, trace_limit=iters, return_info=True)
        for key in ("alpha", "rho", "delta", "alpha_s", "beta_s"):
            assert rel_err(info["trace"][key], o["trace"][key]) < TOL_COEFF, key
        for s in range(len(shifts)):
            assert rel_err(X[s].download(), o["X"][s]) < 1e-10
        prof = ctx.profile()
        nb, nc = (prof.get(k, {}).get("count", 0) for k in ("stencil_form_k_hop4b", "stencil_form_k_hop4c"))
        if bundle == "0" or (bundle == "1" and ring):  # (default tuning keeps the row form for ring windows under 10 slices)
            assert nc > 0 and nb == 0, prof.keys()
        elif bundle == "1" and m == 8:
            assert nb > 0 and nc > 0, prof.keys()  # plain hop on bundles, the hop with the Gram product on rows
        else:
            assert nb > 0 and nc == 0, prof.keys()
        assert "stencil_form_k_hop4" not in prof and "stencil_form_general" not in prof


@pytest.mark.parametrize("m,dims,patch,blocks", [(16, [32, 8, 8, 6], "16,2,2", "32"), (32, [16, 8, 8, 4], "8,2,2", "32")])
def test_super_patch_tile_order(bc, orc, m, dims, patch, blocks, monkeypatch):
    """BCG_HOP_SUPER=1 (HopWalk::super; the round-5 traffic experiment, profiles/r05_stencil_super_patch.txt): the patches the
    eight XCD classes sweep at the same time form a 2 x 2 x 2 super-patch.  Tile order only: the operator is bit-identical
    to the default order and matches the oracle; the fused Gram product sums the same terms in another block order."""
    monkeypatch.setenv("BCG_HOP_PATCH", patch)
    monkeypatch.setenv("BCG_HOP_BLOCKS", blocks)
    U = orc.fill_gauge(dims, 71)
    Bh = orc.fill_field(m, int(np.prod(dims)), 72)
    outs = {}
    for sup in ("0", "1"):
        monkeypatch.setenv("BCG_HOP_SUPER", sup)
        ctx = bc.Context(dims)
        ctx.profiling(True)
        D = bc.dirac_op(ctx, 0.3, U=U)
        B = bc.block_fermion_field(ctx, m, Bh)
        out = bc.block_fermion_field(ctx, m)
        D.op(out, B)
        X = [bc.block_fermion_field(ctx, m)]
        info = bc.SBCGrQ(X, B, D, [0.0], 0.0, 0.0, max_iterations=3, trace_limit=3, return_info=True)
        assert ctx.profile().get("stencil_form_k_hop4b", {}).get("count", 0) > 0
        outs[sup] = (out.download(), info["trace"]["alpha"], X[0].download())
    assert np.array_equal(outs["0"][0], outs["1"][0])
    assert rel_err(outs["1"][0], orc.dirac_apply(U, dims, 0.3, Bh)) < TOL_KERNEL
    assert rel_err(outs["1"][1], outs["0"][1]) < 1e-12 and rel_err(outs["1"][2], outs["0"][2]) < 1e-11


def test_fused_true_residual_check(bc, orc, monkeypatch):
    """bcg_true_residuals (test/solvers.cpp:104-116, benchmark.cpp:93-103) in its one-pass form -- second stencil, `-= B`
    and the Gram product fused in the bundle kernel, AX never written (m = 16) -- against the oracle and against the unfused
    sequence of primitives, on unconverged iterates so that the residuals are O(1e-2)."""
    dims, m, mass, shifts = [16, 8, 8, 8], 16, 0.2, [0.0, 0.01, 0.3]
    monkeypatch.setenv("BCG_HOP_PATCH", "16,2,2")  # a lattice this small needs small patches for the column / bundle sweep
    monkeypatch.setenv("BCG_HOP_BLOCKS", "32")
    V = int(np.prod(dims))
    U = orc.fill_gauge(dims, 91)
    Bh = orc.fill_field(m, V, 92)
    got = {}
    for bundle in ("1", "0"):
        monkeypatch.setenv("BCG_HOP_BUNDLE", bundle)
        ctx = bc.Context(dims)
        ctx.profiling(True)
        D = bc.dirac_op(ctx, mass, U=U)
        B = bc.block_fermion_field(ctx, m, Bh)
        X = [bc.block_fermion_field(ctx, m) for _ in shifts]
        bc.SBCGrQ(X, B, D, shifts, 0.0, 0.0, max_iterations=6)
        got[bundle] = (bc.true_residuals(X, B, D, shifts), np.stack([x.download() for x in X]), ctx.profile())
    assert got["1"][2].get("hop_residual", {}).get("count", 0) == len(shifts)   # the fused form ran ...
    assert "hop_residual" not in got["0"][2]                                      # ... and the primitives' form here
    want = orc.true_residuals(U, dims, mass, Bh, shifts, got["1"][1])
    assert 1e-6 < want.max() < 1.0
    assert rel_err(got["1"][0], want) < 1e-11
    assert rel_err(got["0"][0], want) < 1e-11


@pytest.mark.parametrize("m,dims", [(16, [16, 8, 8, 8]), (8, [16, 8, 4, 8])], ids=["m16", "m8"])
def test_deferred_normalisation_of_q_is_bit_identical(bc, orc, m, dims, monkeypatch):
    """Phase C keeps Q rho^-1 in registers and leaves the un-normalised Q in memory; the next phase B applies the same
    rho^-1 with the same kernel arithmetic (capi_solvers.hip: lazy_q_width): the iterates must equal, bit for bit, those of
    the form that stores Q rho^-1 and reads it back (BCG_LAZY_Q=0), and match the oracle."""
    monkeypatch.setenv("BCG_HOP_PATCH", "16,2,2")
    monkeypatch.setenv("BCG_PAIR_SHIFTS", "0")  # the paired updates build on this and have their own test below
    shifts, iters, mass = [0.0, 1e-3, 0.1], 6, 0.2
    outs = []
    for lazy in ("1", "0"):
        monkeypatch.setenv("BCG_LAZY_Q", lazy)
        ctx = bc.Context(dims)
        ctx.profiling(True)
        D = bc.dirac_op(ctx, mass, seed=51)
        B = bc.block_fermion_field(ctx, m).setRandom(seed=52)
        X = [bc.block_fermion_field(ctx, m) for _ in shifts]
        info = bc.SBCGrQ(X, B, D, shifts, 0.0, 0.0, max_iterations=iters, trace_limit=iters, return_info=True)
        outs.append(([x.download() for x in X], info["trace"], ctx.profile()))
    for s in range(len(shifts)):
        assert np.array_equal(outs[0][0][s], outs[1][0][s])
    for key in ("alpha", "rho", "delta", "alpha_s", "beta_s"):
        assert np.array_equal(outs[0][1][key], outs[1][1][key]), key
    # one field pass less in phase C: (1 + 4 S) s against (2 + 4 S) s of algorithmic bytes per launch
    S = len(shifts)
    assert outs[0][2]["phaseC"]["bytes"] * (2 + 4 * S) == pytest.approx(outs[1][2]["phaseC"]["bytes"] * (1 + 4 * S), rel=1e-9)
    U = orc.fill_gauge(dims, 51)
    Bh = orc.fill_field(m, int(np.prod(dims)), 52)
    o = orc.sbcgrq(U, dims, mass, Bh, shifts, 0.0, 0.0, max_iterations=iters, trace_limit=iters)
    for s in range(len(shifts)):
        assert rel_err(outs[0][0][s], o["X"][s]) < 1e-10


@pytest.mark.parametrize("m,dims,depth", [(16, [16, 8, 8, 8], 2), (16, [16, 8, 8, 8], 3), (16, [16, 8, 8, 8], 4), (8, [16, 8, 4, 8], 2),
                                          (8, [16, 8, 4, 8], 3), (8, [16, 8, 4, 8], 4), (32, [16, 4, 4, 6], 2)],
                         ids=["m16-2", "m16-3", "m16-4", "m8-2", "m8-3", "m8-4", "m32-2"])
@pytest.mark.parametrize("defer_x0", [0, 1], ids=["x0-every-iteration", "x0-deferred"])
def test_grouped_shift_updates_are_bit_identical(bc, orc, m, dims, depth, defer_x0, monkeypatch):
    """The shifts >= 1 are updated `depth` iterations at a time (capi_solvers.hip: pair_shifts_depth, k_phaseC_multi).
    After any number of iterations -- a multiple of the depth or not, run in one call or in pieces, with shifts leaving the
    active set on the way -- X and the residual must equal, bit for bit, those of the solver that updates every shift in
    every iteration (BCG_PAIR_SHIFTS=0), and match the oracle.
    x0-deferred (the default at m = 8, 16: DeferredX0 in capi_solvers.hip): X_0's updates wait for the closing pass as
    well, composed onto the group's first P_0 -- the P_0 sequence, the residual, every X_s with s >= 1 stay bit-identical,
    X_0 agrees to rounding (1e-13), and a group moves 3 (g - 1) + g + 4 S + 1 field passes."""
    monkeypatch.setenv("BCG_HOP_PATCH", "16,2,2")
    monkeypatch.setenv("BCG_DEFER_X0", str(defer_x0))
    deferred = bool(defer_x0) and m != 32
    shifts, mass = [0.0, 1e-3, 0.1, 2.0], 0.2
    S = len(shifts)
    U = orc.fill_gauge(dims, 61)
    Bh = orc.fill_field(m, int(np.prod(dims)), 62)

    def run(pair, pieces, eps_shifts):
        monkeypatch.setenv("BCG_PAIR_SHIFTS", str(pair))
        ctx = bc.Context(dims)
        ctx.profiling(True)
        D = bc.dirac_op(ctx, mass, U=U)
        B = bc.block_fermion_field(ctx, m, Bh)
        X = [bc.block_fermion_field(ctx, m) for _ in shifts]
        st = bc.SBCGrQState(X, B, D, shifts, 0.0, eps_shifts)
        for n in pieces:
            st.iterate(n)
        res = st.residual
        st.end()
        return [x.download() for x in X], res, ctx.profile()

    for pieces, eps_shifts in (([12], 0.0), ([7], 0.0), ([2, 3, 1, 4, 5], 0.0), ([1, 1, 1], 0.0), ([9], 3e-2), ([4, 5], 0.5)):
        a, ra, pa = run(depth, pieces, eps_shifts)
        b, rb, pb = run(0, pieces, eps_shifts)
        assert ra == rb
        for s in range(S):
            if s == 0 and deferred:
                assert rel_err(a[0], b[0]) < 1e-13, (pieces, eps_shifts)
            else:
                assert np.array_equal(a[s], b[s]), (pieces, eps_shifts, s)
        assert not any(k.startswith("phaseC_multi") for k in pb) and "phaseC_p0" not in pb
        assert ("phaseC_p0" in pa) == (deferred and max(pieces) >= 2)
        if eps_shifts == 0.0:
            # every call runs its iterations in groups of `depth`, the rest as one smaller group (or a plain iteration)
            groups = [g for n in pieces for g in [depth] * (n // depth) + [n % depth] if g > 0]
            multi = [g for g in groups if g >= 2]
            for g in (2, 3, 4):  # one profile entry per group size
                assert m == 32 or pa.get(f"phaseC_multi{g}", {}).get("count", 0) == multi.count(g), (pieces, g)
            if m == 32:  # the stored blocks are normalised there, and a closing pass takes two launches for three shifts
                assert pa.get("phaseC_multi2", {}).get("count", 0) == 2 * len(multi)
            if multi and m != 32:
                # a group of g iterations moves 5 (g - 1) + g + 4 S field passes where g plain ones move g (1 + 4 S)
                per_pass = pb["phaseC"]["bytes"] / (sum(pieces) * (1 + 4 * S))
                moved = sum(v["bytes"] for k, v in pa.items() if k.startswith("phaseC"))
                inner = 3 if deferred else 5  # shift 0 inside a group: Q, P_0 read, P_0 written [+ X_0 read and written]
                want = sum(inner * (g - 1) + g + 4 * S + (1 if deferred else 0) if g >= 2 else 1 + 4 * S for g in groups)
                assert moved == pytest.approx(per_pass * want, rel=1e-9)
    # shifts did leave the active set in the runs with eps_shifts > 0 (otherwise those runs test nothing new)
    o = orc.sbcgrq(U, dims, mass, Bh, shifts, 0.0, 3e-2, max_iterations=9, trace_limit=9)
    visited = o["trace"]["residual_shift"][:, 1:] >= 0.0  # -1 = the shift was not updated in that iteration
    assert visited[0].all() and not visited[-1].all()
    a, _, _ = run(depth, [9], 3e-2)
    for s in range(S):
        assert rel_err(a[s], o["X"][s]) < 1e-10


@pytest.mark.parametrize("m,dims", [(16, [16, 8, 8, 8]), (8, [16, 8, 4, 8])], ids=["m16", "m8"])
def test_single_system_groups_for_the_deferred_x0_update(bc, orc, m, dims, monkeypatch):
    """With shift 0 alone -- a one-shift solve, or the tail of a solve whose shifted systems have converged -- there are no
    shifted updates to group, but X_0's can still wait (DeferredX0): groups of three or four iterations move 3 s inside and
    q_0..q_{g-1}, P_0, X_0, the group's first P_0 read + P_0, X_0 written in the closing pass, against 5 s per iteration
    (capi_solvers.hip: pair_shifts_depth, x0_may_wait).  A group of two would save nothing and is not opened.  Residual and
    P_0 sequence bit-identical to the plain solver, X_0 to rounding; the oracle agrees."""
    monkeypatch.setenv("BCG_HOP_PATCH", "16,2,2")
    mass = 0.2
    U = orc.fill_gauge(dims, 63)
    Bh = orc.fill_field(m, int(np.prod(dims)), 64)

    def run(shifts, pair, defer, pieces, eps_shifts=0.0):
        monkeypatch.setenv("BCG_PAIR_SHIFTS", str(pair))
        monkeypatch.setenv("BCG_DEFER_X0", str(defer))
        ctx = bc.Context(dims)
        ctx.profiling(True)
        D = bc.dirac_op(ctx, mass, U=U)
        B = bc.block_fermion_field(ctx, m, Bh)
        X = [bc.block_fermion_field(ctx, m) for _ in shifts]
        st = bc.SBCGrQState(X, B, D, shifts, 0.0, eps_shifts)
        for n in pieces:
            st.iterate(n)
        res, planned = st.residual, ctx.sbcgrq_device_bytes(m, len(shifts))
        st.end()
        return [x.download() for x in X], res, ctx.profile(), planned

    field = int(np.prod(dims)) * 3 * m * 16
    b, rb, pb, mem_plain = run([0.0], 0, 1, [12])
    assert set(k for k in pb if k.startswith("phaseC")) == {"phaseC"}
    per_pass = pb["phaseC"]["bytes"] / (12 * 5)
    for depth, pieces in ((4, [12]), (3, [12]), (4, [7]), (4, [2, 3, 1, 4, 5]), (4, [1, 1, 1])):
        a, ra, pa, mem = run([0.0], depth, 1, pieces)
        ref, rref, _, _ = (b, rb, pb, 0) if sum(pieces) == 12 else run([0.0], 0, 1, pieces)
        assert ra == rref and rel_err(a[0], ref[0]) < 1e-13, (depth, pieces)
        assert mem == mem_plain + (depth - 2 + 1) * field  # the further residual buffers and the spare P_0
        groups = [g for n in pieces for g in [depth] * (n // depth) + [n % depth] if g > 0]
        for g in (2, 3, 4):
            assert pa.get(f"phaseC_multi{g}", {}).get("count", 0) == groups.count(g), (depth, pieces, g)
        assert pa.get("phaseC_p0", {}).get("count", 0) == sum(g - 1 for g in groups)
        moved = sum(v["bytes"] for k, v in pa.items() if k.startswith("phaseC"))
        assert moved == pytest.approx(per_pass * sum(3 * (g - 1) + g + 5 if g >= 2 else 5 for g in groups), rel=1e-9)
    # no group of two, and none without the deferred update: the plain kernel in every iteration, bit-identical
    for pair, defer in ((2, 1), (4, 0)):
        a, ra, pa, mem = run([0.0], pair, defer, [12])
        assert ra == rb and np.array_equal(a[0], b[0]) and mem == mem_plain
        assert set(k for k in pa if k.startswith("phaseC")) == {"phaseC"}
    o = orc.sbcgrq(U, dims, mass, Bh, [0.0], 0.0, 0.0, max_iterations=12)
    a, _, _, _ = run([0.0], 4, 1, [12])
    assert rel_err(a[0], o["X"][0]) < 1e-10
    # the tail of a two-shift solve: the shifted system leaves the active set early, the groups go on for X_0's sake
    shifts = [0.0, 2.0]
    o = orc.sbcgrq(U, dims, mass, Bh, shifts, 0.0, 0.5, max_iterations=12, trace_limit=12)
    visited = o["trace"]["residual_shift"][:, 1] >= 0.0
    assert visited[0] and not visited[4:].any()
    a, ra, pa, _ = run(shifts, 4, 1, [12], 0.5)
    c, rc, pc, _ = run(shifts, 0, 1, [12], 0.5)
    assert ra == rc and np.array_equal(a[1], c[1]) and rel_err(a[0], c[0]) < 1e-13
    assert pa["phaseC_multi4"]["count"] == 3 and pa["phaseC_p0"]["count"] == 9 and "phaseC" not in pa
    for s in range(2):
        assert rel_err(a[s], o["X"][s]) < 1e-10


def test_batched_row_kernels_against_the_plain_ones(bc, orc, monkeypatch):
    """m = 16, row counts that are multiples of 512: phase B, k_phaseC_p0 and the right-multiplications K5 / K6 keep a chunk
    of 32 tiles in LDS and write it out together (kernels_mfma.hip: k_phaseB_batched, k_phaseC_p0_batched, k_rmul_mfma_batched).
    Same products on the same values: K5 / K6 are bit-identical to the plain kernels (BCG_ROW_BATCHED=0); in the solver phase
    B's Gram partial sums take another order, so a solve agrees to rounding -- and with the oracle as before."""
    monkeypatch.setenv("BCG_HOP_PATCH", "16,2,2")
    m, dims, mass, shifts = 16, [16, 8, 8, 8], 0.2, [0.0, 1e-3, 0.1, 2.0]
    assert (int(np.prod(dims)) * 3) % 512 == 0
    U = orc.fill_gauge(dims, 91)
    Bh = orc.fill_field(m, int(np.prod(dims)), 92)
    Xh = orc.fill_field(m, int(np.prod(dims)), 93)
    rng = np.random.default_rng(94)
    C = (rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m))) * 0.2
    out = {}
    for batched in ("1", "0"):
        monkeypatch.setenv("BCG_ROW_BATCHED", batched)
        ctx = bc.Context(dims)
        ctx.profiling(True)
        x = bc.block_fermion_field(ctx, m, Xh)
        y = bc.block_fermion_field(ctx, m, Bh)
        y.add(x, C)                # K5, inc/fields.hpp:70-77
        k5 = y.download()
        y.rescale_add(C, x, 0.5)   # K6, inc/fields.hpp:79-90
        k6 = y.download()
        D = bc.dirac_op(ctx, mass, U=U)
        B = bc.block_fermion_field(ctx, m, Bh)
        X = [bc.block_fermion_field(ctx, m) for _ in shifts]
        st = bc.SBCGrQState(X, B, D, shifts, 0.0, 0.0)
        st.iterate(9)
        res = st.residual
        st.end()
        out[batched] = (k5, k6, [f.download() for f in X], res)
        ctx.close()
    assert np.array_equal(out["1"][0], out["0"][0]) and np.array_equal(out["1"][1], out["0"][1])
    for s in range(len(shifts)):
        assert rel_err(out["1"][2][s], out["0"][2][s]) < 1e-12, s
    assert abs(out["1"][3] - out["0"][3]) <= 1e-12 * out["0"][3]
    o = orc.sbcgrq(U, dims, mass, Bh, shifts, 0.0, 0.0, max_iterations=9)
    for s in range(len(shifts)):
        assert rel_err(out["1"][2][s], o["X"][s]) < 1e-10


def test_grouping_randomised_schedules(bc, orc, monkeypatch):
    """Seeded random schedules through the solver's state machine: width 8 / 16, 1 ... 8 shifts of random size (so that some
    retire early, some never), group depth 2 ... 4, X_0 deferred or not, the iterations asked for in random pieces (a piece
    of one iteration closes a group at once; a short piece leaves a partial group).  Whatever the schedule, the residual
    and every X_s with s >= 1 equal the plain solver's bit for bit, X_0 to rounding, and the oracle agrees."""
    monkeypatch.setenv("BCG_HOP_PATCH", "16,2,2")
    rng = np.random.default_rng(20261005)
    mass = 0.25
    for case in range(40):
        m = int(rng.choice([8, 16]))
        dims = [16, 8, 4, 8] if m == 8 else [16, 4, 4, 8]
        S = int(rng.choice([1, 1, 2, 3, 4, 5, 8]))
        shifts = [0.0] + sorted(float(x) for x in rng.choice([1e-4, 1e-2, 0.05, 0.3, 0.8, 1.5, 3.0, 6.0, 12.0], size=S - 1, replace=False))
        depth, defer = int(rng.integers(2, 5)), int(rng.integers(0, 2))
        eps_shifts = float(rng.choice([0.0, 1e-3, 3e-2, 0.3]))
        pieces = [int(x) for x in rng.integers(1, 7, size=int(rng.integers(1, 5)))]
        U = orc.fill_gauge(dims, 200 + case)
        Bh = orc.fill_field(m, int(np.prod(dims)), 300 + case)

        def run(pair):
            monkeypatch.setenv("BCG_PAIR_SHIFTS", str(pair))
            monkeypatch.setenv("BCG_DEFER_X0", str(defer))
            ctx = bc.Context(dims)
            D = bc.dirac_op(ctx, mass, U=U)
            B = bc.block_fermion_field(ctx, m, Bh)
            X = [bc.block_fermion_field(ctx, m) for _ in shifts]
            st = bc.SBCGrQState(X, B, D, shifts, 0.0, eps_shifts)
            for n in pieces:
                st.iterate(n)
            res, it = st.residual, st.iterations
            st.end()
            return [x.download() for x in X], res, it

        what = dict(case=case, m=m, shifts=shifts, depth=depth, defer=defer, eps_shifts=eps_shifts, pieces=pieces)
        a, ra, ia = run(depth)
        b, rb, ib = run(0)
        assert (ra, ia) == (rb, ib) and ia == sum(pieces), what
        for s in range(S):
            if s == 0:
                assert rel_err(a[0], b[0]) < 1e-13, what
            else:
                assert np.array_equal(a[s], b[s]), (what, s)
        o = orc.sbcgrq(U, dims, mass, Bh, shifts, 0.0, eps_shifts, max_iterations=sum(pieces))
        for s in range(S):
            assert rel_err(a[s], o["X"][s]) < 1e-10, (what, s)


@pytest.mark.parametrize("defer_x0", [0, 1], ids=["x0-every-iteration", "x0-deferred"])
@pytest.mark.parametrize("m,dims,ring", [(16, [32, 4, 4, 12], 4), (8, [32, 4, 4, 8], 4)], ids=["m16", "m8"])
def test_grouped_shift_updates_in_capacity_mode(bc, orc, m, dims, ring, defer_x0, monkeypatch):
    """Capacity mode groups the shift updates over two iterations: that depth needs no memory (the phase B that closes a
    pair writes the new residual block over T and a released buffer becomes the next T).  Bit-identical to the ungrouped
    solver in the same mode, for even and odd iteration counts and with the caller's B consumed or kept.
    x0-deferred (the default): X_0's update of a pair's first iteration waits for the closing pass too, in the spare-less
    form (DeferredX0 in capi_solvers.hip: the lost P_0 is recovered from the next one and the residual block) -- no memory
    either; the first iteration of a pair then moves three field passes for shift 0 instead of five, X_0 agrees to
    rounding, everything else stays bit-identical."""
    monkeypatch.setenv("BCG_HOP_PATCH", "16,2,2")
    monkeypatch.setenv("BCG_DEFER_X0", str(defer_x0))
    shifts, mass = [0.0, 1e-3, 0.1], 0.2
    U = orc.fill_gauge(dims, 71)
    Bh = orc.fill_field(m, int(np.prod(dims)), 72)

    def run(pair, iters, consume):
        monkeypatch.setenv("BCG_PAIR_SHIFTS", str(pair))
        ctx = bc.Context(dims)
        ctx.capacity_mode(ring)
        ctx.profiling(True)
        D = bc.dirac_op(ctx, mass, U=U)
        B = bc.block_fermion_field(ctx, m, Bh)
        X = [bc.block_fermion_field(ctx, m) for _ in shifts]
        st = bc.SBCGrQState(X, B, D, shifts, 0.0, 0.0, consume_B=consume)
        st.iterate(iters)
        res = st.residual
        st.end()
        return [x.download() for x in X], res, ctx.profile(), ctx.sbcgrq_device_bytes(m, len(shifts), consume_B=consume), B.download()

    for iters, consume in ((6, True), (7, False), (1, True)):
        a, ra, pa, mem_a, Ba = run(4, iters, consume)  # asks for four, capacity mode grants two
        b, rb, pb, mem_b, Bb = run(0, iters, consume)
        assert ra == rb and mem_a == mem_b
        for s in range(len(shifts)):
            if s == 0 and defer_x0:
                assert rel_err(a[0], b[0]) < 1e-13, (iters, consume)
            else:
                assert np.array_equal(a[s], b[s]), (iters, consume, s)
        assert pa.get("phaseC_multi2", {}).get("count", 0) == iters // 2 and "phaseC_multi4" not in pa
        assert pa.get("phaseC_p0", {}).get("count", 0) == (iters // 2 if defer_x0 else 0) and "phaseC_p0" not in pb
        if iters >= 2:
            # a pair moves 5 + (2 + 4 S) field passes, 3 + (2 + 4 S) with X_0 deferred (no spare field read); a plain iteration 1 + 4 S
            S = len(shifts)
            per_pass = pb["phaseC"]["bytes"] / (iters * (1 + 4 * S))
            moved = sum(v["bytes"] for k, v in pa.items() if k.startswith("phaseC"))
            want = (iters // 2) * ((3 if defer_x0 else 5) + 2 + 4 * S) + (iters % 2) * (1 + 4 * S)
            assert moved == pytest.approx(per_pass * want, rel=1e-9)
        if not consume:
            assert np.array_equal(Ba, Bh) and np.array_equal(Bb, Bh)  # the rotation of T, Q never touches the caller's B
    o = orc.sbcgrq(U, dims, mass, Bh, shifts, 0.0, 0.0, max_iterations=6)
    a, _, _, _, _ = run(4, 6, True)
    for s in range(len(shifts)):
        assert rel_err(a[s], o["X"][s]) < 1e-10
    if defer_x0:
        # the guard of the spare-less form (||rho||_F ||rho^-1||_F <= limit * m): refused, the iteration runs the plain
        # five-pass update and everything is bit-identical to the ungrouped solver again
        monkeypatch.setenv("BCG_DEBUG_X0_COND_LIMIT", "0")
        g, rg, pg, _, _ = run(4, 6, True)
        b, rb, _, _, _ = run(0, 6, True)
        assert rg == rb and "phaseC_p0" not in pg and pg["phaseC_multi2"]["count"] == 3
        for s in range(len(shifts)):
            assert np.array_equal(g[s], b[s]), s


def test_grouped_shift_updates_with_more_shifts_than_one_launch_holds(bc, orc, monkeypatch):
    """8 shifts at m = 16, groups of four iterations: the 60 coefficient matrices of a closing pass do not fit a CU's LDS, so
    it takes two launches of four shifts, each reading (and normalising) the four residual blocks again.  Bit-identical to
    the ungrouped solver; against the oracle."""
    monkeypatch.setenv("BCG_HOP_PATCH", "16,2,2")
    m, dims, mass = 16, [16, 4, 4, 8], 0.2
    shifts = [0.0, 1e-4, 1e-3, 1e-2, 0.05, 0.1, 0.5, 2.0]
    U = orc.fill_gauge(dims, 81)
    Bh = orc.fill_field(m, int(np.prod(dims)), 82)
    outs = {}
    for pair in (4, 0):
        monkeypatch.setenv("BCG_PAIR_SHIFTS", str(pair))
        ctx = bc.Context(dims)
        ctx.profiling(True)
        D = bc.dirac_op(ctx, mass, U=U)
        B = bc.block_fermion_field(ctx, m, Bh)
        X = [bc.block_fermion_field(ctx, m) for _ in shifts]
        st = bc.SBCGrQState(X, B, D, shifts, 0.0, 0.0)
        st.iterate(9)
        st.end()
        outs[pair] = ([x.download() for x in X], ctx.profile())
    assert rel_err(outs[4][0][0], outs[0][0][0]) < 1e-13  # X_0: deferred, composed products (DeferredX0)
    for s in range(1, len(shifts)):
        assert np.array_equal(outs[4][0][s], outs[0][0][s]), s
    # two groups of four, two launches each (the first takes three shifts besides shift 0: its four composed matrices need room)
    assert outs[4][1]["phaseC_multi4"]["count"] == 4 and "phaseC_multi4" not in outs[0][1]
    o = orc.sbcgrq(U, dims, mass, Bh, shifts, 0.0, 0.0, max_iterations=9)
    for s in range(len(shifts)):
        assert rel_err(outs[4][0][s], o["X"][s]) < 1e-10
