"""Parity of the PRODUCTION launch geometry at BASELINE.json's full per-GPU shapes (default tuning: no BCG_HOP_*
overrides, so 64^4 runs the bundle sweep k_hop4b with 512 blocks over 16x8x8 patches -- k_hop4c for the tile classes --
exactly what bench.py times).

Two independent checks, both against the CPU oracle, neither needs a lattice-sized host computation:

 1. Sampled operator.  U and psi come from the counter-based generator (bit-identical on CPU and GPU); the oracle
    evaluates (D psi)(x) and (A psi)(x) at ~1000 chosen sites from the generator alone (oracle.hpp, sampled evaluator)
    and the device values at those sites are fetched with bcg_field_download_sites.  Sites: every corner of the
    periodic wrap, faces/edges, first and last x3 slice, tile / patch / XCD-class borders, plus uniformly random ones.
    A mis-wrapped slice, a swapped ghost face or a wrong tile offset changes some of these values by O(1).

 2. Replicated solve.  A base lattice b is tiled r times per direction; the solution of the tiled problem is the
    tiled solution of the base problem, Gram sums scale by prod(r), hence alpha_k, rho_k, beta_s are IDENTICAL, delta_k
    and alpha_s scale by sqrt(prod r), iteration counts agree and X_full(x) = X_base(x mod b).  The oracle solves the base lattice
    (seconds); the GPU solves the full lattice with every production kernel (fused-Gram stencil, phase B, phase C,
    reductions over the full grid).  What this cannot see (errors that move data by a multiple of the base extents)
    is exactly what check 1 sees.

The oracle side of both is validated on the CPU in tests/test_oracle_sampled.py.
"""
import numpy as np
import pytest

from conftest import TOL_COEFF, TOL_KERNEL, rel_err

pytestmark = pytest.mark.gpu

SEED_U, SEED_PSI = 71, 72


@pytest.fixture(scope="module")
def bc():
    import blockcg_amd
    return blockcg_amd


def chosen_sites(dims, n_special=700, n_random=300, seed=5):
    """Lexicographic indices (x0 fastest) of the sites described in the module docstring."""
    rng = np.random.default_rng(seed)
    special = []
    for L in dims:
        vals = {0, 1, L - 2, L - 1, L // 2 - 1, L // 2, 7, 8, 15, 16, 17, 31, 32, 47, 48, 63, 64}
        special.append(np.array(sorted(v for v in vals if 0 <= v < L)))
    pts = []
    for corner in range(16):  # all corners of the periodic wrap
        pts.append([(dims[mu] - 1) if (corner >> mu) & 1 else 0 for mu in range(4)])
    for _ in range(n_special):
        pts.append([int(rng.choice(special[mu])) for mu in range(4)])
    for x3 in (0, dims[3] - 1):  # first / last x3 slice, random in the other directions
        for _ in range(40):
            pts.append([int(rng.integers(dims[0])), int(rng.integers(dims[1])), int(rng.integers(dims[2])), x3])
    for _ in range(n_random):
        pts.append([int(rng.integers(d)) for d in dims])
    p = np.array(pts, dtype=np.int64)
    idx = ((p[:, 3] * dims[2] + p[:, 2]) * dims[1] + p[:, 1]) * dims[0] + p[:, 0]
    return np.unique(idx)


OPERATOR_CASES = [
    # id, dims, m, mode
    ("64c4_m16_default", [64, 64, 64, 64], 16, "default"),
    ("64c4_m16_tile_classes", [64, 64, 64, 64], 16, "classes"),
    ("64c4_m16_capacity8", [64, 64, 64, 64], 16, "capacity"),
    ("32c4_m8_default", [32, 32, 32, 32], 8, "default"),
    ("64c3x32_m32_default", [64, 64, 64, 32], 32, "default"),
    ("64c3x128_m16_capacity8", [64, 64, 64, 128], 16, "capacity"),  # the per-GPU share of 128^4 on a (2,2,2,1) grid (row-form windows)
    # ... with the ring `bench.py --gpus 8` selects, in both chunkings that ring runs: serial (one rank: C = 30, windows of
    # 30, 30, 30, 30 and 8 slices) and the overlapped form's C = 15 (windows of 15 x 8 and 8), forced on this one rank
    ("64c3x128_m16_capacity32", [64, 64, 64, 128], 16, "capacity32"),
    ("64c3x128_m16_capacity32_chunks_of_15", [64, 64, 64, 128], 16, "capacity32c15"),
]


def _capacity_of(mode):
    """(ring slices, forced chunk length or 0) of a `capacity*` mode string"""
    if not mode.startswith("capacity"):
        return 0, 0
    rest = mode[len("capacity"):]
    ring, _, chunk = rest.partition("c")
    return (int(ring) if ring else 8), (int(chunk) if chunk else 0)


def _release(ctx, *objs):
    """The 250-285 GB cases must give their memory back BEFORE the next case allocates: destroy the fields now (not when
    the interpreter gets to the frame's locals), then the context, and collect."""
    import gc
    for o in objs:
        for f in (o if isinstance(o, (list, tuple)) else [o]):
            f.__del__()
    ctx.close()
    gc.collect()


@pytest.mark.parametrize("case", OPERATOR_CASES, ids=[c[0] for c in OPERATOR_CASES])
def test_operator_production_geometry_vs_sampled_oracle(bc, orc, case, monkeypatch):
    """bcg_dirac_hop (inc/dirac_op.hpp:14-21) and bcg_dirac_apply (:36-43) with default tuning vs the oracle at chosen
    sites; both stencil variants of the operator apply (plain and shifted) are exercised by `apply`."""
    _, dims, m, mode = case
    for k in ("BCG_HOP_BLOCKS", "BCG_HOP_PATCH", "BCG_HOP_WALK", "BCG_HOP_C2"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("BCG_FORCE_TILE_CLASSES", "1" if mode == "classes" else "0")
    mass = 0.37
    ring, chunk = _capacity_of(mode)
    if chunk:
        monkeypatch.setenv("BCG_RING_CHUNK", str(chunk))  # the overlapped form's chunk length on a rank without neighbours
    ctx = bc.Context(dims)
    if ring:
        ctx.capacity_mode(ring)
    D = bc.dirac_op(ctx, mass, seed=SEED_U)
    psi = bc.block_fermion_field(ctx, m).setRandom(seed=SEED_PSI)
    out = bc.block_fermion_field(ctx, m)
    sites = chosen_sites(dims)
    orc.set_threads(8)
    try:
        # the generator itself, at these sites
        assert np.array_equal(psi.download_sites(sites[:32]),
                              np.stack([orc.fill_field(m, 1, SEED_PSI, first_global_site=int(s))[0] for s in sites[:32]]))
        D.D(out, psi)
        want = orc.hop_sampled(m, dims, SEED_U, SEED_PSI, sites)
        got = out.download_sites(sites)
        assert rel_err(got, want) < TOL_KERNEL
        assert np.abs(got - want).max() < 1e-13 * np.abs(want).max()  # per site, not only in the norm
        D.op(out, psi)
        want = orc.apply_sampled(m, dims, SEED_U, SEED_PSI, mass, sites)
        got = out.download_sites(sites)
        assert rel_err(got, want) < TOL_KERNEL
        assert np.abs(got - want).max() < 1e-13 * np.abs(want).max()
    finally:
        orc.set_threads(1)
        _release(ctx, psi, out, D)


def tile_sites(a, base_dims, reps):
    """Replicate a per-site array [V_base, ...] (x0 fastest) reps[mu] times along direction mu."""
    b = list(base_dims)
    rest = a.shape[1:]
    g = a.reshape([b[3], b[2], b[1], b[0]] + list(rest))
    g = np.tile(g, [reps[3], reps[2], reps[1], reps[0]] + [1] * len(rest))
    return g.reshape([-1] + list(rest))


REPLICA_CASES = [
    # id, base dims, reps, m, shifts, mode
    ("64c4_m16_S4", [16, 8, 8, 16], [4, 8, 8, 4], 16, [0.0, 1e-6, 1e-4, 1e-2], "default"),
    ("64c4_m16_S2_tile_classes", [16, 8, 8, 16], [4, 8, 8, 4], 16, [0.0, 1e-2], "classes"),
    ("32c4_m8_S1", [16, 8, 8, 8], [2, 4, 4, 4], 8, [0.0], "default"),
    ("64c3x32_m32_S8", [16, 8, 8, 8], [4, 8, 8, 4], 32, [0.0, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1.0], "default"),
    ("64c3x128_m16_S4_capacity32", [16, 8, 8, 16], [4, 8, 8, 8], 16, [0.0, 1e-6, 1e-4, 1e-2], "capacity32"),  # bench.py --gpus 8's ring
    ("64c3x128_m16_S4_capacity32_chunks_of_15", [16, 8, 8, 16], [4, 8, 8, 8], 16, [0.0, 1e-6, 1e-4, 1e-2], "capacity32c15"),
]


@pytest.mark.parametrize("case", REPLICA_CASES, ids=[c[0] for c in REPLICA_CASES])
def test_replicated_solve_matches_oracle_on_base_lattice(bc, orc, case, monkeypatch):
    """SBCGrQ (inc/block_solvers.hpp:91-185) at the full per-GPU shapes of BASELINE configs 1-4, default tuning."""
    _, base, reps, m, shifts, mode = case
    for k in ("BCG_HOP_BLOCKS", "BCG_HOP_PATCH", "BCG_HOP_WALK", "BCG_HOP_C2"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("BCG_FORCE_TILE_CLASSES", "1" if mode == "classes" else "0")
    dims = [b * r for b, r in zip(base, reps)]
    nrep = int(np.prod(reps))
    mass, eps = 0.5, 1e-8
    Vb = int(np.prod(base))
    Ub = orc.fill_gauge(base, 81)
    Bb = orc.fill_field(m, Vb, 82)
    ntrace = 3
    orc.set_threads(8)
    try:
        o = orc.sbcgrq(Ub, base, mass, Bb, shifts, eps, eps, max_iterations=2000, trace_limit=ntrace)
    finally:
        orc.set_threads(1)
    assert 3 < o["iterations"] < 2000

    ring, chunk = _capacity_of(mode)
    if chunk:
        monkeypatch.setenv("BCG_RING_CHUNK", str(chunk))
    ctx = bc.Context(dims)
    if ring:
        ctx.capacity_mode(ring)
    D = bc.dirac_op(ctx, mass, U=tile_sites(Ub, base, reps))
    B = bc.block_fermion_field(ctx, m, tile_sites(Bb, base, reps))
    X = [bc.block_fermion_field(ctx, m) for _ in shifts]
    try:
        info = bc.SBCGrQ(X, B, D, shifts, eps, eps, max_iterations=o["iterations"] + 50, trace_limit=ntrace,
                         consume_B=mode.startswith("capacity"), return_info=True)
        sites = chosen_sites(dims, n_special=300, n_random=200)
        got = [X[s].download_sites(sites) for s in range(len(shifts))]
    finally:
        _release(ctx, X, B, D)
    assert abs(info["iterations"] - o["iterations"]) <= 1, (info["iterations"], o["iterations"])
    tr, otr = info["trace"], o["trace"]
    for key in ("alpha", "rho", "beta_s"):
        assert rel_err(tr[key], otr[key]) < TOL_COEFF, key
    for key in ("delta", "alpha_s"):  # these carry delta_0 = chol(B^dag B), which scales with sqrt(copies) (:115-116,167-168)
        assert rel_err(tr[key], np.sqrt(nrep) * otr[key]) < TOL_COEFF, key
    # X_full(x) = X_base(x mod base) at chosen sites of the full lattice
    x = [sites % dims[0], (sites // dims[0]) % dims[1], (sites // (dims[0] * dims[1])) % dims[2],
         sites // (dims[0] * dims[1] * dims[2])]
    bsite = (((x[3] % base[3]) * base[2] + x[2] % base[2]) * base[1] + x[1] % base[1]) * base[0] + x[0] % base[0]
    for s in range(len(shifts)):
        assert rel_err(got[s], o["X"][s][bsite]) < 1e-7, s


GRAM_CASES = [("64c4_m16", [64, 64, 64, 64], 16), ("32c4_m8", [32, 32, 32, 32], 8), ("64c3x32_m32", [64, 64, 64, 32], 32)]


@pytest.mark.parametrize("case", GRAM_CASES, ids=[c[0] for c in GRAM_CASES])
def test_block_inner_products_at_full_size_vs_chunked_oracle(bc, orc, case):
    """hermitian_dot (inc/fields.hpp:103-122) and thinQR's factor (:140-146) on generated, non-periodic fields of the full
    per-GPU sizes, against Gram sums the oracle accumulates in chunks of 4096 sites from the generator (nothing of size V on
    the host): the production grids of the MFMA Gram kernels and their fixed-order reductions."""
    _, dims, m = case
    ctx = bc.Context(dims)
    A = bc.block_fermion_field(ctx, m).setRandom(seed=5)
    B = bc.block_fermion_field(ctx, m).setRandom(seed=6)
    orc.set_threads(16)
    try:
        want_ab = orc.gram_generated(m, ctx.V, 5, 6)
        want_bb = orc.gram_generated(m, ctx.V, 6, 6)
    finally:
        orc.set_threads(1)
    assert rel_err(A.hermitian_dot(B), want_ab) < 1e-12
    G = B.hermitian_dot(B)
    assert rel_err(G, want_bb) < 1e-12
    R = B.thinQR()   # in place: B becomes Q
    assert rel_err(R, orc.cholesky_upper(want_bb)) < 1e-11
    assert rel_err(B.hermitian_dot(B), np.eye(m)) < 1e-11
    _release(ctx, A, B)


@pytest.mark.parametrize("case", GRAM_CASES, ids=[c[0] for c in GRAM_CASES])
def test_row_kernels_at_full_size_vs_sampled_oracle(bc, orc, case):
    """add(rhs, m x m) (K5), rescale_add(m x m, rhs, b) (K6), add(rhs, double) (K3) and
    multiply_upper_triangular_inverse_RHS (K7) -- inc/fields.hpp:70-90,125-136 -- are site-local, so the oracle evaluates
    them on the generated tiles of the chosen sites only; the device runs its persistent full-size grids."""
    _, dims, m = case
    ctx = bc.Context(dims)
    sites = chosen_sites(dims, n_special=200, n_random=300)
    y0 = np.stack([orc.fill_field(m, 1, 7, first_global_site=int(s))[0] for s in sites])
    b0 = np.stack([orc.fill_field(m, 1, 8, first_global_site=int(s))[0] for s in sites])
    rng = np.random.default_rng(m)
    M = rng.uniform(-1, 1, (m, m)) + 1j * rng.uniform(-1, 1, (m, m))
    R = np.triu(rng.uniform(-1, 1, (m, m)) + 1j * rng.uniform(-1, 1, (m, m))) + 3.0 * np.eye(m)
    B = bc.block_fermion_field(ctx, m).setRandom(seed=8)
    new_y = lambda: bc.block_fermion_field(ctx, m).setRandom(seed=7)  # noqa: E731
    assert np.array_equal(new_y().download_sites(sites), y0)
    assert rel_err(new_y().add(B, M).download_sites(sites), orc.add_matrix(y0, b0, M)) < TOL_KERNEL
    assert rel_err(new_y().rescale_add(M, B, 0.7).download_sites(sites), orc.rescale_add_matrix(y0, M, b0, 0.7)) < TOL_KERNEL
    assert rel_err(new_y().add(B, -0.3).download_sites(sites), orc.add_scalar(y0, b0, -0.3)) < TOL_KERNEL
    assert rel_err(new_y().multiply_upper_triangular_inverse_RHS(R).download_sites(sites), orc.tri_solve_rhs(y0, R)) < 1e-12
    _release(ctx, B)
