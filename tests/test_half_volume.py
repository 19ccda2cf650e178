"""Half-volume (parity-decoupled) solve, SURVEY.md section 8f-4 / Appendix D: dirac_op::D couples opposite site parities only
(inc/dirac_op.hpp:14-21), so op = mass^2 - D^2 (inc/dirac_op.hpp:36-43) is block diagonal in the parity and the multi-shift
solve splits into two solves on V/2 sites.  Checked against the full-volume oracle: the parity-compact layout, the
operator blocks, every field primitive on half fields, and the solve itself."""
import numpy as np
import pytest

from conftest import TOL_KERNEL, rel_err

pytestmark = pytest.mark.gpu


def _parity_masks(dims):
    V = int(np.prod(dims))
    idx = np.arange(V)
    par = np.zeros(V, dtype=np.int64)
    for L in dims:           # x0 fastest
        par += idx % L
        idx = idx // L
    return [(par % 2) == p for p in (0, 1)]


@pytest.mark.parametrize("dims,m", [([8, 4, 4, 6], 16), ([4, 6, 2, 4], 8), ([8, 4, 4], 3), ([16, 2, 4, 2], 32)],
                         ids=lambda v: "x".join(map(str, v)) if isinstance(v, list) else str(v))
def test_half_fields_layout_operator_and_primitives(orc, dims, m):
    import blockcg_amd as bc
    ctx = bc.Context(dims)
    mass = 0.37
    D = bc.dirac_op(ctx, mass, seed=11)
    U = orc.fill_gauge(dims, 11)
    Fh = orc.fill_field(m, ctx.V, 12)
    masks = _parity_masks(dims)
    F = bc.block_fermion_field(ctx, m).setRandom(seed=12)
    even, odd = F.split_parity()
    # layout: a half field is the full field restricted to its parity, in the full lattice's order; the generator agrees
    for par, half in enumerate((even, odd)):
        assert half.V == ctx.V // 2 and np.array_equal(half.download(), Fh[masks[par]])
        g = bc.block_fermion_field(ctx, m, parity=par).setRandom(seed=12)
        assert np.array_equal(g.download(), Fh[masks[par]])
    # merge is the inverse of split
    G = bc.block_fermion_field(ctx, m)
    G.merge_parity(even, odd)
    assert np.array_equal(G.download(), Fh)
    # D maps a half field to the other parity; A = mass^2 - D^2 stays within one
    Dh = orc.hop(U, dims, Fh)
    Ah = orc.dirac_apply(U, dims, mass, Fh)
    for par, half in enumerate((even, odd)):
        out = bc.block_fermion_field(ctx, m, parity=par)
        D.op(out, half)
        assert rel_err(out.download(), Ah[masks[par]]) < TOL_KERNEL
        other = bc.block_fermion_field(ctx, m, parity=1 - par)
        D.D(other, half)
        # (D F) at the sites of parity 1 - par only sees F's sites of parity par
        assert rel_err(other.download(), Dh[masks[1 - par]]) < TOL_KERNEL
    # field primitives on half fields = the full-volume ones restricted (site-local), Gram sums split over the parities
    Gfull = F.hermitian_dot(F)
    Ge, Go = even.hermitian_dot(even), odd.hermitian_dot(odd)
    assert rel_err(Ge + Go, Gfull) < 1e-13
    M = (np.arange(m * m).reshape(m, m) % 7 - 3 + 1j * (np.arange(m * m).reshape(m, m) % 5 - 2)) / 7.0
    y = bc.block_fermion_field(ctx, m, parity=1).setRandom(seed=13)
    yh = orc.fill_field(m, ctx.V, 13)[masks[1]]
    y.add(odd, M)
    assert rel_err(y.download(), orc.add_matrix(yh, Fh[masks[1]], M)) < TOL_KERNEL
    # operands of different parity are refused
    with pytest.raises(bc.BlockCGError):
        y.add(even, M)


def test_half_volume_solve_matches_full_volume_oracle(orc):
    import blockcg_amd as bc
    dims, m, mass = [8, 4, 4, 4], 16, 0.2
    shifts = [0.0, 1e-3, 5e-2]
    eps = 1e-10
    ctx = bc.Context(dims)
    D = bc.dirac_op(ctx, mass, seed=21)
    B = bc.block_fermion_field(ctx, m).setRandom(seed=22)
    X = [bc.block_fermion_field(ctx, m) for _ in shifts]
    its = bc.SBCGrQ_half_volume(X, B, D, shifts, eps, eps)
    U = orc.fill_gauge(dims, 21)
    Bh = orc.fill_field(m, ctx.V, 22)
    ref = orc.sbcgrq(U, dims, mass, Bh, shifts, eps, eps)
    Xh = np.stack([x.download() for x in X])
    # the reference's acceptance test (test/solvers.cpp:104-116), recomputed by the oracle on the FULL lattice
    res = orc.true_residuals(U, dims, mass, Bh, shifts, Xh)
    assert res.max() < 2 * eps, res
    assert rel_err(Xh, ref["X"]) < 1e-8
    # each half solve sees a Krylov space of its own: no more operator applications than the full-volume solve needs
    assert max(its) <= ref["iterations"] + 1, (its, ref["iterations"])
    # the full-volume residual check on the device agrees
    dres = bc.true_residuals(X, B, D, shifts)
    assert dres.max() < 2 * eps
    # and the half fields' own residual check (half fields throughout)
    Be, Bo = B.split_parity()
    for par, Bp in enumerate((Be, Bo)):
        Xp = [x.split_parity()[par] for x in X]
        assert bc.true_residuals(Xp, Bp, D, shifts).max() < 2 * eps


def test_half_volume_memory_and_rejections():
    import blockcg_amd as bc
    ctx = bc.Context([8, 4, 4, 4])
    h = bc.block_fermion_field(ctx, 16, parity=0)
    assert ctx.lib.bcg_field_sites(h.h) == ctx.V // 2 and ctx.lib.bcg_field_parity(h.h) == 0
    with pytest.raises(bc.BlockCGError):      # odd extent: the parity is not consistent across the periodic wrap
        bc.block_fermion_field(bc.Context([6, 3, 4, 4]), 4, parity=0)
    full = bc.block_fermion_field(ctx, 16)
    assert ctx.lib.bcg_field_parity(full.h) == -1
    # the memory plan of one half-volume solve at 64^4, m = 16, 4 shifts (source consumed): 14 half fields (X_s, P_s, Q, T,
    # tmp, the two further residual buffers of the grouped shift updates and the spare P_0 of the deferred X_0 update) + full
    # links, against 14 full fields
    big = bc.Context([64, 64, 64, 64])
    half, whole = big.sbcgrq_device_bytes_half(16, 4, consume_B=True), big.sbcgrq_device_bytes(16, 4, consume_B=True)
    assert 99e9 < half < 101e9 and 189e9 < whole < 192e9 and half < 0.54 * whole


@pytest.mark.parametrize("m,dims,patch", [(16, [64, 8, 8, 6], "16,2,2"), (32, [32, 8, 8, 4], "8,2,2")], ids=["m16", "m32"])
@pytest.mark.parametrize("sync", ["0", "4"])
def test_checkerboard_bundle_sweep_matches_generic_and_oracle(orc, m, dims, patch, sync, monkeypatch):
    """The fast form of the half-volume operator -- k_hop4b in its checkerboard mode (m = 16: 2 x 2 column bundles on the
    compact lattice, links of every other full-lattice site by LDS-DMA, all four directions' backward links gathered, fused
    Gram product folded in the kernel) -- on a lattice small enough for the whole-lattice oracle, with small patches so that
    the column walk applies; against the oracle and against the generic half-volume kernel."""
    import blockcg_amd as bc
    monkeypatch.setenv("BCG_HOP_PATCH", patch)
    monkeypatch.setenv("BCG_HOP_BLOCKS", "32")
    monkeypatch.setenv("BCG_HOP_BUNDLE_SYNC", sync if sync == "0" else "-" + sync)  # < 0: pace short sweeps too
    mass = 0.3
    shifts, iters = [0.0, 0.05], 4
    V = int(np.prod(dims))
    U = orc.fill_gauge(dims, 31)
    Fh = orc.fill_field(m, V, 32)
    Ah = orc.dirac_apply(U, dims, mass, Fh)
    masks = _parity_masks(dims)
    results = {}
    for generic in (False, True):
        ctx = bc.Context(dims)
        ctx.force_generic(generic)
        ctx.profiling(True)
        D = bc.dirac_op(ctx, mass, U=U)
        per_parity = []
        for par in (0, 1):
            B = bc.block_fermion_field(ctx, m, parity=par).setRandom(seed=32)
            out = bc.block_fermion_field(ctx, m, parity=par)
            D.op(out, B)
            assert rel_err(out.download(), Ah[masks[par]]) < TOL_KERNEL, (generic, par)
            X = [bc.block_fermion_field(ctx, m, parity=par) for _ in shifts]
            info = bc.SBCGrQ(X, B, D, shifts, 0.0, 0.0, max_iterations=iters, trace_limit=iters, return_info=True)
            per_parity.append((info["trace"], [x.download() for x in X]))
        prof = ctx.profile()
        assert ("stencil_form_k_hop4b_checkerboard" in prof) == (not generic), prof.keys()
        if not generic and m == 16:
            assert "hop_half_shifted_gram" in prof and "gram_pair" not in prof  # the fused product, folded in the kernel
        results[generic] = per_parity
    for par in (0, 1):
        for key in ("alpha", "rho", "delta", "alpha_s", "beta_s"):
            assert rel_err(results[False][par][0][key], results[True][par][0][key]) < 1e-10, (par, key)
        for s in range(len(shifts)):
            assert rel_err(results[False][par][1][s], results[True][par][1][s]) < 1e-11


def test_checkerboard_operator_at_full_size_sampled(orc):
    """The production geometry of the half-volume operator (64^4, m = 16: default patches, 512 blocks, paced) at sampled
    sites against the oracle's per-site evaluation from the generator (tests/test_fullsize_parity.py's method)."""
    import blockcg_amd as bc
    dims, m, mass = [64, 64, 64, 64], 16, 1e-3
    ctx = bc.Context(dims)
    ctx.profiling(True)
    D = bc.dirac_op(ctx, mass, seed=41)
    rng = np.random.default_rng(5)
    for par in (0, 1):
        F = bc.block_fermion_field(ctx, m, parity=par).setRandom(seed=42)
        out = bc.block_fermion_field(ctx, m, parity=par)
        D.op(out, F)
        # half sites: corners of the compact lattice, first / last slice, random ones
        h = np.unique(np.concatenate([rng.integers(0, ctx.V // 2, 600), np.arange(0, 64), np.arange(ctx.V // 2 - 64, ctx.V // 2),
                                      np.arange(32 * 64 * 64 - 40, 32 * 64 * 64 + 40)]))
        k, rest = h % 32, h // 32
        x1, x2, x3 = rest % 64, (rest // 64) % 64, rest // (64 * 64)
        x0 = 2 * k + ((x1 + x2 + x3 + par) & 1)
        full = x0 + 64 * (x1 + 64 * (x2 + 64 * x3))
        want = orc.apply_sampled(m, dims, 41, 42, mass, full)
        got = out.download_sites(h)
        err = np.abs(got - want).reshape(len(h), -1).max(axis=1) / np.abs(want).max()
        assert err.max() < 1e-13, (par, err.max())
    assert "stencil_form_k_hop4b_checkerboard" in ctx.profile()
