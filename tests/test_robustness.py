"""Error paths, memory hygiene and allocation order of the C ABI on the GPU (round-4 review items).

 * the host <-> device transfer pipeline survives a growth of the site-download staging buffer in between, and a context
   gives back everything it allocated (the leak that ran the 280-GB full-size case out of memory in round 3);
 * an error inside a group of deferred shift updates leaves every X_s at the last completed iteration, bit-identical to the
   solver that updates every shift in every iteration (the reference's behaviour: inc/block_solvers.hpp:161-181 run per iteration);
 * a solve that fits without the optional residual buffers still runs when memory is short (smaller grouping depth);
 * bcg_profile_reset clears flops as well as bytes.
"""
import gc

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bc():
    import blockcg_amd
    return blockcg_amd


def _free_bytes():
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info(0)[0]


def test_transfers_survive_staging_growth_and_contexts_release_their_memory(bc):
    """upload -> download_sites (allocates / grows the staging buffer) -> upload -> download in ONE context: exact round
    trips (round 3 freed the transfer pipeline's buffers in ensure_staging and kept using them).  Then a create / transfer /
    destroy loop must not drift in free device memory (round 3 leaked 2 x 64 MB of device and pinned memory per context)."""
    dims = [16, 8, 8, 8]
    rng = np.random.default_rng(3)
    ctx = bc.Context(dims)
    for m in (8, 16, 32):  # a wider field grows the staging buffer again
        a = rng.standard_normal((ctx.V, m, 3)) + 1j * rng.standard_normal((ctx.V, m, 3))
        f = bc.block_fermion_field(ctx, m, a)
        sites = np.array([0, 5, ctx.V - 1, 77], dtype=np.int64)
        assert np.array_equal(f.download_sites(sites), a[sites])
        g = bc.block_fermion_field(ctx, m, 2.0 * a)              # upload after the staging buffer changed
        assert np.array_equal(g.download(), 2.0 * a)
        assert np.array_equal(f.download(), a)
        assert np.array_equal(g.download_sites(sites), 2.0 * a[sites])
        del f, g
    ctx.close()
    del ctx
    gc.collect()

    def cycle():
        c = bc.Context(dims)
        a = np.ones((c.V, 16, 3), dtype=np.complex128)
        f = bc.block_fermion_field(c, 16, a)
        f.download_sites(np.arange(4, dtype=np.int64))
        f.download()
        D = bc.dirac_op(c, 0.1, seed=1)
        out = bc.block_fermion_field(c, 16)
        D.op(out, f)
        c.synchronize()
        del f, D, out
        c.close()

    cycle()  # first use: runtime-side pools (code objects, signal pools) settle
    cycle()
    before = _free_bytes()
    for _ in range(24):
        cycle()
    gc.collect()
    after = _free_bytes()
    assert before - after < (32 << 20), (before, after)  # round 3: 24 x 128 MB = 3 GB


@pytest.mark.parametrize("m,dims,depth", [(16, [16, 8, 8, 8], 4), (8, [16, 8, 4, 8], 4), (8, [16, 8, 4, 8], 3), (32, [16, 4, 4, 6], 2)],
                         ids=["m16-4", "m8-4", "m8-3", "m32-2"])
def test_error_inside_a_group_leaves_every_shift_current(bc, m, dims, depth, monkeypatch):
    """BCG_DEBUG_FAIL_ITER=k makes the Gram matrix after phase B of iteration k non-finite (the thinQR breakdown the header
    documents as BCG_ERR_NUMERIC).  With the shift updates grouped, 0 .. depth-1 iterations' updates of the shifts >= 1 are
    still deferred at that point; the library applies them before returning, so X_s of EVERY shift equals, bit for bit, what
    the ungrouped solver (BCG_PAIR_SHIFTS=0) leaves at the same failure.  A state that failed refuses further iterations."""
    monkeypatch.setenv("BCG_HOP_PATCH", "16,2,2")
    shifts, mass = [0.0, 1e-3, 0.1, 2.0], 0.2

    def run(pair, fail_at):
        monkeypatch.setenv("BCG_PAIR_SHIFTS", str(pair))
        monkeypatch.setenv("BCG_DEBUG_FAIL_ITER", str(fail_at))
        ctx = bc.Context(dims)
        D = bc.dirac_op(ctx, mass, seed=91)
        B = bc.block_fermion_field(ctx, m).setRandom(seed=92)
        X = [bc.block_fermion_field(ctx, m) for _ in shifts]
        st = bc.SBCGrQState(X, B, D, shifts, 0.0, 0.0)
        with pytest.raises(bc.BlockCGError) as e:
            st.iterate(12)
        assert e.value.code == 6 and "not finite" in str(e.value)  # BCG_ERR_NUMERIC
        with pytest.raises(bc.BlockCGError) as e2:
            st.iterate(1)
        assert e2.value.code == 1  # BCG_ERR_INVALID: the state is spent
        st.end()
        out = [x.download() for x in X]
        ctx.close()
        return out

    for fail_at in range(1, depth + 3):  # 0 .. depth-1 deferred iterations at the failure, and the first of the next group
        a = run(depth, fail_at)
        b = run(0, fail_at)
        for s in range(len(shifts)):
            if s == 0 and m != 32:  # X_0's deferred updates are composed products (DeferredX0): equal to rounding
                assert rel_err(a[0], b[0]) < 1e-13, fail_at
            else:
                assert np.array_equal(a[s], b[s]), (fail_at, s)
            assert np.isfinite(a[s]).all()
        if fail_at > 1:
            assert np.abs(a[1]).max() > 0  # the shifted systems did move before the failure


@pytest.mark.parametrize("m,dims,depth", [(16, [16, 8, 8, 8], 4), (8, [16, 8, 4, 8], 3)], ids=["m16-4", "m8-3"])
def test_error_inside_a_group_of_a_single_system(bc, m, dims, depth, monkeypatch):
    """A single system groups its iterations for the deferred X_0 update alone (pair_shifts_depth, x0_may_wait): a failure
    inside a group finds nothing to apply for the shifts >= 1 and the composed X_0 updates of 1 .. depth-1 iterations
    pending.  X_0 must equal the plain solver's at the same failure (to rounding), whatever the position in the group."""
    monkeypatch.setenv("BCG_HOP_PATCH", "16,2,2")
    mass = 0.2

    def run(pair, fail_at):
        monkeypatch.setenv("BCG_PAIR_SHIFTS", str(pair))
        monkeypatch.setenv("BCG_DEBUG_FAIL_ITER", str(fail_at))
        ctx = bc.Context(dims)
        ctx.profiling(True)
        D = bc.dirac_op(ctx, mass, seed=93)
        B = bc.block_fermion_field(ctx, m).setRandom(seed=94)
        X = [bc.block_fermion_field(ctx, m)]
        st = bc.SBCGrQState(X, B, D, [0.0], 0.0, 0.0)
        with pytest.raises(bc.BlockCGError) as e:
            st.iterate(12)
        assert e.value.code == 6  # BCG_ERR_NUMERIC
        st.end()
        out, prof = X[0].download(), ctx.profile()
        ctx.close()
        return out, prof

    for fail_at in range(1, depth + 3):
        a, pa = run(depth, fail_at)
        b, pb = run(0, fail_at)
        assert np.isfinite(a).all() and rel_err(a, b) < 1e-13, fail_at
        assert "phaseC_p0" not in pb
        if fail_at >= 2:  # at least one iteration completed inside a group before the failure
            assert pa["phaseC_p0"]["count"] >= 1 and np.abs(a).max() > 0


@pytest.mark.parametrize("m,dims,ring", [(16, [32, 4, 4, 12], 4), (8, [32, 4, 4, 8], 4)], ids=["m16", "m8"])
def test_error_inside_a_pair_in_capacity_mode(bc, m, dims, ring, monkeypatch):
    """Capacity mode groups in pairs and defers X_0 in the spare-less form: at a failure in the second iteration of a pair the
    first one's X_0 update is pending and its P_0 is gone -- it is applied from the current P_0 and the kept residual block
    (sbcgrq_flush_pending).  Every X_s equals what the ungrouped solver leaves at the same failure (X_0 to rounding)."""
    monkeypatch.setenv("BCG_HOP_PATCH", "16,2,2")
    shifts, mass = [0.0, 1e-3, 0.1], 0.2

    def run(pair, fail_at):
        monkeypatch.setenv("BCG_PAIR_SHIFTS", str(pair))
        monkeypatch.setenv("BCG_DEBUG_FAIL_ITER", str(fail_at))
        ctx = bc.Context(dims)
        ctx.capacity_mode(ring)
        ctx.profiling(True)
        D = bc.dirac_op(ctx, mass, seed=95)
        B = bc.block_fermion_field(ctx, m).setRandom(seed=96)
        X = [bc.block_fermion_field(ctx, m) for _ in shifts]
        st = bc.SBCGrQState(X, B, D, shifts, 0.0, 0.0)
        with pytest.raises(bc.BlockCGError) as e:
            st.iterate(12)
        assert e.value.code == 6
        st.end()
        out, prof = [x.download() for x in X], ctx.profile()
        ctx.close()
        return out, prof

    for fail_at in range(1, 6):
        a, pa = run(4, fail_at)
        b, pb = run(0, fail_at)
        for s in range(len(shifts)):
            assert np.isfinite(a[s]).all()
            if s == 0:
                assert rel_err(a[0], b[0]) < 1e-13, fail_at
            else:
                assert np.array_equal(a[s], b[s]), (fail_at, s)
        assert pa.get("phaseC_p0", {}).get("count", 0) == fail_at // 2 and "phaseC_p0" not in pb


def test_optional_residual_buffers_never_cost_the_solve_its_memory(bc, monkeypatch):
    """bcg_sbcgrq_begin allocates what the operator needs (tmp, scratch) BEFORE the optional residual buffers of the grouped
    shift updates.  With room for the base plan + 1.5 fields the default depth 4 (two extra fields) must fall back to depth 3
    and run; round 3 took the one extra field that fitted and then failed on tmp in the first iteration.
    The "full device" is BCG_DEBUG_FIELD_BUDGET, the library's deterministic stand-in for an out-of-memory on field
    allocations (free device memory as the runtime reports it is no measure of what a process can still allocate once
    earlier tests have allocated and freed hundreds of gigabytes in it)."""
    for k in ("BCG_HOP_BLOCKS", "BCG_HOP_PATCH", "BCG_PAIR_SHIFTS"):
        monkeypatch.delenv(k, raising=False)
    dims, m, shifts, mass, iters = [32, 32, 32, 32], 16, [0.0, 1e-4, 1e-2, 1.0], 0.3, 7
    field = 32 ** 4 * 3 * m * 16

    def run(extra_fields):
        # the caller's B and X_s (5 fields), then inside begin: P_s (4), T, Q (B is kept), tmp = 7 fields
        if extra_fields is None:
            monkeypatch.delenv("BCG_DEBUG_FIELD_BUDGET", raising=False)
        else:
            monkeypatch.setenv("BCG_DEBUG_FIELD_BUDGET", str(int((12 + extra_fields) * field)))
        ctx = bc.Context(dims)
        ctx.profiling(True)
        D = bc.dirac_op(ctx, mass, seed=11)
        B = bc.block_fermion_field(ctx, m).setRandom(seed=12)
        X = [bc.block_fermion_field(ctx, m) for _ in shifts]
        st = bc.SBCGrQState(X, B, D, shifts, 0.0, 0.0)
        st.iterate(iters)
        st.end()
        prof = ctx.profile()
        out = [x.download_sites(np.arange(0, ctx.V, 4099, dtype=np.int64)) for x in X]
        del X, B, D
        ctx.close()
        gc.collect()
        return out, {k for k in prof if k.startswith("phaseC")}

    a, pa = run(1.5)     # one of the two extra buffers fits: groups of three (and no room for the spare P_0: X_0 every iteration)
    b, pb = run(None)    # no limit: groups of four, X_0's updates deferred too (one more field, the spare P_0)
    c, pc = run(0.5)     # none fits: groups of two (T doubles as the second residual buffer)
    d, pd = run(3.25)    # both extra buffers and the spare fit: as without a limit
    assert pa == {"phaseC", "phaseC_multi3"}, pa            # 7 iterations = 3 + 3 + 1
    assert pb == {"phaseC_p0", "phaseC_multi4", "phaseC_multi3"}, pb   # 4 + 3, shift 0 inside a group as P_0 alone
    # 2 + 2 + 2 + 1: pairs defer X_0 without a spare field (the spare-less form), the odd last iteration is a plain one
    assert pc == {"phaseC", "phaseC_multi2", "phaseC_p0"}, pc
    assert pd == pb
    for s in range(len(shifts)):
        assert np.array_equal(d[s], b[s])
        if s == 0:  # deferring X_0 (either form) changes it by rounding only
            assert rel_err(a[0], b[0]) < 1e-13 and rel_err(a[0], c[0]) < 1e-13
        else:       # ... and the grouping depth never changes the iterates
            assert np.array_equal(a[s], b[s]) and np.array_equal(a[s], c[s])
    # and a budget below the base plan is an error at begin, not a crash in the first iteration
    monkeypatch.setenv("BCG_DEBUG_FIELD_BUDGET", str(int(11.5 * field)))
    ctx = bc.Context(dims)
    D = bc.dirac_op(ctx, mass, seed=11)
    B = bc.block_fermion_field(ctx, m).setRandom(seed=12)
    X = [bc.block_fermion_field(ctx, m) for _ in shifts]
    with pytest.raises(bc.BlockCGError) as e:
        bc.SBCGrQState(X, B, D, shifts, 0.0, 0.0)
    assert e.value.code == 3 and "out of memory" in str(e.value)  # BCG_ERR_HIP
    del X, B, D
    ctx.close()


def test_profile_reset_clears_bytes_and_flops(bc):
    ctx = bc.Context([16, 8, 8, 8])
    ctx.profiling(True)
    D = bc.dirac_op(ctx, 0.2, seed=1)
    B = bc.block_fermion_field(ctx, 16).setRandom(seed=2)
    X = [bc.block_fermion_field(ctx, 16) for _ in range(2)]
    seen = []
    for _ in range(2):
        ctx.profile_reset()
        st = bc.SBCGrQState(X, B, D, [0.0, 0.1], 0.0, 0.0)
        st.iterate(4)
        st.end()
        p = ctx.profile()
        seen.append({k: (v["count"], v["bytes"], v["flops"]) for k, v in p.items() if k.startswith("phase")})
    assert seen[0] == seen[1] and any(v[2] > 0 for v in seen[0].values())
    ctx.close()
