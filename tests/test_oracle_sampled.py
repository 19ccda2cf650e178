"""CPU checks of the oracle machinery behind tests/test_fullsize_parity.py: the sampled evaluator equals the
whole-lattice restatement, the multi-threaded form equals the sequential one to rounding, and the replication
property the full-size solver test relies on holds in the oracle itself."""
import numpy as np

from conftest import rel_err


def test_sampled_evaluator_equals_whole_lattice_oracle(orc):
    for dims, m in (([8, 6, 4, 6], 16), ([4, 4, 2, 2], 8), ([6, 2, 4, 2], 32), ([12], 3)):
        V = int(np.prod(dims))
        U = orc.fill_gauge(dims, 3)
        P = orc.fill_field(m, V, 4)
        sites = np.arange(V)
        assert np.array_equal(orc.hop_sampled(m, dims, 3, 4, sites), orc.hop(U, dims, P))  # same expression, bit for bit
        a = orc.dirac_apply(U, dims, 0.3, P)
        assert np.abs(orc.apply_sampled(m, dims, 3, 4, 0.3, sites) - a).max() < 4e-15 * np.abs(a).max()
        sub = np.array([V - 1, 0, V // 2])
        assert np.array_equal(orc.hop_sampled(m, dims, 3, 4, sub), orc.hop(U, dims, P)[sub])


def test_threaded_oracle_matches_sequential(orc):
    dims, m, shifts = [8, 8, 8, 16], 8, [0.0, 0.01]   # 8192 sites: two Gram chunks
    g1 = orc.sbcgrq_generated(m, dims, 5, 6, 0.4, shifts, 3)
    orc.set_threads(4)
    try:
        g4 = orc.sbcgrq_generated(m, dims, 5, 6, 0.4, shifts, 3)
        sites = np.arange(0, 8192, 37)
        h4 = orc.hop_sampled(m, dims, 5, 6, sites)
    finally:
        orc.set_threads(1)
    assert np.array_equal(h4, orc.hop_sampled(m, dims, 5, 6, sites))
    for k in ("alpha", "rho", "delta", "alpha_s", "beta_s"):
        assert rel_err(g4[k], g1[k]) < 1e-12, k
    U = orc.fill_gauge(dims, 5)
    B = orc.fill_field(m, int(np.prod(dims)), 6)
    o = orc.sbcgrq(U, dims, 0.4, B, shifts, 0.0, 0.0, 3, trace_limit=3)
    assert np.array_equal(o["trace"]["alpha"], g1["alpha"])


def test_replicated_lattice_property(orc):
    """Tiling a base problem r times per direction leaves alpha, rho, beta_s unchanged, scales delta and alpha_s by
    sqrt(copies), keeps the iteration count and tiles the solution."""
    from test_fullsize_parity import tile_sites
    base, reps, m, shifts, mass = [4, 4, 2, 4], [2, 1, 2, 2], 4, [0.0, 0.05], 0.5
    dims = [b * r for b, r in zip(base, reps)]
    Ub = orc.fill_gauge(base, 11)
    Bb = orc.fill_field(m, int(np.prod(base)), 12)
    o = orc.sbcgrq(Ub, base, mass, Bb, shifts, 1e-9, 1e-9, 500, trace_limit=3)
    f = orc.sbcgrq(tile_sites(Ub, base, reps), dims, mass, tile_sites(Bb, base, reps), shifts, 1e-9, 1e-9, 500, trace_limit=3)
    assert abs(f["iterations"] - o["iterations"]) <= 1
    sc = np.sqrt(np.prod(reps))
    for k in ("alpha", "rho", "beta_s"):
        assert rel_err(f["trace"][k], o["trace"][k]) < 1e-11, k
    for k in ("delta", "alpha_s"):
        assert rel_err(f["trace"][k], sc * o["trace"][k]) < 1e-11, k
    for s in range(len(shifts)):
        assert rel_err(f["X"][s], tile_sites(o["X"][s], base, reps)) < 1e-7


def test_chunked_generated_gram_equals_whole_field_oracle(orc):
    for m, V in ((16, 3 * 4096 + 17), (8, 100), (32, 5000), (3, 4096)):
        a = orc.fill_field(m, V, 5)
        b = orc.fill_field(m, V, 6)
        want = orc.hermitian_dot(a, b)
        for thr in (1, 4):
            orc.set_threads(thr)
            try:
                got = orc.gram_generated(m, V, 5, 6)
            finally:
                orc.set_threads(1)
            assert rel_err(got, want) < 1e-13
