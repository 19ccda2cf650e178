#!/usr/bin/env python3
"""Round 5: what retiring k_hop_fast (the general form of the m = 8 / 16 / 32 stencil) costs the lattices that used to run it --
fewer than four dimensions, or L0 not a multiple of the tile -- which now run k_hop_generic like every other width.
Usage (GPU box): BCG_LIB=<library with k_hop_fast> python tools/r05_general_stencil_ab.py   and without BCG_LIB."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import blockcg_amd as bc  # noqa: E402

for dims, m in (([1 << 22], 16), ([256, 256, 64], 16), ([60, 64, 64, 32], 16), ([1 << 22], 8), ([1 << 21], 32)):
    ctx = bc.Context(dims)
    D = bc.dirac_op(ctx, 1e-3, seed=1)
    B = bc.block_fermion_field(ctx, m).setRandom(seed=2)
    X = [bc.block_fermion_field(ctx, m)]
    st = bc.SBCGrQState(X, B, D, [0.0], 0.0, 0.0, consume_B=True)
    st.iterate(2)
    ctx.profiling(True)
    st.iterate(6)
    prof = ctx.profile()
    st.end()
    V = ctx.V
    gb = {k: v["ms"] / v["count"] for k, v in prof.items() if k.startswith("hop") or k.startswith("gram")}
    a1 = V * (2 * 48 * m + 144 * len(dims)) / 1e9
    print("x".join(map(str, dims)), f"m={m}", {k: round(v, 3) for k, v in gb.items()},
          f"A1 {a1 / gb['hop']:.0f} GB/s" if "hop" in gb else "")
    del st, X, B, D
    ctx.close()
