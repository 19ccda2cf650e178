#!/usr/bin/env python3
"""Profiling aid (GPU box): a few launches of the two stencil kernels at 64^4, m = 16 and nothing else.
Usage: python tools/hop_only.py [reps] [dims...]   (run it under rocprofv3 --pmc ...)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import blockcg_amd as bc  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dims = [int(x) for x in sys.argv[2:]] or [64, 64, 64, 64]
m = 16
ctx = bc.Context(dims)
D = bc.dirac_op(ctx, 1e-3, seed=1)
B = bc.block_fermion_field(ctx, m).setRandom(seed=2)
X = [bc.block_fermion_field(ctx, m)]
st = bc.SBCGrQState(X, B, D, [0.0], 0.0, 0.0, consume_B=True)  # one shift: phase A is the two stencil kernels of the bench
ctx.profiling(True)
st.iterate(reps)
prof = ctx.profile()
print({k: round(v["ms"] / v["count"], 3) for k, v in prof.items()})
st.end()
