#!/usr/bin/env python3
"""GPU box: does k_phaseC's time depend on WHERE its fields lie?  The kernel's time differs by 7 % between processes
(11.0 vs 11.8 ms at 64^4, m = 16, one shift) and is stable inside one.  This runs the same one-shift solve several times in
ONE process, each time with a different amount of device memory allocated in front of the solver's fields, and for each
per-field address stagger given (the fields are exactly 12 GiB, back to back).
usage: phasec_spread.py [iterations [stagger ...]]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import blockcg_amd as bc  # noqa: E402

dims, m = [64, 64, 64, 64], 16
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
staggers = [int(a) for a in sys.argv[2:]] or [0]  # BCG_FIELD_STAGGER values to sample (bytes, multiples of 256)
for stagger, pad in [(s, p) for s in staggers for p in [0, 1, 2, 3, 0, 5, 7, 0]]:
    os.environ["BCG_FIELD_STAGGER"] = str(stagger)
    ctx = bc.Context(dims)
    dummies = [bc.block_fermion_field(ctx, 1) for _ in range(pad)]  # 0.8 GB each
    D = bc.dirac_op(ctx, 1e-3, seed=1)
    B = bc.block_fermion_field(ctx, m).setRandom(seed=2)
    X = [bc.block_fermion_field(ctx, m)]
    st = bc.SBCGrQState(X, B, D, [0.0], 0.0, 0.0, consume_B=True)
    st.iterate(2)
    ctx.profiling(True)
    ctx.profile_reset()
    st.iterate(iters)
    prof = ctx.profile()
    print(stagger, pad, {k: round(v["ms"] / v["count"], 3) for k, v in prof.items() if k in ("phaseC", "phaseB", "hop")}, flush=True)
    st.end()
    del st, X, B, D, dummies
    ctx.close()
