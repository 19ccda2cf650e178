#!/usr/bin/env python3
"""Tuning aid (GPU box): how far apart in x3 are the 64 blocks of one XCD while the plain stencil runs?
Needs a library built with -DBCG_HOP4_TRACE (tools/build_variant.sh trace "-DBCG_HOP4_TRACE"; BCG_LIB=...)."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import blockcg_amd as bc  # noqa: E402

dims, m = [64, 64, 64, 64], 16
ctx = bc.Context(dims)
D = bc.dirac_op(ctx, 1e-3, seed=1)
B = bc.block_fermion_field(ctx, m).setRandom(seed=2)
X = [bc.block_fermion_field(ctx, m)]
st = bc.SBCGrQState(X, B, D, [0.0], 0.0, 0.0, consume_B=False)
st.iterate(1)  # allocates the scratch buffer
y = bc.block_fermion_field(ctx, m)
D.D(y, B)
D.D(y, B)
nblk, steps = 512, 2048
buf = np.zeros(nblk * 4096, dtype=np.int64)
lib = ctx.lib
lib.bcg_debug_read_scratch.restype = ctypes.c_int
lib.bcg_debug_read_scratch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
rc = lib.bcg_debug_read_scratch(ctx.h, buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes)
assert rc == 0, rc
tr = buf.reshape(nblk, 2048, 2)
t = tr[:, :, 0].astype(np.float64)
site = tr[:, :, 1]
t0 = t.min()
t = (t - t0) * 0.01  # microseconds (100 MHz counter)
print("kernel span %.1f us; first-tile start spread %.1f us" % (t.max(), t[:, 0].max() - t[:, 0].min()))
S3 = 64 ** 3
x3 = site // S3
for cls in (0, 3):
    blocks = np.arange(cls, nblk, 8)
    tc = t[blocks]          # [64, steps]
    dur = np.diff(tc, axis=1)
    print("class %d: step duration us: median %.2f p10 %.2f p90 %.2f max %.1f" % (cls, np.median(dur), np.percentile(dur, 10), np.percentile(dur, 90), dur.max()))
    # progress of each block at sample times
    samples = np.linspace(tc[:, 1].max(), tc[:, -1].min(), 400)
    prog = np.stack([np.searchsorted(tc[b], samples) for b in range(len(blocks))])  # [64, 400] steps started
    spread = prog.max(axis=0) - prog.min(axis=0)
    print("  spread of step index over the 64 blocks: median %d p90 %d max %d (steps; one step = one x3 slice)" % (np.median(spread), np.percentile(spread, 90), spread.max()))
    sd = prog.std(axis=0)
    print("  std of step index: median %.1f" % np.median(sd))
    print("  x3 of the blocks at mid-kernel:", sorted(x3[blocks, steps // 2].tolist())[:8], "...")
