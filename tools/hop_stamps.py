#!/usr/bin/env python3
"""Tuning aid (GPU box): where the cycles of a stencil tile go (plain form), from in-kernel s_memtime stamps.
k_hop4c: library built with -DBCG_HOP4C_STAMPS and BCG_HOP_BUNDLE=0; k_hop4b (the default form): -DBCG_HOP4B_STAMPS and
`python tools/hop_stamps.py 4b`  (tools/build_variant.sh stamps "-DBCG_HOP4B_STAMPS"; BCG_LIB=...)."""
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
for _ in range(3):
    D.D(y, B)
ctx.synchronize()
nblk = 512
NSEG = 16 if len(sys.argv) > 1 and sys.argv[1] in ("4b", "pipe") else 8
buf = np.zeros(nblk * 4 * NSEG, dtype=np.float64)
lib = ctx.lib
lib.bcg_debug_read_scratch.restype = ctypes.c_int
lib.bcg_debug_read_scratch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
assert lib.bcg_debug_read_scratch(ctx.h, buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
seg = buf.reshape(nblk, 4, NSEG)
tiles = 64 ** 4 // 16 // nblk
names = ["park+pace", "barrier", "issue(links,dir0)", "dir0", "dir1", "dir2", "dir3(+x3)", "tail(p,store)"]
if len(sys.argv) > 1 and sys.argv[1] == "4b":
    names = ["pace(thread 0)", "barrier", "issue(DMAs,loads)", "dir0 (LDS only)", "dir1 (+wait loads)", "dir2", "dir3", "tail(park,store)"]
if len(sys.argv) > 1 and sys.argv[1] == "pipe":  # the software-pipelined step (PIPE in hop4b_body)
    names = ["loop overhead", "barrier", "-x3 readback + row DMAs", "dir0", "issue p + link DMAs", "dir1", "issue next rows", "dir2",
             "WAIT +x3 row", "dir3", "carry U3, WAIT p, output, stores", "WAIT links + next rows"]
tot = seg.sum(axis=2)
print("cycles per tile per wave (s_memtime ticks = shader cycles): total median %.0f" % np.median(tot / tiles))
for i, n in enumerate(names):
    v = seg[:, :, i] / tiles
    print("  %-20s median %7.0f  p10 %7.0f  p90 %7.0f  share %4.1f %%" % (n, np.median(v), np.percentile(v, 10), np.percentile(v, 90),
                                                                       100 * v.sum() / (tot / tiles).sum()))
