"""Worker for tests/test_distributed_gpu.py: the ranks of a domain-decomposed SBCGrQ run as THREADS of one process.

The BASELINE headline runs 8 ranks on a (2,2,2,1) process grid -- three divided directions, each two ranks wide, so that
the plus and the minus neighbour are the same peer in all three -- with the capacity ring in its overlapped form.  A one-GPU
test box admits at most six processes on its card, so this grid cannot be rehearsed with one process per rank there; here
every rank is a thread with a context, a stream and a native transport (libblockcg_rccl_mock.so: comm_rccl.cpp unchanged
over the stand-in of tests/cpp/mock_rccl.hpp) of its own.  The stand-in runs in its synchronous mode (BCG_MOCK_SYNC=1: a
call returns when its bytes have arrived; blocking host functions of eight streams in one process could otherwise share
one runtime callback thread) -- the asynchronous begin/end choreography is what the 2- and 4-process tests exercise; this
one is about the message plan, the ghost layout and the ring windows of the headline's grid.  Every rank checks operator,
Gram matrix and a fixed-work solve of its sub-lattice against the CPU oracle on the whole lattice."""
import os
import sys
import threading
import traceback

import numpy as np
import torch  # noqa: F401  (first: one HIP runtime per process, see tests/conftest.py)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import blockcg_amd as bc  # noqa: E402
import oracle  # noqa: E402
from blockcg_amd import rccl  # noqa: E402
from blockcg_amd.comm import coords_of  # noqa: E402


def main():
    gdims = [int(x) for x in os.environ["BCG_TEST_DIMS"].split(",")]
    grid = [int(x) for x in os.environ["BCG_TEST_GRID"].split(",")]
    m = int(os.environ["BCG_TEST_M"])
    ring = int(os.environ.get("BCG_TEST_RING", "0"))
    assert os.environ.get("BCG_MOCK_SYNC") == "1" and "mock" in os.path.basename(rccl.LIB_PATH)
    world = int(np.prod(grid))
    mass, shifts, iters = 0.1, [0.0, 1e-3, 1e-1], 4
    orc = oracle.Oracle()
    orc.set_threads(8)
    V = int(np.prod(gdims))
    U = orc.fill_gauge(gdims, 3)
    Bh = orc.fill_field(m, V, 4)
    want_op = orc.dirac_apply(U, gdims, mass, Bh)
    want_G = orc.hermitian_dot(Bh, want_op)
    o = orc.sbcgrq(U, gdims, mass, Bh, shifts, 0.0, 0.0, max_iterations=iters, trace_limit=iters)
    half = os.environ.get("BCG_TEST_HALF") == "1"  # and the half-volume fields on this grid (dist_gpu_worker.half_volume_checks)
    if half:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from dist_gpu_worker import half_volume_checks, half_volume_oracle
        half_pre = half_volume_oracle(orc, U, Bh, gdims, mass)
    orc.set_threads(1)
    uid = rccl.get_unique_id()
    errors, notes = [None] * world, [None] * world

    def rel(a, b):
        return np.linalg.norm(a - b) / np.linalg.norm(b)

    def rank_main(rank):
        try:
            coords = coords_of(rank, grid)
            ctx = bc.Context(gdims, device=0, grid=grid, coords=coords)
            comm = rccl.RcclComm(ctx, uid, rank, world)
            assert comm.communicators == 2
            ctx.capacity_mode(ring)
            D = bc.dirac_op(ctx, mass, seed=3)
            B = bc.block_fermion_field(ctx, m).setRandom(seed=4)
            out = bc.block_fermion_field(ctx, m)
            ctx.profiling(True)
            D.op(out, B)
            prof = ctx.profile()
            ctx.profiling(False)
            L, og = ctx.local_dims, ctx.origin
            sl = tuple(slice(a, a + b) for a, b in zip(og, L))[::-1]

            def local(a):
                return np.ascontiguousarray(a.reshape(gdims[::-1] + [m, 3])[sl]).reshape(-1, m, 3)

            assert np.array_equal(B.download(), local(Bh))
            e_op = rel(out.download(), local(want_op))
            assert e_op < 2e-13, ("op", rank, e_op)
            if ring:
                C = (ring - 2) // 2
                chunks = (L[3] + C - 1) // C
                # the source's faces in two windows (the slices the first launches read, then the rest), the tmp faces per chunk:
                # every exchange in the split form, none blocking
                p_windows = 2 if C + 1 < L[3] - 1 else 1
                assert prof["halo_exchange_begin"]["count"] == chunks + p_windows == prof["halo_exchange_end"]["count"], prof
                assert "hop_ring" in prof and "halo_exchange" not in prof, sorted(prof)
            assert rel(B.hermitian_dot(out), want_G) < 1e-13
            X = [bc.block_fermion_field(ctx, m) for _ in shifts]
            info = bc.SBCGrQ(X, B, D, shifts, 0.0, 0.0, max_iterations=iters, trace_limit=iters, return_info=True)
            for key in ("alpha", "rho", "delta", "alpha_s", "beta_s"):
                assert rel(info["trace"][key], o["trace"][key]) < 1e-10, (key, rank)
            for s in range(len(shifts)):
                e = rel(X[s].download(), local(o["X"][s]))
                assert e < 1e-11, ("X", s, rank, e)
            if half:
                half_volume_checks(ctx, comm, orc, D, B, U, Bh, gdims, mass, m, local, rel, pre=half_pre)
            notes[rank] = (e_op, {k[len("stencil_form_"):]: v["count"] for k, v in prof.items() if k.startswith("stencil_form_")})
            comm.barrier()
            comm.close()
            del X, out, B, D
            ctx.close()
        except BaseException:  # noqa: BLE001 -- reported by the main thread
            errors[rank] = traceback.format_exc()

    threads = [threading.Thread(target=rank_main, args=(r,), name=f"rank{r}") for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=900)
    hung = [t.name for t in threads if t.is_alive()]
    bad = [(r, e) for r, e in enumerate(errors) if e]
    if hung or bad:
        for r, e in bad:
            print(f"--- rank {r} ---\n{e}", file=sys.stderr)
        print("hung:", hung, file=sys.stderr, flush=True)
        os._exit(1)
    print("DIST_THREADS_OK", world, grid, "ring", ring, "op err %.2e" % max(n[0] for n in notes), notes[0][1])


if __name__ == "__main__":
    main()
