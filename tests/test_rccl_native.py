"""The native RCCL transport (libblockcg_rccl.so, include/blockcg_rccl.h) on the one GPU a test box has: a communicator of
one rank.  What can be checked there: the communicator initialises on the context's device and installs its callbacks;
the all-reduce, barrier and max run on the context's stream; the grouped ncclSend/ncclRecv of the blocking and of the split
(begin/end, second stream + events) exchange move the right bytes between the library's halo buffers when every peer is
this rank itself; a process grid that disagrees with the world size is reported as BCG_ERR_COMM, not a hang.  Two real
ranks need two GPUs (RCCL refuses two ranks on one device): the driver's multi-GPU bench is the first place that runs."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _view(torch, ptr, nbytes):
    from blockcg_amd.comm import _DevMem
    t = torch.as_tensor(_DevMem(ptr, nbytes), device="cuda:0")
    assert t.data_ptr() == ptr
    return t


def test_world_of_one_native_rccl():
    import torch
    import blockcg_amd as bc
    from blockcg_amd import rccl
    uid = rccl.get_unique_id()
    assert len(uid) == rccl.UNIQUE_ID_BYTES
    # rank 0 of a (1,1,1,2) process grid: the context is "distributed" (it packs faces and sizes halo buffers) ...
    dims, m = [16, 4, 4, 8], 16
    ctx = bc.Context(dims, grid=[1, 1, 1, 2], coords=[0, 0, 0, 0])
    comm = rccl.RcclComm(ctx, uid, rank=0, world=1)
    cb = comm.callbacks
    assert cb.halo_exchange and cb.allreduce_sum and cb.halo_exchange_begin and cb.halo_exchange_end
    assert comm.max(3.25) == 3.25
    comm.barrier()
    D = bc.dirac_op(ctx, 0.1, seed=3)
    x = bc.block_fermion_field(ctx, m).setRandom(seed=4)
    y = bc.block_fermion_field(ctx, m)
    # ... but its x3 neighbour is rank 1, which a communicator of one rank does not have: a clean error, no hang
    with pytest.raises(bc.BlockCGError) as e:
        D.D(y, x)
    assert e.value.code == 5  # BCG_ERR_COMM
    assert "outside the communicator" in comm.last_error()
    # the halo buffers exist now; exchange with myself through the callbacks, as the library would call them
    sp, rp, each = ctx.halo_buffers()
    assert each > 0
    send, recv = _view(torch, sp, each), _view(torch, rp, each)
    rng = torch.Generator(device="cuda:0").manual_seed(5)
    half = (each // 2) // 16 * 16
    for split in (False, True):
        send.copy_(torch.randint(0, 255, (each,), dtype=torch.uint8, device="cuda:0", generator=rng))
        recv.zero_()
        torch.cuda.synchronize()
        n = 2
        peers = (ctypes.c_int * n)(0, 0)
        off_s = (ctypes.c_size_t * n)(0, half)
        off_r = (ctypes.c_size_t * n)(half, 0)     # crossed, like message 2k / 2k+1 of bcg_halo_plan
        nb = (ctypes.c_size_t * n)(half, half)
        if split:
            assert cb.halo_exchange_begin(cb.user, n, peers, peers, off_s, off_r, nb) == 0, comm.last_error()
            assert cb.halo_exchange_end(cb.user) == 0
        else:
            assert cb.halo_exchange(cb.user, n, peers, peers, off_s, off_r, nb) == 0, comm.last_error()
        ctx.synchronize()
        assert torch.equal(recv[half:2 * half], send[:half]) and torch.equal(recv[:half], send[half:2 * half])
    # out-of-range message is refused before RCCL sees it
    bad = (ctypes.c_size_t * 1)(each)
    one = (ctypes.c_int * 1)(0)
    zero = (ctypes.c_size_t * 1)(0)
    assert cb.halo_exchange(cb.user, 1, one, one, zero, zero, bad) == 0  # exactly the buffer: fine
    bad[0] = each + 16
    assert cb.halo_exchange(cb.user, 1, one, one, zero, zero, bad) != 0
    # all-reduce of the Gram buffer: identity in a world of one, on the context's stream
    g = torch.arange(512, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    assert cb.allreduce_sum(cb.user, ctypes.c_void_p(g.data_ptr()), 512) == 0
    ctx.synchronize()
    assert torch.equal(g, torch.arange(512, dtype=torch.float64, device="cuda:0"))
    comm.close()


def test_undivided_solve_with_native_comm_installed():
    """With a (1,1,1,1) grid nothing communicates; installing the transport must not change a solve."""
    import blockcg_amd as bc
    from blockcg_amd import rccl
    dims, m, shifts = [16, 4, 4, 4], 16, [0.0, 0.01]
    out = []
    for native in (False, True):
        ctx = bc.Context(dims)
        comm = rccl.RcclComm(ctx, rccl.get_unique_id(), 0, 1) if native else None
        D = bc.dirac_op(ctx, 0.3, seed=7)
        B = bc.block_fermion_field(ctx, m).setRandom(seed=8)
        X = [bc.block_fermion_field(ctx, m) for _ in shifts]
        it = bc.SBCGrQ(X, B, D, shifts, 1e-10, 1e-10)
        out.append((it, np.stack([x.download() for x in X])))
        if comm:
            comm.close()
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])


def test_headline_share_fits_the_device():
    """Rank 0 of the BASELINE headline (V = 128^4 on a (2,2,2,1) grid, m = 16, 4 shifts, capacity ring 32 as bench.py selects it, source consumed):
    the library's own memory plan for one SBCGrQ solve stays under the 288 GiB of one MI355X with room for the runtime and
    RCCL's buffers; without the ring it does not."""
    import blockcg_amd as bc
    ctx = bc.Context([128] * 4, grid=[2, 2, 2, 1], coords=[0, 0, 0, 0])
    assert ctx.local_dims == [64, 64, 64, 128]
    whole = ctx.sbcgrq_device_bytes(16, 4, consume_B=True)
    ctx.capacity_mode(32)
    ring = ctx.sbcgrq_device_bytes(16, 4, consume_B=True)
    hbm = 288 * 2**30
    assert ring < hbm - 16 * 2**30, ring / 2**30      # >= 16 GiB of headroom
    assert whole > hbm - 4 * 2**30, whole / 2**30     # the whole-field plan leaves (next to) nothing
    assert whole - ring > 19e9


def test_interior_grid_cap_does_not_change_results(monkeypatch):
    """bcg_overlap_tuning / BCG_HOP_BLOCKS_OVERLAP: a smaller interior-class grid (448 of 512 blocks, 7 of the 8 blocks of
    every XCD group of a CU pair) leaves compute units to RCCL's kernels during the split exchange; the tiles are merely
    dealt to fewer blocks, so the operator is bit-identical and the pacing counters still complete."""
    import blockcg_amd as bc
    monkeypatch.setenv("BCG_FORCE_TILE_CLASSES", "1")  # take the interior/boundary launch pair on one GPU
    dims, m = [64, 64, 64, 16], 16
    outs = []
    for blocks in (0, 448, 256):
        ctx = bc.Context(dims)
        assert ctx.lib.bcg_overlap_tuning(ctx.h, blocks) == 0
        D = bc.dirac_op(ctx, 0.2, seed=3)
        x = bc.block_fermion_field(ctx, m).setRandom(seed=4)
        y = bc.block_fermion_field(ctx, m)
        D.op(y, x)
        X = [bc.block_fermion_field(ctx, m)]
        info = bc.SBCGrQ(X, x, D, [0.0], 0.0, 0.0, max_iterations=3, trace_limit=3, return_info=True)
        sites = np.arange(0, ctx.V, 4099)
        outs.append((y.download_sites(sites), info["trace"]["alpha"], X[0].download_sites(sites)))
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0])
        assert np.allclose(o[1], outs[0][1], rtol=1e-12, atol=0) and np.allclose(o[2], outs[0][2], rtol=1e-11, atol=1e-14)
