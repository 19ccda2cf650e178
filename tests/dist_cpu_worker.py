"""Worker for tests/test_distributed_cpu.py: one rank of a gloo process group on the CPU.

Checks the host side of the multi-GPU path -- the process grid, the library's halo message plan
(bcg_halo_plan) and the exchange routine bench.py uses (blockcg_amd.comm.exchange_messages, gloo branch)
-- by moving faces of a seeded global field between ranks and comparing every ghost site with the global
field, then applies the n-D operator on the ghost-extended sub-lattice (numpy) and compares with the
oracle's periodic operator on the whole lattice, and all-reduces a Gram matrix.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from blockcg_amd.comm import coords_of, exchange_messages, grid_for, halo_plan  # noqa: E402


def local_slices(L, origin):
    return tuple(slice(o, o + l) for o, l in zip(origin, L))


def main():
    gdims = [int(x) for x in os.environ["BCG_TEST_DIMS"].split(",")]
    m = int(os.environ["BCG_TEST_M"])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    nd = len(gdims)
    grid = grid_for(world, nd)
    coords = coords_of(rank, grid)
    L = [g // p for g, p in zip(gdims, grid)]
    origin = [c * l for c, l in zip(coords, L)]
    V = int(np.prod(gdims))
    orc = oracle.Oracle()
    # global field [x3,x2,x1,x0, m, 3] (lexicographic, x0 fastest), identical on every rank
    psi = orc.fill_field(m, V, 7).reshape(gdims[::-1] + [m, 3])
    loc = psi[local_slices(L, origin)[::-1]]
    site_bytes = 48 * m
    msgs, ghost_sites = halo_plan(gdims, grid, coords, site_bytes)
    assert len(msgs) == 2 * sum(1 for g in grid if g > 1)
    send = torch.zeros(ghost_sites * site_bytes, dtype=torch.uint8)
    recv = torch.zeros_like(send)
    # pack: per split direction [low face x_mu = 0][high face x_mu = L-1], other coordinates lexicographic
    off = 0
    split = [mu for mu in range(nd) if grid[mu] > 1]
    for mu in split:
        ax = nd - 1 - mu
        for xm in (0, L[mu] - 1):
            face = np.ascontiguousarray(np.take(loc, xm, axis=ax))
            b = torch.from_numpy(face.view(np.uint8).reshape(-1).copy())
            send[off:off + b.numel()] = b
            off += b.numel()
    assert off == send.numel()
    exchange_messages(send, recv, msgs, None, direct=False)
    # check every ghost face against the global field
    off = 0
    ghosts = {}
    for mu in split:
        ax = nd - 1 - mu
        for side, gx in (("minus", (origin[mu] - 1) % gdims[mu]), ("plus", (origin[mu] + L[mu]) % gdims[mu])):
            idx = list(local_slices(L, origin))
            idx[mu] = slice(gx, gx + 1)
            want = np.ascontiguousarray(np.take(psi[tuple(idx[::-1])], 0, axis=ax))
            n = want.size * 16
            got = recv[off:off + n].numpy().view(np.complex128).reshape(want.shape)
            assert np.array_equal(got, want), (rank, mu, side)
            ghosts[(mu, side)] = got
            off += n
    # operator on the ghost-extended local block vs the oracle's periodic operator on the whole lattice
    U = orc.fill_gauge(gdims, 9).reshape(gdims[::-1] + [nd, 3, 3])  # [.., mu, k, r] = U(r,k)
    want = orc.hop(U.reshape(V, nd, 3, 3), gdims, psi.reshape(V, m, 3)).reshape(gdims[::-1] + [m, 3])
    want = want[local_slices(L, origin)[::-1]]
    Uloc = U[local_slices(L, origin)[::-1]]
    acc = np.zeros_like(loc)
    gx = np.meshgrid(*[np.arange(o, o + l) for o, l in zip(origin[::-1], L[::-1])], indexing="ij")[::-1]  # gx[mu]
    for mu in range(nd):
        ax = nd - 1 - mu
        eta = (-1.0) ** sum(gx[nu] for nu in range(mu)) if mu > 0 else np.ones(L[::-1])
        if grid[mu] > 1:
            fwd = np.concatenate([np.take(loc, range(1, L[mu]), axis=ax), np.expand_dims(ghosts[(mu, "plus")], ax)], axis=ax)
            bwd = np.concatenate([np.expand_dims(ghosts[(mu, "minus")], ax), np.take(loc, range(0, L[mu] - 1), axis=ax)], axis=ax)
            idx = list(local_slices(L, origin))
            gxm = (origin[mu] - 1) % gdims[mu]
            idx[mu] = slice(gxm, gxm + 1)
            Ub_ghost = U[tuple(idx[::-1])][..., mu, :, :]
            Ub = np.concatenate([Ub_ghost, np.take(Uloc[..., mu, :, :], range(0, L[mu] - 1), axis=ax)], axis=ax)
        else:
            fwd = np.roll(loc, -1, axis=ax)
            bwd = np.roll(loc, 1, axis=ax)
            Ub = np.roll(Uloc[..., mu, :, :], 1, axis=ax)
        Uf = Uloc[..., mu, :, :]
        # out(r, j) = sum_k U(r,k) f(k, j) - conj(Ub(k,r)) b(k, j);  arrays are [.., j, c] and [.., k, r]
        t = np.einsum("...kr,...jk->...jr", Uf, fwd) - np.einsum("...rk,...jk->...jr", np.conj(Ub), bwd)
        acc += eta[..., None, None] * t
    got = 0.5 * acc
    err = np.linalg.norm(got - want) / np.linalg.norm(want)
    assert err < 1e-13, (rank, err)
    # Gram all-reduce: partial sums over sub-lattices add up to the global Gram matrix
    a = loc.reshape(-1, m, 3)
    G = torch.from_numpy(np.einsum("xic,xjc->ij", np.conj(a), a).copy())
    dist.all_reduce(G)
    Gw = orc.hermitian_dot(psi.reshape(V, m, 3), psi.reshape(V, m, 3))
    assert np.linalg.norm(G.numpy() - Gw) / np.linalg.norm(Gw) < 1e-13
    dist.barrier()
    if rank == 0:
        print("DIST_CPU_OK", world, grid)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
