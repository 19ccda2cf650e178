"""torch.distributed plumbing for the multi-GPU path: the two bcg_comm callbacks.

The library packs halo faces into a device send buffer and needs the neighbours' faces in a device
receive buffer (halo_exchange), and needs m x m Gram partials summed over ranks (allreduce_sum).
With backend "nccl" (= RCCL on ROCm) both operate directly on device memory, point-to-point over
xGMI, ordered on the context's HIP stream.  With backend "gloo" (CPU tests, or several ranks
sharing one GPU) the same messages are staged through host memory.

Process grid: rank = lexicographic index of grid coordinates with direction 0 fastest (the library's
rank_of); `grid_for(world_size, ndim)` splits the slowest directions first so that packed faces are
contiguous or long-strided runs.
"""
import ctypes

import torch
import torch.distributed as dist

from . import _lib


def grid_for(world_size, ndim):
    """Factor world_size into a process grid, halving over the slowest directions first."""
    grid = [1] * ndim
    n, mu = world_size, ndim - 1
    while n > 1:
        if n % 2:
            raise ValueError("world size must be a power of two")
        grid[mu] *= 2
        n //= 2
        mu = mu - 1 if mu > 0 else ndim - 1
    return grid


def coords_of(rank, grid):
    c = []
    for g in grid:
        c.append(rank % g)
        rank //= g
    return c


class _DevMem:
    """Expose library-owned device memory to torch without copying."""

    def __init__(self, ptr, nbytes, typestr="|u1", itemsize=1):
        self.__cuda_array_interface__ = {"shape": (nbytes // itemsize,), "typestr": typestr, "data": (ptr, False),
                                         "version": 3, "strides": None}


class TorchDistComm:
    """Owns the HIP stream the context enqueues on and implements bcg_comm with torch.distributed."""

    def __init__(self, device_index=0, group=None):
        self.group = group
        self.device = torch.device("cuda", device_index)
        self.backend = dist.get_backend(group)
        self.direct = self.backend == "nccl"
        self.stream = torch.cuda.Stream(self.device)
        self.ctx = None
        self._views = {}
        self.error = None
        self._halo_cb = _lib.HALO_CB(self._halo)
        self._allreduce_cb = _lib.ALLREDUCE_CB(self._allreduce)
        self.struct = _lib.bcg_comm(None, self._halo_cb, self._allreduce_cb)

    @property
    def stream_ptr(self):
        return ctypes.c_void_p(self.stream.cuda_stream)

    def attach(self, ctx):
        self.ctx = ctx
        ctx.set_comm(self.struct, self)

    def _view(self, ptr, nbytes, f64=False):
        key = (ptr, nbytes, f64)
        t = self._views.get(key)
        if t is None:
            mem = _DevMem(ptr, nbytes, "<f8", 8) if f64 else _DevMem(ptr, nbytes)
            t = torch.as_tensor(mem, device=self.device)
            self._views = {k: v for k, v in self._views.items() if k[2] != f64 or k[0] != ptr}
            self._views[key] = t
        return t

    def _halo(self, user, n, peer_s, peer_r, off_s, off_r, nbytes):
        try:
            sp, rp, each = self.ctx.halo_buffers()
            send = self._view(sp, each)
            recv = self._view(rp, each)
            with torch.cuda.stream(self.stream):
                if self.direct:
                    ops = []
                    for k in range(n):
                        ops.append(dist.P2POp(dist.isend, send[off_s[k]:off_s[k] + nbytes[k]], peer_s[k], self.group))
                        ops.append(dist.P2POp(dist.irecv, recv[off_r[k]:off_r[k] + nbytes[k]], peer_r[k], self.group))
                    for w in dist.batch_isend_irecv(ops):
                        w.wait()
                else:
                    self.stream.synchronize()
                    reqs, stage = [], []
                    for k in range(n):
                        s = send[off_s[k]:off_s[k] + nbytes[k]].cpu()
                        r = torch.empty(nbytes[k], dtype=torch.uint8)
                        stage.append((k, r))
                        reqs.append(dist.isend(s, peer_s[k], group=self.group, tag=k))
                        reqs.append(dist.irecv(r, peer_r[k], group=self.group, tag=k))
                    for w in reqs:
                        w.wait()
                    for k, r in stage:
                        recv[off_r[k]:off_r[k] + nbytes[k]].copy_(r)
                    self.stream.synchronize()
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            self.error = e
            return 1

    def _allreduce(self, user, buf, count):
        try:
            t = self._view(buf, count * 8, f64=True)
            with torch.cuda.stream(self.stream):
                if self.direct:
                    dist.all_reduce(t, group=self.group)
                else:
                    self.stream.synchronize()
                    h = t.cpu()
                    dist.all_reduce(h, group=self.group)
                    t.copy_(h)
                    self.stream.synchronize()
            return 0
        except Exception as e:
            self.error = e
            return 1
