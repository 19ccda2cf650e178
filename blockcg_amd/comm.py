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


def grid_for(world_size, ndim, keep_last=False):
    """Factor world_size into a process grid, halving over the slowest directions first.  keep_last leaves the last
    direction undivided (capacity mode sweeps it slice by slice)."""
    grid = [1] * ndim
    top = ndim - 2 if keep_last and ndim > 1 else ndim - 1
    n, mu = world_size, top
    while n > 1:
        if n % 2:
            raise ValueError("world size must be a power of two")
        grid[mu] *= 2
        n //= 2
        mu = mu - 1 if mu > 0 else top
    return grid


def grid_for_half(world_size, ndim):
    """The process grid of the half-volume ladder (bench.py --half): direction 0 -- the one half fields are compact in --
    and the last direction -- swept in chunks whose exchanges overlap the stencil -- stay whole; the others halve in turn,
    the slower one first: 2 -> (1,1,2,1), 4 -> (1,2,2,1), 8 -> (1,2,4,1)."""
    if ndim < 4:
        return grid_for(world_size, ndim)
    grid = [1] * ndim
    n, mu = world_size, ndim - 2
    while n > 1:
        if n % 2:
            raise ValueError("world size must be a power of two")
        grid[mu] *= 2
        n //= 2
        mu = mu - 1 if mu > 1 else ndim - 2
    return grid


def coords_of(rank, grid):
    c = []
    for g in grid:
        c.append(rank % g)
        rank //= g
    return c


def halo_plan(dims, grid, coords, site_bytes):
    """The library's message plan for one halo exchange (bcg_halo_plan; pure host code, no GPU needed).
    Returns (messages, ghost_sites); a message is (peer_send, peer_recv, send_offset, recv_offset, nbytes)."""
    lib = _lib.load()
    nd = len(dims)
    iv = lambda v: (ctypes.c_int * nd)(*[int(x) for x in v])  # noqa: E731
    ps, pr = (ctypes.c_int * 8)(), (ctypes.c_int * 8)()
    so, ro, nb = (ctypes.c_size_t * 8)(), (ctypes.c_size_t * 8)(), (ctypes.c_size_t * 8)()
    ghost = ctypes.c_int64(0)
    n = lib.bcg_halo_plan(nd, iv(dims), iv(grid), iv(coords), site_bytes, ps, pr, so, ro, nb, ctypes.byref(ghost))
    if n < 0:
        raise ValueError("invalid decomposition")
    return [(ps[k], pr[k], so[k], ro[k], nb[k]) for k in range(n)], ghost.value


def post_messages(send, recv, msgs, group=None):
    """Device-direct (RCCL) form split in two: post the batched sends/receives and return the work handles; the
    caller waits on them later (TorchDistComm._halo_end), so independent kernels enqueued in between overlap."""
    ops = []
    for ps, pr, so, ro, nb in msgs:
        ops.append(dist.P2POp(dist.isend, send[so:so + nb], ps, group))
        ops.append(dist.P2POp(dist.irecv, recv[ro:ro + nb], pr, group))
    return dist.batch_isend_irecv(ops)


def exchange_messages(send, recv, msgs, group=None, direct=False, sync=None):
    """Move the faces of one halo exchange.  send/recv: flat uint8 tensors (device memory when `direct`,
    i.e. backend nccl = RCCL; otherwise staged through host memory for gloo).  Messages to the same peer
    are matched by posting order (RCCL point-to-point has no tags); gloo also gets the message index as tag."""
    if direct:
        ops = []
        for ps, pr, so, ro, nb in msgs:
            ops.append(dist.P2POp(dist.isend, send[so:so + nb], ps, group))
            ops.append(dist.P2POp(dist.irecv, recv[ro:ro + nb], pr, group))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        return
    if sync is not None:
        sync()
    reqs, stage = [], []
    for k, (ps, pr, so, ro, nb) in enumerate(msgs):
        s = send[so:so + nb].cpu().contiguous()
        r = torch.empty(nb, dtype=torch.uint8)
        stage.append((ro, nb, r, s))
        reqs.append(dist.isend(s, ps, group=group, tag=k))
        reqs.append(dist.irecv(r, pr, group=group, tag=k))
    for w in reqs:
        w.wait()
    for ro, nb, r, _ in stage:
        recv[ro:ro + nb].copy_(r)
    if sync is not None:
        sync()


class _DevMem:
    """Expose library-owned device memory to torch without copying."""

    def __init__(self, ptr, nbytes, typestr="|u1", itemsize=1):
        self.__cuda_array_interface__ = {"shape": (nbytes // itemsize,), "typestr": typestr, "data": (ptr, False),
                                         "version": 3, "strides": None}


class TorchDistComm:
    """Owns the HIP stream the context enqueues on and implements bcg_comm with torch.distributed."""

    def __init__(self, device_index=0, group=None, overlap=True):
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
        self._begin_cb = _lib.HALO_CB(self._halo_begin)
        self._end_cb = _lib.HALO_END_CB(self._halo_end)
        self._pending = []
        self.overlap = overlap
        if overlap:
            self.struct = _lib.bcg_comm(None, self._halo_cb, self._allreduce_cb, self._begin_cb, self._end_cb)
        else:
            self.struct = _lib.bcg_comm(None, self._halo_cb, self._allreduce_cb, _lib.HALO_CB(), _lib.HALO_END_CB())

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
            if t.data_ptr() != ptr:  # must alias the library's buffer, never a copy
                raise RuntimeError("torch did not alias the library's device buffer")
            self._views = {k: v for k, v in self._views.items() if k[2] != f64 or k[0] != ptr}
            self._views[key] = t
        return t

    def _halo(self, user, n, peer_s, peer_r, off_s, off_r, nbytes):
        try:
            sp, rp, each = self.ctx.halo_buffers()
            send = self._view(sp, each)
            recv = self._view(rp, each)
            msgs = [(peer_s[k], peer_r[k], off_s[k], off_r[k], nbytes[k]) for k in range(n)]
            with torch.cuda.stream(self.stream):
                exchange_messages(send, recv, msgs, self.group, self.direct, sync=self.stream.synchronize)
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            self.error = e
            return 1

    def _halo_begin(self, user, n, peer_s, peer_r, off_s, off_r, nbytes):
        """Post the exchange; with RCCL the stream is not made to wait here (see _halo_end)."""
        try:
            sp, rp, each = self.ctx.halo_buffers()
            send = self._view(sp, each)
            recv = self._view(rp, each)
            msgs = [(peer_s[k], peer_r[k], off_s[k], off_r[k], nbytes[k]) for k in range(n)]
            with torch.cuda.stream(self.stream):
                if self.direct:
                    self._pending.append(post_messages(send, recv, msgs, self.group))
                else:  # host-staged transport has nothing to overlap: do it all now
                    exchange_messages(send, recv, msgs, self.group, False, sync=self.stream.synchronize)
                    self._pending.append([])
            return 0
        except Exception as e:
            self.error = e
            return 1

    def _halo_end(self, user):
        try:
            with torch.cuda.stream(self.stream):
                for w in self._pending.pop(0):  # exchanges end in the order they began (up to two outstanding)
                    w.wait()  # the context's stream waits for the exchange; the host does not block
            return 0
        except Exception as e:
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
