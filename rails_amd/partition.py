"""Row partition of the hot path over the GPUs of one node.

Rows of A, V, AV, B are split in contiguous blocks, one block per rank (one process per GPU).  The only
exchanges are (1) sum-all-reduces of the small projected blocks / Lanczos sums and (2) the ghost rows
of W needed by the local rows of A.  Both go through torch.distributed (backend "nccl" = RCCL over xGMI
on the GPUs, "gloo" in the CPU tests): the C library calls back into the hooks installed here with
device (or host) pointers and the stream it is working on.

This module is host logic only and works without a GPU (the CPU tests drive it with the oracle).
"""
import ctypes as C

import numpy as np


def row_ranges(m_global, nranks):
    """Contiguous, balanced row blocks: rank r owns [starts[r], starts[r+1])."""
    base, rem = divmod(m_global, nranks)
    starts = np.zeros(nranks + 1, dtype=np.int64)
    for r in range(nranks):
        starts[r + 1] = starts[r] + base + (1 if r < rem else 0)
    return starts


class HaloPlan:
    """Ghost-row plan for a local CSR block whose column indices are GLOBAL.

    After construction:
      col_local      column indices remapped to [0, m_local) for own rows and m_local + g for ghost g
      ghost_globals  sorted global indices of the ghost rows (grouped by owner rank, ranks ascending)
      recv_counts[r] number of ghost rows owned by rank r
      send_rows      local row indices to pack, grouped by destination rank (ranks ascending)
      send_counts[r] number of rows sent to rank r
    `exchange(obj_list)` must be an all-to-all of python objects (dist.all_to_all / all_gather based).
    """

    def __init__(self, starts, rank, col_global, all_gather_object):
        starts = np.asarray(starts, dtype=np.int64)
        nranks = starts.size - 1
        self.rank, self.nranks = rank, nranks
        r0, r1 = starts[rank], starts[rank + 1]
        self.m_local = int(r1 - r0)
        col_global = np.asarray(col_global, dtype=np.int64)
        own = (col_global >= r0) & (col_global < r1)
        ghosts = np.unique(col_global[~own])
        self.ghost_globals = ghosts
        owner = np.searchsorted(starts, ghosts, side="right") - 1
        self.recv_counts = np.bincount(owner, minlength=nranks).astype(np.int64)
        col_local = np.empty(col_global.size, dtype=np.int64)
        col_local[own] = col_global[own] - r0
        col_local[~own] = self.m_local + np.searchsorted(ghosts, col_global[~own])
        self.col_local = col_local.astype(np.int32)
        # tell every owner which of its rows we need
        requests = [ghosts[owner == r] for r in range(nranks)]
        gathered = all_gather_object(requests)  # gathered[src][dst] = rows src needs from dst
        send_lists = [np.asarray(gathered[src][rank], dtype=np.int64) - r0 for src in range(nranks)]
        self.send_counts = np.array([len(x) for x in send_lists], dtype=np.int64)
        self.send_rows = np.concatenate(send_lists) if send_lists else np.zeros(0, dtype=np.int64)
        self.n_ghost = int(ghosts.size)
        self.n_send = int(self.send_rows.size)


class _DevArray:
    """Minimal __cuda_array_interface__ holder so torch can wrap a raw device pointer without copying."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def wrap_buffer(ptr, n, on_device):
    import torch

    if n == 0:
        return torch.zeros(0, dtype=torch.float64, device="cuda" if on_device else "cpu")
    if on_device:
        return torch.as_tensor(_DevArray(ptr, n), device="cuda")
    arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(n,))
    return torch.from_numpy(arr)


def _on_stream(stream, on_device):
    """torch.distributed orders its device work against torch's CURRENT stream: make that the stream the library passes (its own,
    when the context was not created on a torch stream), so that the collective runs after the kernels that produced the buffer and
    before the library's next use of it."""
    import contextlib

    if not on_device or not stream:
        return contextlib.nullcontext()
    import torch

    if torch.cuda.current_stream().cuda_stream == stream:
        return contextlib.nullcontext()
    return torch.cuda.stream(torch.cuda.ExternalStream(stream))


def make_allreduce(on_device=True, group=None, host_staged=False):
    """pyfunc(dev_ptr, n, stream) for Context.set_allreduce / the oracle's hook.  host_staged: device buffers travel through host
    tensors (rehearsals of the multi-process path on a backend without device collectives, i.e. gloo)."""
    import torch.distributed as dist

    wrapped = {}  # (ptr, n) -> tensor view: the library re-uses one small device buffer, wrapping it costs more than the collective

    def allreduce(ptr, n, stream):
        with _on_stream(stream, on_device):
            return _allreduce(ptr, n)

    def _allreduce(ptr, n):
        t = wrapped.get((ptr, n))
        if t is None:
            if len(wrapped) > 64:
                wrapped.clear()
            t = wrapped[(ptr, n)] = wrap_buffer(ptr, n, on_device)
        if on_device and host_staged:
            h = t.cpu()  # synchronises with the current stream
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            t.copy_(h)
            return 0
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return 0

    return allreduce


def make_halo(plan, on_device=True, group=None, host_staged=False):
    """pyfunc(send_ptr, recv_ptr, ncols, stream): exchange the packed rows with point-to-point messages."""
    import torch.distributed as dist

    def halo(send_ptr, recv_ptr, ncols, stream):
        with _on_stream(stream, on_device):
            return _halo(send_ptr, recv_ptr, ncols)

    def _halo(send_ptr, recv_ptr, ncols):
        send = wrap_buffer(send_ptr, plan.n_send * ncols, on_device)
        recv = wrap_buffer(recv_ptr, plan.n_ghost * ncols, on_device)
        dev_recv = None
        if on_device and host_staged:
            import torch

            send, dev_recv = send.cpu(), recv
            recv = torch.empty(plan.n_ghost * ncols, dtype=torch.float64)
        ops = []
        so = ro = 0
        for r in range(plan.nranks):
            ns, nr = int(plan.send_counts[r]) * ncols, int(plan.recv_counts[r]) * ncols
            if r != plan.rank:
                if nr:
                    ops.append(dist.P2POp(dist.irecv, recv[ro:ro + nr], r, group))
                if ns:
                    ops.append(dist.P2POp(dist.isend, send[so:so + ns], r, group))
            so += ns
            ro += nr
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        if dev_recv is not None and plan.n_ghost:
            dev_recv.copy_(recv)
        return 0

    return halo


def all_gather_object_fn(group=None):
    import torch.distributed as dist

    def fn(obj):
        out = [None] * dist.get_world_size(group)
        dist.all_gather_object(out, obj, group=group)
        return out

    return fn
