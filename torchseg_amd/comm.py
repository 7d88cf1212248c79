"""RCCL / xGMI collectives of the hot path through the C-ABI (`tsg_comm_*`, include/tsg_hip.h), enqueued on the
CURRENT HIP stream.

The reference's exchange steps (legacy/sync_bn/syncbn.py:75-78, comm.py:57-132; apex's torch.distributed calls) run
here as one library call between the producing and the consuming kernel: no ProcessGroup stream handshake, no Work
object, capturable by a hipGraph.  `torch.distributed` is only the bootstrap (it carries the 128-byte RCCL unique id
and the IPC handles of the mailboxes between the ranks) and the fallback for CPU tensors (gloo tests).

  TSG_COMM=0           keep every collective on torch.distributed (A/B comparisons)
  TSG_XGMI_ONESHOT=1   SyncBN statistics through the one-shot peer-mailbox all-reduce (tsg_xgmi_small_allreduce with
                       mailboxes attached) instead of ncclAllReduce
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _lib as L

_MAX_SMALL = 2 * 2048 + 8          # floats: the largest SyncBN message (2C+2 at C = 2048)


class Comm(object):
    """One communicator per process group (one rank per GPU)."""

    def __init__(self, group=None, device=None, rccl=True, xgmi=None):
        self.lib = L.lib()
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = torch.cuda.current_device() if device is None else int(device)
        ident = None
        if rccl:
            n = self.lib.tsg_comm_unique_id_bytes()
            buf = C.create_string_buffer(n)
            if self.rank == 0:
                L.check(self.lib.tsg_comm_get_unique_id(buf), "tsg_comm_get_unique_id")
            box = [buf.raw if self.rank == 0 else None]
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(box, src=src, group=group)
            ident = C.create_string_buffer(box[0], n)
        handle = C.c_void_p()
        L.check(self.lib.tsg_comm_create(ident, self.rank, self.world, self.device, C.byref(handle)), "tsg_comm_create")
        self.handle = handle
        self.has_rccl = bool(rccl)
        self.one_shot = False
        if xgmi is None:
            xgmi = os.environ.get("TSG_XGMI_ONESHOT", "0") == "1"
        if xgmi:
            self._attach_mailboxes()

    def _attach_mailboxes(self):
        hb = self.lib.tsg_comm_xgmi_handle_bytes()
        mine = C.create_string_buffer(hb)
        L.check(self.lib.tsg_comm_xgmi_export(self.handle, _MAX_SMALL, mine), "tsg_comm_xgmi_export")
        handles = [None] * self.world
        dist.all_gather_object(handles, mine.raw, group=self.group)
        allh = C.create_string_buffer(b"".join(handles), hb * self.world)
        L.check(self.lib.tsg_comm_xgmi_attach(self.handle, allh), "tsg_comm_xgmi_attach")
        dist.barrier(group=self.group)          # every mailbox is mapped before the first store into it
        self.one_shot = True

    @staticmethod
    def _check(t):
        if not t.is_cuda or not t.is_contiguous():
            raise L.TsgError("tsg_comm collectives take contiguous tensors on an AMD GPU")

    def all_reduce(self, t):
        """t <- sum over ranks, in place, on the current stream."""
        self._check(t)
        L.check(self.lib.tsg_comm_allreduce(self.handle, t.data_ptr(), t.numel(), L.dtype_code(t), L.stream_ptr(t)),
                "tsg_comm_allreduce")
        return t

    def small_all_reduce(self, t):
        """fp32 message of a few KB (SyncBN statistics): the one-shot mailbox kernel when attached, else RCCL."""
        self._check(t)
        if t.dtype != torch.float32:
            raise L.TsgError("small_all_reduce takes float32 messages")
        L.check(self.lib.tsg_xgmi_small_allreduce(self.handle, t.data_ptr(), t.numel(), L.stream_ptr(t)),
                "tsg_xgmi_small_allreduce")
        return t

    def all_gather(self, send, recv):
        self._check(send)
        self._check(recv)
        if recv.numel() != send.numel() * self.world or recv.dtype != send.dtype:
            raise L.TsgError("all_gather: recv must hold world x send")
        L.check(self.lib.tsg_comm_allgather(self.handle, send.data_ptr(), recv.data_ptr(), send.numel(),
                                            L.dtype_code(send), L.stream_ptr(send)), "tsg_comm_allgather")
        return recv

    def reduce_scatter(self, send, recv):
        """recv <- this rank's slice of the sum over ranks of send (recv may be send's own slice: in place)."""
        self._check(send)
        self._check(recv)
        if send.numel() != recv.numel() * self.world or recv.dtype != send.dtype:
            raise L.TsgError("reduce_scatter: send must hold world x recv")
        L.check(self.lib.tsg_comm_reduce_scatter(self.handle, send.data_ptr(), recv.data_ptr(), recv.numel(),
                                                 L.dtype_code(send), L.stream_ptr(send)), "tsg_comm_reduce_scatter")
        return recv

    def broadcast(self, t, root=0):
        self._check(t)
        L.check(self.lib.tsg_comm_broadcast(self.handle, t.data_ptr(), t.numel(), L.dtype_code(t), int(root),
                                            L.stream_ptr(t)), "tsg_comm_broadcast")
        return t

    def destroy(self):
        if self.handle is not None:
            h, self.handle = self.handle, None
            self.lib.tsg_comm_destroy(h)


_comms = {}


def get(group=None, like=None):
    """The communicator of `group` for HIP tensors, created on first use; None when the collective has to stay on
    torch.distributed (process group not initialised, CPU tensors / gloo backend, TSG_COMM=0)."""
    if like is not None and not like.is_cuda:
        return None
    if not (dist.is_available() and dist.is_initialized()) or os.environ.get("TSG_COMM", "1") == "0":
        return None
    key = id(group) if group is not None else None
    c = _comms.get(key)
    if c is None:
        if dist.get_backend(group) != "nccl":
            return None
        c = _comms[key] = _create_agreed(group)
    return c or None


def _create_agreed(group):
    """Comm(group), or False on EVERY rank when the communicator could not be built on any of them (librccl.so not found
    or lacking a symbol, ncclGetUniqueId / ncclCommInitRank refusing): the ranks agree through the process group that
    carried the unique id, so that they all take torch.distributed's collectives instead of splitting over two paths."""
    err = None
    try:
        c = Comm(group)
    except Exception as e:                                 # noqa: BLE001 - whatever it was, the other ranks must hear of it
        c, err = None, e
    ok = torch.tensor([1 if c is not None else 0], dtype=torch.int32, device="cuda" if torch.cuda.is_available() else "cpu")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if int(ok.item()) == 1:
        return c
    if c is not None:
        c.destroy()
    import warnings
    warnings.warn("torchseg_amd.comm: no tsg_comm communicator on this process group (%s); SyncBN exchanges and gradient "
                  "buckets stay on torch.distributed" % (err if err is not None else "another rank failed to build its"))
    return False


def get_extra(group=None, tag="extra", like=None):
    """A SECOND communicator on `group` (ddp.py's TSG_DDP_COMM=separate: gradient buckets that should not serialise
    with the SyncBN exchanges).  Every rank must ask for it at the same point of its program."""
    if get(group, like) is None:
        return None
    key = (id(group) if group is not None else None, tag)
    c = _comms.get(key)
    if c is None:
        c = _comms[key] = Comm(group, xgmi=False)
    return c


def shutdown():
    """Destroy every communicator.  ncclCommDestroy has to run BEFORE dist.destroy_process_group() and before the HIP
    runtime is torn down: Engine.__exit__ and bench.py call this explicitly; `_atexit_shutdown` is the guarded fallback
    for every other entry point that wrapped a model (eval scripts, user code)."""
    for c in list(_comms.values()):
        if c:
            c.destroy()
    _comms.clear()


def _atexit_shutdown():
    """Fallback for entry points that never call shutdown(): runs at interpreter exit (before the HIP runtime's static
    destructors) and only while the process group that carried the unique id still exists — if the caller already
    destroyed it, the ordering the docstring above warns about can no longer be kept, and the communicators are left to
    process teardown exactly as before."""
    if not _comms:
        return
    try:
        if dist.is_available() and dist.is_initialized():
            shutdown()
    except Exception:                                      # never turn a clean exit into a failing one
        pass


import atexit as _atexit  # noqa: E402
_atexit.register(_atexit_shutdown)
