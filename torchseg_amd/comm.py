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


class CommUnavailable(RuntimeError):
    """Raised by Comm(...) on EVERY rank of the group when any of them could not build its communicator."""


def _agree(ok, group):
    """MIN all-reduce of a local outcome over `group` (on the device the group's backend moves)."""
    dev = "cuda" if (dist.get_backend(group) == "nccl" and torch.cuda.is_available()) else "cpu"
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return int(t.item()) == 1


class Comm(object):
    """One communicator per process group (one rank per GPU)."""

    def __init__(self, group=None, device=None, rccl=True, xgmi=None):
        """Every rank of `group` runs the SAME sequence of torch.distributed collectives here whatever fails locally
        (ADVICE r5): local steps are tried, their outcome is agreed on with a MIN all-reduce, and only then does anybody
        raise — CommUnavailable, on every rank together.  Round 5 raised where the error happened: rank 0 failing in
        tsg_comm_get_unique_id skipped the id broadcast the other ranks were already waiting in.
          1 local   load the library; rank 0: ncclGetUniqueId
          2 coll    broadcast of the id (rank 0 sends None when step 1 failed there)
          3 coll    agree
          4 local   tsg_comm_create (ncclCommInitRank: RCCL's own rendezvous, entered by all ranks or by none)
          5 coll    agree
          6 xgmi    export (local) -> all-gather of the handles (None = failed) -> attach (local) -> agree"""
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.handle = None
        self.has_rccl = bool(rccl)
        self.one_shot = False
        err, ident = None, None
        try:                                                   # 1
            self.lib = L.lib()
            self.device = (torch.cuda.current_device() if torch.cuda.is_available() else 0) if device is None else int(device)
            if rccl:
                n = self.lib.tsg_comm_unique_id_bytes()
                buf = C.create_string_buffer(n)
                if self.rank == 0:
                    L.check(self.lib.tsg_comm_get_unique_id(buf), "tsg_comm_get_unique_id")
                    ident = buf.raw
        except Exception as e:                                 # noqa: BLE001 - the other ranks must hear of it, whatever it was
            err = e
        if rccl:                                               # 2
            box = [ident if self.rank == 0 else None]
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(box, src=src, group=group)
            ident = box[0]
            if ident is None and err is None:
                err = RuntimeError("rank 0 could not produce the RCCL unique id")
        self._agree_or_raise(err, "load the library / exchange the RCCL unique id")        # 3
        try:                                                   # 4
            handle = C.c_void_p()
            idbuf = C.create_string_buffer(ident, len(ident)) if rccl else None
            L.check(self.lib.tsg_comm_create(idbuf, self.rank, self.world, self.device, C.byref(handle)), "tsg_comm_create")
            self.handle = handle
        except Exception as e:                                 # noqa: BLE001
            err = e
        try:
            self._agree_or_raise(err, "create the communicator")                             # 5
        except CommUnavailable:
            self.destroy()
            raise
        if xgmi is None:
            xgmi = os.environ.get("TSG_XGMI_ONESHOT", "0") == "1"
        if xgmi:
            self._attach_mailboxes()

    def _agree_or_raise(self, err, what):
        if not _agree(err is None, self.group):
            raise CommUnavailable("could not %s on %s" % (what, "this rank: %s" % (err,) if err is not None
                                                          else "another rank")) from err

    def _attach_mailboxes(self):
        """The one-shot mailboxes (step 6 above); a rank that cannot export or map them leaves EVERY rank on RCCL."""
        hb, mine = 0, None
        try:
            hb = self.lib.tsg_comm_xgmi_handle_bytes()
            buf = C.create_string_buffer(hb)
            L.check(self.lib.tsg_comm_xgmi_export(self.handle, _MAX_SMALL, buf), "tsg_comm_xgmi_export")
            mine = buf.raw
        except Exception as e:                                 # noqa: BLE001
            import warnings
            warnings.warn("torchseg_amd.comm: mailbox export failed (%s); SyncBN messages stay on RCCL" % (e,))
        handles = [None] * self.world
        dist.all_gather_object(handles, mine, group=self.group)
        ok = all(h is not None for h in handles)
        if ok:
            try:
                allh = C.create_string_buffer(b"".join(handles), hb * self.world)
                L.check(self.lib.tsg_comm_xgmi_attach(self.handle, allh), "tsg_comm_xgmi_attach")
            except Exception as e:                             # noqa: BLE001
                import warnings
                warnings.warn("torchseg_amd.comm: mailbox attach failed (%s); SyncBN messages stay on RCCL" % (e,))
                ok = False
        # every mailbox is mapped before the first store into it (the all-reduce is the barrier round 5 had here), and
        # all ranks use the mailboxes or none does
        self.one_shot = _agree(ok, self.group)

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
    or lacking a symbol, ncclGetUniqueId / ncclCommInitRank refusing): Comm.__init__ keeps the ranks' collective sequences
    identical and raises CommUnavailable on all of them together, so that they all take torch.distributed's collectives
    instead of splitting over two paths (or waiting in a collective a failing rank skipped)."""
    try:
        return Comm(group)
    except CommUnavailable as e:
        import warnings
        warnings.warn("torchseg_amd.comm: no tsg_comm communicator on this process group (%s); SyncBN exchanges and gradient "
                      "buckets stay on torch.distributed" % (e,))
        return False


def get_extra(group=None, tag="extra", like=None):
    """A SECOND communicator on `group` (ddp.py's TSG_DDP_COMM=separate: gradient buckets that should not serialise
    with the SyncBN exchanges).  Every rank must ask for it at the same point of its program."""
    if get(group, like) is None:
        return None
    key = (id(group) if group is not None else None, tag)
    c = _comms.get(key)
    if c is None:
        try:
            c = _comms[key] = Comm(group, xgmi=False)
        except CommUnavailable:                                # every rank lands here together: share the first communicator
            c = _comms[key] = get(group, like)
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
