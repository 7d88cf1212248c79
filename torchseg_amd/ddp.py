"""DistributedDataParallel for one-process-per-GPU training over RCCL/xGMI.

Replaces `apex.parallel.DistributedDataParallel(model)` as the reference wraps
its model (model/bisenet/cityscapes.bisenet.R18/train.py:98-99): parameter
broadcast from rank 0 at construction, then per backward pass a bucketed
gradient all-reduce (SUM, / world_size) launched WHILE autograd is still
running, fenced at the end of backward so `optimizer.step()` sees averaged
gradients.  State-dict keys gain the 'module.' prefix exactly like apex's
wrapper (engine.py:97-101 strips it).

MI355X design (instead of apex's flatten -> all_reduce -> unflatten copies):
  * gradients LIVE in per-bucket flat fp32 buffers (`param.grad` is a view), so a
    bucket is all-reduced in place with no flatten/unflatten traffic;
  * buckets follow the order in which gradients actually arrived in the first
    backward (reverse-autograd order), which also makes statically unused
    parameters (DFN has 5) a non-event: they are simply never bucketed;
  * each bucket's all-reduce is issued from the autograd hook of its last
    gradient.  On HIP tensors it is ONE `tsg_comm_allreduce` (RCCL) on a side
    stream this reducer owns, fenced by two events (gather -> side stream,
    side stream -> end of backward): no ProcessGroup Work object and no
    ProcessGroup stream handshake (round 3; CPU tensors / gloo keep
    `dist.all_reduce(async_op=True)`).  By default it is the SAME communicator
    the SyncBN exchanges use on the compute stream: RCCL serialises launches of
    one communicator in issue order whatever their streams, and every rank issues
    the same program order (autograd hooks of one model), so there is no
    cross-communicator ordering to get wrong; the reference's 1-float loss
    all-reduce (furnace/utils/pyt_utils.py all_reduce_tensor) and late "straggler"
    gradients go through that communicator too, so it is the ONLY one in flight
    during a step.  The price is that a SyncBN
    exchange issued while a bucket is on the wire waits for it (<= 0.25 ms per
    40 MB bucket at 8 GPUs).  TSG_DDP_COMM=separate gives the buckets their own
    communicator (full overlap; not validated on a multi-GPU node yet);
    TSG_DDP_COMM=torch restores torch.distributed;
  * TSG_DDP_RS=1 splits a bucket's all-reduce into reduce-scatter + all-gather
    (direct exchange over all 7 xGMI links instead of a
    ring), in place on the flat buffer (padded to a multiple of the world size);
  * bucket size defaults to 1e7 elements (40 MB): xGMI rings are per-link bound
    (~153 GB/s), so few large messages beat many small ones.

This wrapper's forward is also the one choke point we own around an unchanged
network.py (which computes its loss inside forward, network.py:103-109): it
enters bf16 autocast (BASELINE config 2; the reference has no AMP), keeps the
model in channels_last where MIOpen is faster on MI355X, and installs the aten
upsample overrides.  Controlled by env so train.py needs no edit:
  TSG_DTYPE=bf16|fp32   (default bf16 on GPU)    TSG_CHANNELS_LAST=1|0 (default 1 on GPU)
  TSG_FUSE_PSA=1|0      (default: on when the model has a PointwiseSpatialAttention block)
  TSG_FUSE_LOSS=1|0     (default 1 on GPU: nn.CrossEntropyLoss / F.cross_entropy heads, also behind F.log_softmax, run
                        on tsg_ohem_* in plain-CE mode; fusion.py)
  TSG_FUSE_ADD_UP=1|0   (default 1 on GPU: `fm += last_fm` followed by F.interpolate is one kernel; fusion.py)
  TSG_FUSE_HEAD=1|0     (default 1 on GPU: a bilinear F.interpolate of <= 32-channel logits by >= 4 stays pending and is
                        evaluated inside the criterion's kernels, tsg_ohem_up_*: 3x faster than writing and re-reading
                        the full-resolution logits; any other consumer materialises it.  fusion.py)
  TSG_FUSE_CHAIN=1|0    (default 1 on GPU: a ConvBnRelu called right after another one (bisenet network.py:131-137) applies
                        the first one's BatchNorm + ReLU while its own convolution loads its input; fusion.PendingCbr)
  TSG_SPLIT_BIAS=1|0    (default 1 on GPU: conv bias add / bias grad through our column-sum kernel)
  TSG_STEM_CONV=1|0     (default 1 on GPU: 7x7/2 image stems on tsg_stem_conv_* instead of MIOpen)
  TSG_ADAPTIVE_POOL=1|0 (default 1 on GPU: nn.AdaptiveAvgPool2d on channels_last maps -> tsg_adaptive_avgpool_nhwc_*)
  TSG_CONV_WRW=1|0      (default 1 on GPU: weight gradient of the 64->64 3x3/1 convolutions on tsg_conv3x3_wrw;
                        TSG_CONV_WRW_IMPL=gen|tr|v1 picks the kernel of the 64 -> 64 layers, default gen)
  TSG_CLS_HEAD=1|0      (default 1 on GPU: the 1x1 classifier convolution of a head (<= 32 classes) on tsg_cls_head_*:
                        planar logits for the criterion kernels, no layout copies, no separate bias passes; clshead.py)
  TSG_CONV_S2_DGRAD=1|0 (default 1 on GPU: data gradient of the stride-2 3x3 layers other than 64 -> 64 on tsg_conv3x3_s2_dgrad, the
                        shortcut branch's gradient as its epilogue addend; convwrw.py)
  TSG_CAT=1|0           (default 1 on GPU: FeatureFusion's torch.cat([x1, x2], 1) on tsg_cat2_rows; pool.py)
  TSG_VEC_CONV=1|0      (default 1 on GPU: bias-free 1x1 convolutions applied to pooled [B, C, 1, 1] maps (channel attention,
                        global context) on tsg_conv1x1_vec_*: one launch forward, one backward, fp32 master weight; vecconv.py)
  TSG_PW_CONV=1|0       (default 1 on GPU: the bias-free 1x1 convolutions of whole maps (ResNet shortcuts, SpatialPath.conv_1x1,
                        FeatureFusion.conv_1x1) compute their weight gradient as a chunked batched GEMM folded in a fixed order —
                        the vendor library's split-K atomics were the last run-to-run difference of the step; pwconv.py)
  TSG_FORK_MODULES=a,b  (default none; e.g. "spatial_path": direct sub-modules of an unchanged network.py that run on a side HIP
                        stream and are joined where their output is first used — BiSeNet's detail branch beside its context
                        path.  Box-dependent (+1.4 % / -0.2 %), hence opt-in; our own BiSeNet builder: TSG_FORK_SPATIAL=1; fusion.py)
  TSG_WRW_STREAM=1|0    (default: 1 on GPU without a gradient reducer, 0 with one — world > 1: the 3x3 weight gradients run on a
                        side HIP stream beside the SyncBatchNorm backward passes of the layers in front of them and are
                        joined at the end of the backward pass.  With RCCL's streams in the process the side stream shares the
                        compute stream's hardware queue and costs 7 % instead of gaining 1.7 %; convwrw.py)
  TSG_FP32_EXACT=1|0    (default 1: with TSG_DTYPE=fp32 every convolution runs on tsg_conv2d_f32_exact_* — exact products,
                        fp64 accumulation — instead of the vendor library's fp32 kernels: the parity mode, exactconv.py)
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn
from torch.autograd import Variable


def _env_flag(name, default):
    v = os.environ.get(name)
    if v is None:
        return default
    return v.strip().lower() not in ("0", "false", "no", "off", "")


def apply_channels_last(module):
    """channels_last for every 4-D conv weight except stems reading <= 4 input
    channels: MIOpen's NHWC kernels lose badly at C_in = 3 on gfx950 (measured
    15.6 ms vs 3.5 ms for the 7x7/2 stem at 16x3x1024^2, tools/probe_conv.py)."""
    for m in module.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            cin = m.weight.shape[1] * m.groups
            fmt = torch.contiguous_format if cin <= 4 else torch.channels_last
            m.weight.data = m.weight.data.contiguous(memory_format=fmt)
    return module


class _Bucket(object):
    __slots__ = ("params", "offsets", "flat", "pending", "work", "ready", "views", "maps")

    def __init__(self, params, device, pad_to=1):
        self.params = params
        self.offsets = []
        n = 0
        for p in params:
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4        # keep every gradient view 16-byte aligned (vector kernels)
        q = 4 * max(1, int(pad_to))              # reduce-scatter: equal, 16-byte aligned slices per rank
        n = (n + q - 1) // q * q
        self.flat = torch.zeros(n, dtype=torch.float32, device=device)
        self.pending = len(params)
        self.work = None
        self.ready = [False] * len(params)
        self.views = None           # filled lazily: one strided view per parameter
        self.maps = {}              # static block maps of the multi-tensor gather, keyed by the copied indices

    def view(self, i):
        """The slot of parameter i, shaped and strided like the parameter (a channels_last weight gets a
        channels_last gradient view: same element order, so gather and optimizer address both as flat arrays)."""
        p = self.params[i]
        flat = self.flat[self.offsets[i]:self.offsets[i] + p.numel()]
        if p.dim() == 4 and not p.is_contiguous() and p.is_contiguous(memory_format=torch.channels_last):
            return flat.as_strided(p.shape, p.stride())
        return flat.view_as(p)


class Reducer(object):
    """Bucketed, overlapped gradient averaging over a process group."""

    def __init__(self, params, process_group=None, message_size=10000000, delay_allreduce=False,
                 gradient_average=True, gradient_predivide_factor=1.0):
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.message_size = int(message_size)
        self.delay = bool(delay_allreduce)
        self.average = gradient_average
        self.prediv = float(gradient_predivide_factor)
        self.params = [p for p in params if p.requires_grad]
        self.buckets = None          # built after the first backward
        self._slot = {}              # param -> (bucket index, index in bucket)
        self._order = []             # arrival order during the first backward
        self._seen = set()
        self._callback_queued = False
        self._stragglers = []        # params outside the plan that got a grad later
        mode = os.environ.get("TSG_DDP_COMM", "shared").strip().lower()
        self._comm_mode = mode if mode in ("shared", "separate", "torch") else "shared"
        self._rs = _env_flag("TSG_DDP_RS", False)
        self._comm = None            # tsg_comm of the buckets (HIP tensors), resolved at the first launch
        self._side = None            # the HIP stream the bucket collectives run on
        from .convwrw import allow_grad_hook
        allow_grad_hook(self._on_grad)  # joins the weight-gradient side stream (_launch / _finish_backward) before it reads a .grad
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    # -- autograd side -------------------------------------------------------
    def _queue_callback(self):
        if not self._callback_queued:
            Variable._execution_engine.queue_callback(self._finish_backward)
            self._callback_queued = True

    def _on_grad(self, p):
        self._queue_callback()
        if self.buckets is None:
            if p not in self._seen:
                self._seen.add(p)
                self._order.append(p)
            return
        slot = self._slot.get(p)
        if slot is None:
            self._stragglers.append(p)
            return
        b, i = slot
        bucket = self.buckets[b]
        if not bucket.ready[i]:
            bucket.ready[i] = True
            bucket.pending -= 1
            if bucket.pending == 0 and not self.delay:
                self._launch(bucket)

    @staticmethod
    def _gather(bucket, scale=1.0):
        """Move the gradients autograd produced this pass into the flat buffer, multiplied by `scale` (the 1/world of
        the average, folded into the copy so that no separate pass over the 54-361 MB of gradients is needed after
        the all-reduce), one multi-tensor launch per 128 tensors on the GPU, and make `param.grad` the bucket views."""
        if bucket.views is None:
            bucket.views = [bucket.view(i) for i in range(len(bucket.params))]
        todo = []
        for i, p in enumerate(bucket.params):
            v = bucket.views[i]
            if not bucket.ready[i]:
                v.zero_()                                # planned parameter without a gradient this pass
            elif p.grad.data_ptr() != v.data_ptr():
                todo.append(i)
            elif scale != 1.0:
                v.mul_(scale)                            # accumulated in place (zero_grad(set_to_none=False))
        if not todo:
            return
        on_gpu = bucket.flat.is_cuda
        if on_gpu:
            from .optim import _same_dense_order
        fast = []
        for i in todo:
            p, v = bucket.params[i], bucket.views[i]
            g = p.grad
            if on_gpu and g.dtype == torch.float32 and _same_dense_order(g, v):
                fast.append(i)
            else:
                v.copy_(g)
                if scale != 1.0:
                    v.mul_(scale)
                p.grad = v
        if fast:
            from . import kernels as K
            kp = K.provider()
            for c0 in range(0, len(fast), kp.SGD_MAX_SEGS):
                idx = tuple(fast[c0:c0 + kp.SGD_MAX_SEGS])
                srcs = [bucket.params[i].grad for i in idx]
                bucket.maps[idx] = kp.multi_copy(srcs, [bucket.views[i] for i in idx], bucket.maps.get(idx), scale)
            for i in fast:
                bucket.params[i].grad = bucket.views[i]

    def _bucket_comm(self, flat):
        """The tsg_comm the buckets use (None: torch.distributed)."""
        if not flat.is_cuda or self._comm_mode == "torch":
            return None
        if self._comm is None:
            from . import comm
            if self._comm_mode == "separate":
                self._comm = comm.get_extra(self.group, "ddp-buckets", like=flat)
            else:
                self._comm = comm.get(self.group, like=flat)
            if self._comm is None:
                self._comm_mode = "torch"                # gloo group, TSG_COMM=0, ...
                return None
            self._side = torch.cuda.Stream(device=flat.device)
        return self._comm

    def _reduce(self, flat, comm):
        """SUM over ranks of one flat bucket, in place; returns what _finish_backward has to wait for."""
        if comm is not None:
            cur = torch.cuda.current_stream(flat.device)
            ready = torch.cuda.Event()
            ready.record(cur)                            # the gather copy (and everything before it) is enqueued
            self._side.wait_event(ready)
            with torch.cuda.stream(self._side):
                if self._rs:
                    n = flat.numel() // self.world
                    mine = flat[comm.rank * n:(comm.rank + 1) * n]
                    comm.reduce_scatter(flat, mine)      # in place: my slice of the sum ...
                    comm.all_gather(mine, flat)          # ... then everybody's slices
                else:
                    comm.all_reduce(flat)
                done = torch.cuda.Event()
                done.record(self._side)
            return done
        if self._rs and self.world > 1 and not flat.is_cuda:
            # the same split on torch.distributed primitives (gloo has no reduce_scatter: one reduce per slice)
            n = flat.numel() // self.world
            rank = dist.get_rank(self.group)
            for r in range(self.world):
                dst = dist.get_global_rank(self.group, r) if self.group is not None else r
                dist.reduce(flat[r * n:(r + 1) * n], dst=dst, op=dist.ReduceOp.SUM, group=self.group)
            parts = [torch.empty(n, dtype=flat.dtype) for _ in range(self.world)]
            dist.all_gather(parts, flat[rank * n:(rank + 1) * n].clone(), group=self.group)
            for r in range(self.world):
                flat[r * n:(r + 1) * n].copy_(parts[r])
            return None
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    @staticmethod
    def _wait(work, device):
        if work is None:
            return
        if isinstance(work, torch.cuda.Event):
            torch.cuda.current_stream(device).wait_event(work)      # stream-level fence, the host does not block
        else:
            work.wait()                                             # torch.distributed Work: fence on HIP, blocking on gloo

    def _launch(self, bucket):
        from .convwrw import join_wrw_stream
        join_wrw_stream()                                # weight gradients issued on the side stream (convwrw.py) are complete
        # averaging = pre-division by the world size inside the gather copy; apex's gradient_predivide_factor only
        # chooses where the same division happens (overflow control for fp16 buckets), irrelevant for fp32 buckets
        self._gather(bucket, 1.0 / self.world if self.average else 1.0)
        bucket.work = self._reduce(bucket.flat, self._bucket_comm(bucket.flat))
        if bucket.work is None:
            bucket.work = True                           # synchronous path: nothing to wait for

    def _build_plan(self):
        device = self._order[0].device if self._order else torch.device("cpu")
        groups, cur, n = [], [], 0
        for p in self._order:
            cur.append(p)
            n += p.numel()
            if n >= self.message_size:
                groups.append(cur)
                cur, n = [], 0
        if cur:
            groups.append(cur)
        self.buckets = []
        for bi, ps in enumerate(groups):
            bucket = _Bucket(ps, device, self.world if self._rs else 1)
            for i, p in enumerate(ps):
                self._slot[p] = (bi, i)
                view = bucket.view(i)
                view.copy_(p.grad)
                p.grad = view
                bucket.ready[i] = True
            bucket.pending = 0
            self.buckets.append(bucket)

    def _finish_backward(self):
        self._callback_queued = False
        from .convwrw import join_wrw_stream
        join_wrw_stream()                                # this callback may run before convwrw's own: the plan's first copies
        #                                                  and the stragglers read gradients written on the side stream
        if self.buckets is None:
            if not self._order:
                return
            self._build_plan()
        for bucket in self.buckets:
            if bucket.work is None:
                # incomplete (a planned param got no grad this pass; _gather zeroes the hole) or delayed
                self._launch(bucket)
        for bucket in self.buckets:
            if bucket.work is not True:
                self._wait(bucket.work, bucket.flat.device)
            bucket.work = None
            bucket.pending = len(bucket.params)
            bucket.ready = [False] * len(bucket.params)
        # parameters outside the plan that received a gradient after all: AFTER the buckets are fenced, and through the
        # buckets' own communicator when there is one, so that no second communicator (the ProcessGroup's) is ever in
        # flight beside it
        for p in self._stragglers:
            g = p.grad
            comm = self._comm if (self._comm is not None and g.is_cuda and g.is_contiguous()
                                  and g.dtype in (torch.float32, torch.bfloat16)) else None
            if comm is not None:
                comm.all_reduce(g.view(-1))
            else:
                dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
            if self.average:
                g.div_(self.world)
        self._stragglers = []


def _broadcast_coalesced(tensors, src, group):
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault((t.dtype, t.device), []).append(t)
    for ts in by_dtype.values():
        flat = torch.cat([t.detach().reshape(-1) for t in ts])
        dist.broadcast(flat, src, group=group)
        off = 0
        with torch.no_grad():
            for t in ts:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()


class DistributedDataParallel(nn.Module):
    """See module docstring.  Positional/keyword arguments follow apex's wrapper;
    the ones that only tune apex's internal copies are accepted and ignored."""

    def __init__(self, module, message_size=10000000, delay_allreduce=False, shared_param=None,
                 allreduce_trigger_params=None, retain_allreduce_buffers=False,
                 allreduce_always_fp32=False, num_allreduce_streams=1, allreduce_communicators=None,
                 gradient_average=True, gradient_predivide_factor=1.0,
                 gradient_average_split_factor=None, prof=False,
                 process_group=None, compute_dtype=None, channels_last=None):
        super(DistributedDataParallel, self).__init__()
        self.module = module
        self.process_group = process_group
        first = next(module.parameters(), None)
        self.on_gpu = first is not None and first.is_cuda
        if compute_dtype is None:
            name = os.environ.get("TSG_DTYPE", "bf16" if self.on_gpu else "fp32").lower()
            compute_dtype = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16,
                             "fp32": torch.float32, "float32": torch.float32}[name]
        self.compute_dtype = compute_dtype
        if channels_last is None:
            channels_last = _env_flag("TSG_CHANNELS_LAST", self.on_gpu)
        self.channels_last = bool(channels_last)
        if self.channels_last:
            apply_channels_last(self.module)
            from . import syncbn
            syncbn.PREFER_CHANNELS_LAST_OUTPUT = True
        # An unchanged reference network.py reaches the fused operators through fusion.FuseMode; our own workload
        # builders call them directly and say so (`tsg_native_fusions`), which spares them the mode's Python dispatch.
        native = bool(getattr(module, "tsg_native_fusions", False))
        self.fuse_psa = False
        self.fuse_loss = self.on_gpu and _env_flag("TSG_FUSE_LOSS", not native)
        self.fuse_add_up = self.on_gpu and _env_flag("TSG_FUSE_ADD_UP", not native)
        self.fuse_head = self.on_gpu and _env_flag("TSG_FUSE_HEAD", not native)
        self.fuse_chain = self.on_gpu and _env_flag("TSG_FUSE_CHAIN", not native)
        if self.on_gpu:
            from .upsample import install_aten_overrides
            install_aten_overrides()
            from .psa import model_has_psa
            self.fuse_psa = _env_flag("TSG_FUSE_PSA", model_has_psa(module) and not native)
            if _env_flag("TSG_SPLIT_BIAS", True):
                from .convbias import split_conv_bias
                split_conv_bias(self.module)
            if _env_flag("TSG_STEM_CONV", True):
                from .stemconv import install_stem_conv
                install_stem_conv(self.module)
            if _env_flag("TSG_CONV_WRW", True):
                from .convwrw import install_conv_wrw
                install_conv_wrw(self.module)
            if _env_flag("TSG_ADAPTIVE_POOL", True):
                from .pool import install_adaptive_pool
                install_adaptive_pool(self.module)
            if _env_flag("TSG_CLS_HEAD", True):
                from .clshead import install_cls_head
                install_cls_head(self.module)                        # 1x1 classifier convolutions -> planar logits
            if _env_flag("TSG_VEC_CONV", True):
                from .vecconv import install_pooled_conv
                install_pooled_conv(self.module)                     # 1x1 convolutions of pooled [B, C, 1, 1] maps
            if _env_flag("TSG_PW_CONV", True):
                from .pwconv import install_pointwise_conv
                install_pointwise_conv(self.module)                  # remaining full-map 1x1 layers: reproducible weight gradient
            if self.compute_dtype == torch.float32:
                # fp32 = the parity mode: convolutions on the reference-accuracy kernels (exactconv.py: logits 1.3-1.9e-5
                # from the float64 truth, the reference's CPU path 7-8e-5, the vendor library's fp32 kernels 5-6e-5)
                from . import exactconv
                exactconv.install(self.module)

        self._fork_hooks = []
        if self.on_gpu and not native:
            # TSG_FORK_MODULES (default none; e.g. "spatial_path"): direct sub-modules that run on a side HIP stream (fusion.py)
            from . import fusion
            for name in os.environ.get("TSG_FORK_MODULES", "").split(","):
                sub = getattr(self.module, name.strip(), None) if name.strip() else None
                if isinstance(sub, nn.Module):
                    self._fork_hooks.append(sub.register_forward_pre_hook(fusion.fork_pre_hook))
                    self._fork_hooks.append(sub.register_forward_hook(fusion.fork_post_hook))
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.reducer = None
        force = dist.is_initialized() and _env_flag("TSG_FORCE_COLLECTIVES", False)   # 1-rank validation of the N>1 path
        if self.world_size > 1 or force:
            tensors = [p for p in module.parameters()] + [b for b in module.buffers()]
            _broadcast_coalesced(tensors, 0, process_group)
            self.reducer = Reducer(module.parameters(), process_group, message_size, delay_allreduce,
                                   gradient_average, gradient_predivide_factor)
            if self.on_gpu:
                from .convwrw import side_stream_off_for_collectives
                side_stream_off_for_collectives()        # its hardware queue collides with the compute stream's (convwrw.py)

    def forward(self, *inputs, **kwargs):
        import contextlib
        if self.on_gpu:
            from . import convwrw
            if convwrw._wrw_side and not torch.cuda.is_current_stream_capturing():
                convwrw.join_wrw_stream()                # normally a no-op wait: the last backward's callback joined already
                convwrw._wrw_join_queued[0] = False      # re-arm (a backward pass that raised may have left it set)
        with contextlib.ExitStack() as stack:
            if self.on_gpu and self.compute_dtype != torch.float32:
                stack.enter_context(torch.autocast("cuda", dtype=self.compute_dtype))
            if self.fuse_psa or self.fuse_loss or self.fuse_add_up or self.fuse_head or self.fuse_chain:
                from .fusion import FuseMode, materialize
                stack.enter_context(FuseMode(psa=self.fuse_psa, loss=self.fuse_loss, add_up=self.fuse_add_up,
                                             head=self.fuse_head, chain=self.fuse_chain))
                return materialize(self.module(*inputs, **kwargs))
            return self.module(*inputs, **kwargs)
