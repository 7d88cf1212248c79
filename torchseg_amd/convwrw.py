"""Weight gradient of the stride-1 3x3 convolutions (C_in, C_out multiples of 64) on our MFMA kernels.

Round 2: the layer1 kernel below runs over (oc tile, ci tile) pairs of 64 x 64 channels (`tsg_conv3x3_wrw_gen`), which
covers every stride-1 3x3 convolution of BiSeNet-R18 (layer2-4, refines, attention-refinement modules, heads): MIOpen's
split-K kernels for them take 94-288 us each plus a zero fill and a cast (tools/probe_conv2.py), 4.2 ms per step in
total for the weight gradients.  TSG_CONV_WRW_MAXC (default 512) caps the channel count that is re-classed.

Round 1 text: weight gradient of ResNet-18 layer1's 3x3 convolutions on our MFMA kernel.

The four `conv3x3(64, 64)` of layer1 (furnace/base_model/resnet.py:24-29,36-53) see the largest
activations of the context path ([16, 64, 256, 256] at BASELINE config 2).  MIOpen computes their
weight gradient as a split-K implicit GEMM framed by a zero fill and a cast (244 µs each for
268 MB of operands, tools/bench_conv3wrw.py); `tsg_conv3x3_wrw` streams the two operands once
(≈ 95–108 µs through the transposing LDS read, fp32 result instead of a bf16-rounded one).  Forward and the data
gradient stay on MIOpen: the module is re-classed to `WrwConv2d`, whose autograd function calls
`aten::convolution_backward` for dx only.

On by default under the DDP wrapper (bf16, channels_last); TSG_CONV_WRW=0 restores MIOpen's path."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import kernels as K


import os as _os
# TSG_CONV_DGRAD_FWD=0|1|2 (default 2): data gradient through a forward convolution for the C_in == C_out stride-1 layers
# (1) or for every stride-1 layer (2; the forward shapes this creates are in the shipped MIOpen find-db).  Measured at
# the bench shape, same box: 0 -> 807, 1 -> 827 img/s; on the re-tuned db 1 -> 839, 2 -> 845
_DGRAD_MODE = _os.environ.get("TSG_CONV_DGRAD_FWD", "2").strip().lower()
_DGRAD_FWD = _DGRAD_MODE not in ("0", "false", "no", "off", "")
_DGRAD_ANY = _DGRAD_MODE == "2"


# TSG_CONV_C64=1|0 (default 1): forward and data gradient of the 64 -> 64 stride-1 layers on tsg_conv3x3_c64_fwd
# (104 us against 172 us for the library's kernel at [16, 64, 256, 256], tools/bench_conv64.py)
_OWN_C64 = _os.environ.get("TSG_CONV_C64", "1") != "0"
# TSG_CONV_C64_S1=1|0 (default 1): 0 sends the STRIDE-1 64 -> 64 layers (ResNet layer1) to the general kernel
# (csrc/conv3g.hip: 3.8 non-MFMA instructions per MFMA against conv64's 8.9, statistics epilogue) and keeps conv64 for the
# stride-2 spatial-path layers only (A/B of VERDICT r3 item 6)
_OWN_C64_S1 = _os.environ.get("TSG_CONV_C64_S1", "1") != "0"
# TSG_CONV_C64_STATS=1|0 (default 1, round 4): the BatchNorm statistics of a conv64 output in that kernel's epilogue
# (conv64_fwd_k<STATS = true>, built and parity-tested since round 2 but never requested by the autograd nodes): the
# SyncBatchNorm behind each of the six 64 -> 64 layers (layer1 x 4 at 16 x 64 x 256^2, SpatialPath x 2) skips its own
# pass over 134 / 34 MB
_C64_STATS = _os.environ.get("TSG_CONV_C64_STATS", "1") != "0"
# TSG_WEIGHT_SHADOW=1|0 (default 1 since round 5, when the fragment-order filters of the general 3x3 kernel joined the
# refresh launch: ~57 tiny launches per step gone; round 2-4 text follows): bf16 / rotated filters from torchseg_amd.shadow (one refresh launch per step instead
# of ~45 cast / rotate launches).  Measured neutral on one MI355X (1032.5 vs 1035.5 img/s: the 4-us launches it removes sit
# back to back in the queue and cost the GPU almost nothing), so it stays opt-in for hosts that are launch-bound.
_SHADOW = _os.environ.get("TSG_WEIGHT_SHADOW", "1") != "0"


# TSG_WRW_STREAM=1|0: the 3x3 weight gradients on a side HIP stream (round 5).  A layer's weight gradient is needed by
# nobody until the optimizer step, while everything else in the backward pass waits for the DATA gradient; issued on the
# compute stream it sits in the dependent chain all the same.  On a side stream its matrix-core work (25 launches, 2.7 ms
# of the step, MFMA pipe busy 0.41) runs beside the HBM-bound SyncBatchNorm backward passes of the layers in front of it
# (2.3 ms that leave the matrix pipes idle).  The operands are kept alive for the side stream (record_stream), the result
# belongs to the compute stream, and the end of the backward pass (an autograd engine callback) makes the compute stream
# wait for the side stream, so the optimizer, the DDP buckets' gather and anything after .backward() see finished gradients.
#
# Default: on in a process WITHOUT gradient collectives, off once a DDP reducer exists (N > 1, or TSG_FORCE_COLLECTIVES=1);
# TSG_WRW_STREAM=1 forces it on there too.  The HIP runtime multiplexes a process's streams onto 4 hardware queues, least
# used first; with RCCL's and the process group's streams in the process the side stream landed on the compute stream's OWN
# hardware queue (rocprofv3 kernel trace: Queue_Id 1 for both), where two streams run strictly in order and every
# cross-stream event is a bubble: zero overlap and 1 091 img/s on the N > 1 code path of one rank, against 1 175 without
# the side stream and 1 173 with GPU_MAX_HW_QUEUES=8 — i.e. no gain left to defend there
# (profiles/r05_side_stream_hw_queue.txt).
_WRW_ENV = _os.environ.get("TSG_WRW_STREAM")
_WRW_STREAM = _WRW_ENV != "0"
# TSG_WRW_IN_GRAPH=1|0 (default 0): under hipGraph capture the side stream can be kept — its fork (side waits for the
# capturing stream) and its join (the end-of-backward callback) become EDGES of the captured graph.  Measured in round 6
# (profiles/r06_graph_ab.txt, interleaved on one box): a single-stream graph of the step 1 198 / 1 203 img/s (host 0.15-0.2 ms
# per step), the same graph with the 27 weight-gradient forks inside it 1 168 / 1 166 (host 7.2 ms: hipGraphLaunch walks a
# branching graph node by node), eager with the side stream 1 196 / 1 202 (host 12.2-13.2 ms).  A linear graph replays as
# fast as the eager step overlaps, without the host; so the captured step stays on one stream.
_WRW_IN_GRAPH = _os.environ.get("TSG_WRW_IN_GRAPH", "0") == "1"


def side_stream_off_for_collectives():
    """ddp.DistributedDataParallel calls this when it builds a reducer (see above); an explicit TSG_WRW_STREAM wins."""
    global _WRW_STREAM
    if _WRW_ENV is None:
        _WRW_STREAM = False


_wrw_side = {}
_wrw_join_queued = [False]
_wrw_pending = set()                   # id(param) of every parameter whose gradient of THIS backward pass is on the side stream
_OWN_GRAD_HOOKS = set()                # post-accumulate-grad hooks that join the side stream themselves (ddp.Reducer._on_grad)


def allow_grad_hook(fn):
    """Register a post-accumulate-grad hook (by its function object) as one that calls join_wrw_stream() before it reads
    p.grad: the DDP reducer's.  Any OTHER such hook on a parameter (an optimizer-in-backward step) keeps that parameter's
    weight gradient on the compute stream."""
    _OWN_GRAD_HOOKS.add(getattr(fn, "__func__", fn))


def _foreign_grad_hooks(param):
    hooks = getattr(param, "_post_accumulate_grad_hooks", None)
    if not hooks:
        return False
    return any(getattr(h, "__func__", h) not in _OWN_GRAD_HOOKS for h in hooks.values())


def _wrw_join():
    _wrw_join_queued[0] = False
    _wrw_pending.clear()
    for dev, side in _wrw_side.items():
        torch.cuda.current_stream(dev).wait_stream(side)


def join_wrw_stream():
    """Make the current stream wait for weight gradients still running on the side stream (the engine callback does this
    at the end of every backward pass; the DDP reducer calls it before it gathers a bucket early, FusedSGD before it
    reads the gradients, the DDP wrapper at the start of the next forward — which also re-arms the callback should a
    backward pass have died between queueing and running it)."""
    if _wrw_side:
        for dev, side in _wrw_side.items():
            torch.cuda.current_stream(dev).wait_stream(side)


def _kept_as_gradient(out, param):
    """autograd's layout contract for a first gradient (torch/csrc/autograd/functions/accumulate_grad.h): the strides of
    the parameter on every dimension longer than 1."""
    if out.shape != param.shape or out.dtype != param.dtype:
        return False
    return all(a == b for a, b, n in zip(out.stride(), param.stride(), out.shape) if n != 1)


# A list while a caller wants the 3x3 weight gradients LAUNCHED LATER (bench.SegmentedStep: captured into a graph of their own that
# replays on another stream beside the rest of the backward pass): wrw_on_side_stream then hands autograd the (still unwritten)
# result tensor and appends the launch to the list.  The caller runs the list and keeps it alive as long as the operands are read.
_DEFER = None


def wrw_on_side_stream(fn, param, *operands, defer_out=None):
    """`fn()` (launches a weight-gradient kernel, returns its result tensor) on the side stream of the operands' device.
    `param`: the parameter the result is the gradient of.  Only a parameter WITHOUT a gradient takes the side stream:
    autograd's AccumulateGrad then just keeps the tensor; with a gradient already there (accumulation over several backward
    passes, zero_grad(set_to_none=False)) it runs `grad += result` on the compute stream right after this node returns,
    which would read the result under the running kernel.  The same holds for a parameter used TWICE in one backward pass
    (shared weights; `loss = net(x1) + net(x2)`): both uses see `grad is None` at node time, and the engine sums the two
    results on the compute stream — so a parameter that already has a side-stream result in this pass (`_wrw_pending`,
    cleared by the end-of-backward join) makes the compute stream wait for the side stream and computes the second one
    there.  A post-accumulate-grad hook that is not the DDP reducer's would read p.grad before the join."""
    if _DEFER is not None and defer_out is not None:
        Cout, Cin = defer_out
        buf = torch.empty((Cout, Cin, 3, 3), dtype=torch.float32, device=operands[0].device, memory_format=torch.channels_last)
        _DEFER.append((lambda: fn(out=buf), operands, buf))
        # an ALIAS of its own: AccumulateGrad keeps an incoming gradient only when nobody else holds that tensor object, and
        # clones it otherwise — the copy would be taken here, before the deferred launch has written anything
        return buf.detach()
    if not _WRW_STREAM or not operands[0].is_cuda or param is None or not param.is_leaf or param.grad is not None \
            or param.dtype != torch.float32 or param._backward_hooks or _foreign_grad_hooks(param) \
            or (torch.cuda.is_current_stream_capturing() and not _WRW_IN_GRAPH):
        if _wrw_side and operands[0].is_cuda:
            # anything computed here may be summed with (or share scratch state with) a result still on the side stream
            dev = operands[0].device
            side = _wrw_side.get(dev)
            if side is not None and (param is None or id(param) in _wrw_pending):
                torch.cuda.current_stream(dev).wait_stream(side)
        return fn()                                    # (a non-leaf, a cast or a tensor hook would touch the result at once)
    dev = operands[0].device
    cur = torch.cuda.current_stream(dev)
    if id(param) in _wrw_pending:                      # second use of the parameter in this backward pass: see above
        side = _wrw_side.get(dev)
        if side is not None:
            cur.wait_stream(side)
        return fn()
    side = _wrw_side.get(dev)
    if side is None:
        side = _wrw_side[dev] = torch.cuda.Stream(device=dev)
    side.wait_stream(cur)                              # the operands (dy above all) are complete
    with torch.cuda.stream(side):
        out = fn()
    for t in operands:
        if t is not None:
            t.record_stream(side)                      # their memory must not be handed out again under the running kernel
    out.record_stream(cur)
    _wrw_pending.add(id(param))
    if not _kept_as_gradient(out, param):
        # AccumulateGrad keeps a first gradient as it is only when it has the parameter's layout; otherwise it copies it
        # into that layout on the compute stream, at once: the copy must see the finished kernel
        cur.wait_stream(side)
    if not _wrw_join_queued[0]:
        _wrw_join_queued[0] = True
        from torch.autograd import Variable
        try:
            Variable._execution_engine.queue_callback(_wrw_join)
        except RuntimeError:                           # not inside a backward pass (a test calling the function directly)
            _wrw_join()
    return out


def _skip_addend(dskip, like_shape):
    """The gradient of a skip connection as an epilogue addend of the data-gradient kernel (bf16, channels_last,
    the shape of dx), or None when it cannot be one."""
    if dskip is None or tuple(dskip.shape) != tuple(like_shape):
        return None
    if dskip.dtype != torch.bfloat16:
        dskip = dskip.to(torch.bfloat16)
    return dskip.contiguous(memory_format=torch.channels_last)


class _ConvWrwFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, wb, stride, wrt=None, skip=False, with_stats=False):
        # x bf16 channels_last, wb = weight rounded to bf16 (what autocast feeds the convolution)
        ctx.own = _OWN_C64 and K.provider().conv3x3_c64_supported(x, wb, stride, 1, 1, 1) \
            and wb.is_contiguous(memory_format=torch.channels_last)
        ctx.in_hw = (x.shape[2], x.shape[3])
        partial = None
        if ctx.own and with_stats:
            y, partial = K.provider().conv3x3_c64_fwd(x, wb, with_stats=True, stride=stride)
        elif ctx.own:
            y = K.provider().conv3x3_c64_fwd(x, wb, stride=stride)     # 64 -> 64: our kernels (csrc/conv64.hip)
        else:
            y = F.conv2d(x, wb, None, stride, 1)
        ctx.stride = stride
        # stride 2, not 64 -> 64: the data gradient by output parity on tsg_conv3x3_s2_dgrad (csrc/conv3g.hip)
        ctx.s2_gen = (_OWN_S2_DGRAD and not ctx.own and stride == 2 and wb.is_contiguous(memory_format=torch.channels_last)
                      and K.provider().conv3x3_s2_dgrad_supported(wb.shape[1], wb.shape[0]))
        ctx.wrt = wrt                                  # not a graph tensor: a shadow owned by torchseg_amd.shadow
        # the fp32 master (a parameter) for the parity data gradient: its fragment-order image comes from the shadow bank
        ctx.master = weight if (ctx.s2_gen and _SHADOW and weight.dtype == torch.float32) else None
        ctx.save_for_backward(x, wb)
        ctx.wparam = weight                            # (not a graph tensor here: only its .grad is looked at in backward)
        ctx.wdtype = weight.dtype
        ctx.need_dx = x.requires_grad
        ctx.dgrad_fwd = _DGRAD_FWD and stride == 1 and (_DGRAD_ANY or weight.shape[0] == weight.shape[1])
        ctx.set_materialize_grads(False)
        ctx.has_stats = bool(with_stats)
        if with_stats:                                     # second output: the statistics partial of y (or an empty tensor)
            if partial is None:
                partial = x.new_empty(0, dtype=torch.float32)
            ctx.mark_non_differentiable(partial)
        # skip: x is returned as a further output for the block's skip connection, so that the gradient of that path
        # arrives HERE (dskip) and is added in the epilogue of the data-gradient kernel instead of by autograd's own pass.
        # skip == 2 (round 6, stride 2): the further output is x[:, :, ::2, ::2] — what the block's 1x1 / stride-2 shortcut
        # convolution reads — as a compact tensor; the shortcut then runs as a stride-1 convolution of it and its gradient
        # comes back compact (tsg_conv3x3_s2_dgrad_subadd adds it at the even pixels of dx)
        ctx.skip_sub = skip == 2
        xs = x[:, :, ::2, ::2].contiguous(memory_format=torch.channels_last) if ctx.skip_sub else x
        outs = (y,) + ((partial,) if with_stats else ()) + ((xs,) if skip else ())
        return outs if len(outs) > 1 else y

    @staticmethod
    def backward(ctx, dy, *rest):
        x, wb = ctx.saved_tensors
        rest = rest[1:] if ctx.has_stats else rest         # drop the (non-differentiable) partial's slot
        dskip = rest[0] if rest else None
        sub = None
        if dskip is not None and getattr(ctx, "skip_sub", False):
            sub, dskip = dskip, None                       # compact: on the grid of x[:, :, ::2, ::2]
            if sub.dtype != torch.bfloat16:
                sub = sub.to(torch.bfloat16)
            sub = sub.contiguous(memory_format=torch.channels_last)
        if dy is None:                                     # only the skip path was used
            if sub is not None:
                dskip = torch.zeros_like(x)
                dskip[:, :, ::2, ::2] = sub
            return dskip, None, None, None, None, None, None
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = None
        rot = (lambda: ctx.wrt) if ctx.wrt is not None else (lambda: K.provider().conv3x3_weight_rot180_t(wb))
        if ctx.need_dx:
            add = _skip_addend(dskip, x.shape) if ((ctx.own and ctx.stride == 1) or ctx.s2_gen) else None
            if ctx.s2_gen:
                dx = K.provider().conv3x3_s2_dgrad(dy, wb if ctx.master is None else ctx.master, ctx.in_hw, addend=add,
                                                   addend_sub=sub)
                sub = None
                if add is not None:
                    dskip = None
            elif ctx.own and ctx.stride == 2:
                dx = K.provider().conv3x3_c64_s2_dgrad(dy, rot(), ctx.in_hw)
            elif ctx.own:
                dx = K.provider().conv3x3_c64_fwd(dy, rot(), addend=add)
                if add is not None:
                    dskip = None
            elif ctx.dgrad_fwd:
                # dx = conv(dy, rot180(w)^T): the library's forward kernels beat its backward-data kernels on the
                # symmetric layers (tools/probe_conv2.py); same bf16 operands, fp32 accumulation
                dx = F.conv2d(dy, rot(), None, 1, 1)
            else:
                st = ctx.stride
                dx = torch.ops.aten.convolution_backward(dy, x, wb, None, [st, st], [1, 1], [1, 1], False, [0, 0], 1,
                                                         [True, False, False])[0]
            if dskip is not None:
                dx = dx + dskip.to(dx.dtype)
            if sub is not None:                            # a data-gradient path without the compact addend
                dx[:, :, ::2, ::2] += sub.to(dx.dtype)
        dw = wrw_on_side_stream(lambda out=None: K.provider().conv3x3_wrw(x, dy, stride=ctx.stride, out=out), ctx.wparam, x, dy,
                                defer_out=(dy.shape[1], x.shape[1]))
        return dx, dw.to(ctx.wdtype), None, None, None, None, None


# TSG_CONV_S2_DGRAD=1|0 (default 1): data gradient of the stride-2 3x3 layers other than 64 -> 64 (ResNet layer2-4's first
# convolution) on tsg_conv3x3_s2_dgrad, with the shortcut branch's gradient as its epilogue addend (conv_with_skip)
_OWN_S2_DGRAD = _os.environ.get("TSG_CONV_S2_DGRAD", "1") != "0"


# TSG_CONV_GEN=1|0 (default 1): forward and data gradient of every other stride-1 3x3 layer (C_in % 16 == 0, C_out % 64
# == 0) on tsg_conv3x3_gen_fwd (csrc/conv3g.hip) instead of the vendor library; TSG_CONV_GEN_STATS=1|0 (default 1): the
# BatchNorm statistics of the output in its epilogue (every one of these convolutions feeds a SyncBatchNorm)
_OWN_GEN = _os.environ.get("TSG_CONV_GEN", "1") != "0"
_GEN_STATS = _os.environ.get("TSG_CONV_GEN_STATS", "1") != "0"
# TSG_CONV_GEN_BN_ON_LOAD=0|1 (default 0): BatchNorm + ReLU in front of such a layer applied while it loads its input.
# Bit-equal to the separate pass (tests/test_bnconv_gpu.py) and it never writes the normalised activation, but measured
# slower in the step (same box: 1103 img/s with it, 1116 without, gpurun_out/r3d): the 6 BatchNorm passes it removes
# (layer2-4's bn1: ~0.1 ms) cost less than the transform adds to the staging code of 12 convolution / weight-gradient
# launches.  Opt-in, like the round-2 finding for the 64-channel layers (time-neutral there, kept for the memory).
_GEN_BN_ON_LOAD = _os.environ.get("TSG_CONV_GEN_BN_ON_LOAD", "0") == "1"


class _ConvGenFn(torch.autograd.Function):
    """conv3x3 / stride 1 / padding 1 on the general MFMA kernel: forward from the fp32 master weight (the bf16 cast
    happens inside the filter preparation), data gradient as the forward convolution of dy with the rotated / transposed
    filter, weight gradient on tsg_conv3x3_wrw_gen.  Second output: the statistics partial of y (or an empty tensor)."""

    @staticmethod
    def forward(ctx, x, weight, with_stats, skip=False):
        kp = K.provider()
        wf = kp.conv3x3_gen_prep_filter(weight, 0, x)
        out = kp.conv3x3_gen_fwd(x, wf, weight.shape[0], with_stats=with_stats)
        y, partial = out if with_stats else (out, x.new_empty(0, dtype=torch.float32))
        ctx.save_for_backward(x, weight)
        ctx.need_dx = x.requires_grad
        ctx.mark_non_differentiable(partial)
        ctx.set_materialize_grads(False)
        return (y, partial, x) if skip else (y, partial)      # skip: see _ConvWrwFn.forward

    @staticmethod
    def backward(ctx, dy, _dpartial, dskip=None):
        kp = K.provider()
        x, weight = ctx.saved_tensors
        if dy is None:                                     # only the skip path was used
            return dskip, None, None, None
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = None
        if ctx.need_dx:
            O, I = weight.shape[0], weight.shape[1]
            if O % 16 == 0 and I % 64 == 0:
                add = _skip_addend(dskip, x.shape)
                dx = kp.conv3x3_gen_fwd(dy, kp.conv3x3_gen_prep_filter(weight, 1, dy), I, addend=add)
                if add is not None:
                    dskip = None
            else:
                dx = F.conv2d(dy, kp.conv3x3_weight_rot180_t(weight.detach().to(torch.bfloat16)), None, 1, 1)
            if dskip is not None:
                dx = dx + dskip.to(dx.dtype)
        dw = wrw_on_side_stream(lambda out=None: kp.conv3x3_wrw(x, dy, stride=1, out=out), weight, x, dy,
                                defer_out=(dy.shape[1], x.shape[1]))
        return dx, dw.to(weight.dtype), None, None


def _gen_eligible(xb, conv):
    return (_OWN_GEN and conv.stride == (1, 1)
            and not (_OWN_C64 and _OWN_C64_S1 and conv.in_channels == 64 and conv.out_channels == 64)
            and conv.weight.is_contiguous(memory_format=torch.channels_last)
            and K.provider().conv3x3_gen_supported(xb, conv.weight, 1, conv.padding[0], conv.dilation[0], conv.groups))


# TSG_FUSE_SKIP_GRAD=1|0 (default 1): in a residual block whose skip connection is the block input itself (resnet.py:
# 48-52), the gradient of the skip path is added in the epilogue of conv1's data-gradient kernel (conv_with_skip)
_FUSE_SKIP = _os.environ.get("TSG_FUSE_SKIP_GRAD", "1") != "0"


class WrwConv2d(nn.Conv2d):
    def forward(self, x):
        return self._forward(x, False)

    def _forward(self, x, want_skip):
        """want_skip=True: returns (y, x_skip): x_skip is x as a second output of the convolution's autograd node when the
        fused path is taken, else None.  want_skip=False: returns y."""
        def ret(y, x_skip=None):
            return (y, x_skip) if want_skip else y
        if (x.is_cuda and self.bias is None and self.weight.dtype == torch.float32 and x.dim() == 4
                and self.padding_mode == "zeros" and torch.is_grad_enabled() and self.weight.requires_grad
                and (x.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and x.dtype == torch.float32
                                                    and torch.get_autocast_dtype("cuda") == torch.bfloat16))):
            xb = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
            xb = xb.contiguous(memory_format=torch.channels_last)
            if K.provider().conv3x3_wrw_supported(xb, self.weight, self.stride[0], self.padding[0], self.dilation[0],
                                                  self.groups):
                fuse = want_skip and xb is x and x.requires_grad      # the alias must BE the block input
                if _gen_eligible(xb, self):
                    with torch.autocast("cuda", enabled=False):
                        out = _ConvGenFn.apply(xb, self.weight, _GEN_STATS and self.training, fuse)
                    y, partial = out[0], out[1]
                    if partial.numel():
                        from .stemconv import attach_bn_partial
                        attach_bn_partial(y, partial)      # the SyncBatchNorm behind it skips its statistics pass
                    return ret(y, out[2] if fuse else None)
                with torch.autocast("cuda", enabled=False):
                    if _SHADOW and self.weight.is_contiguous(memory_format=torch.channels_last):
                        from .shadow import bank           # bf16 (and rotated) filters kept fresh once per step
                        wb, wrt = bank.get(self.weight, want_rot=True)
                    else:
                        wb, wrt = self.weight.detach().to(torch.bfloat16), None
                    own64 = _OWN_C64 and _OWN_C64_S1 and self.stride == (1, 1) and self.in_channels == 64 \
                        and self.out_channels == 64
                    c64 = _OWN_C64 and self.in_channels == 64 and self.out_channels == 64
                    if not own64:                          # stride 2 on the parity kernel: its epilogue takes the shortcut's gradient
                        own64 = (_OWN_S2_DGRAD and self.stride == (2, 2) and not c64
                                 and self.weight.is_contiguous(memory_format=torch.channels_last)
                                 and K.provider().conv3x3_s2_dgrad_supported(self.in_channels, self.out_channels))
                    stats = _C64_STATS and c64 and self.training
                    sk = bool(fuse and own64)
                    if sk and want_skip == 2 and self.stride == (2, 2) and not c64:
                        sk = 2                             # compact sub-sampled alias (see _ConvWrwFn.forward)
                    out = _ConvWrwFn.apply(xb, self.weight, wb, self.stride[0], wrt, sk, stats)
                    if not stats:
                        return out if (fuse and own64) else ret(out)
                    y, partial = out[0], out[1]
                    if partial.numel():
                        from .stemconv import attach_bn_partial
                        attach_bn_partial(y, partial)      # the SyncBatchNorm behind it skips its statistics pass
                    return ret(y, out[2] if (fuse and own64) else None)
        return ret(super().forward(x))


# TSG_SKIP_SUBSAMPLE=1|0 (default 1, round 6): a stride-2 block hands its shortcut convolution x[:, :, ::2, ::2] as a compact
# tensor (conv_with_skip(subsample=True)) instead of x
_SKIP_SUB = _os.environ.get("TSG_SKIP_SUBSAMPLE", "1") != "0"


def conv_with_skip(conv, x, subsample=False):
    """`conv(x)` for the first convolution of a residual block whose skip connection is x itself (stride 1) or starts from x
    (stride 2: the 1x1 shortcut convolution, resnet.py:139-146).  Returns (y, x_skip):
    x_skip is x routed through the convolution's autograd node when our kernels compute its data gradient (then the
    block must use x_skip for `out += residual`: the skip path's gradient is added in that kernel's epilogue), else None
    (use x).  subsample=True (stride 2 only): x_skip may come back as the COMPACT x[:, :, ::2, ::2] — recognisable by its
    size — on which the 1x1 / stride-2 shortcut convolution is a stride-1 convolution; its gradient then returns compact
    too and is added at the even pixels of dx (no zero-filled full-size gradient)."""
    if _FUSE_SKIP and isinstance(conv, WrwConv2d) and conv.stride in ((1, 1), (2, 2)) and isinstance(x, torch.Tensor) \
            and x.is_cuda and torch.is_grad_enabled():
        return conv._forward(x, 2 if (subsample and _SKIP_SUB and conv.stride == (2, 2)) else True)
    return conv(x), None


# TSG_BN_BSUM=1|0 (default 1, round 6): the backward sums of a BatchNorm -> ReLU in the epilogue of the data gradient of the
# convolution behind it (tsg_conv3x3_c64_*dgrad_bnsums; the general kernels: _gen_dgrad_with_bn_sums below) instead of a
# pass of their own over the gradient and the BatchNorm's input (tsg_bn_bwd_reduce): one read of the gradient less, and the
# read of x under a matrix kernel that leaves most of the HBM rate unused.  Same values up to the order of the fp32 sums.
_BN_BSUM = _os.environ.get("TSG_BN_BSUM", "0") == "1"


def _c64_dgrad_and_bn_sums(kp, dy, rot, x, fp, stride, layout, N, C, HW):
    """(da, partial, S): the data gradient of a 64 -> 64 convolution whose input was relu(bn(x)) and the backward sums of
    that BatchNorm — from the kernel's epilogue where it has one, else by the separate pass."""
    if _BN_BSUM and hasattr(kp, "conv3x3_c64_bnsums_supported") and kp.conv3x3_c64_bnsums_supported(x.shape[0], x.shape[2], x.shape[3], stride):
        if stride == 2:
            da, partial = kp.conv3x3_c64_s2_dgrad(dy, rot, (x.shape[2], x.shape[3]), bsum=(x, fp))
        else:
            da, partial = kp.conv3x3_c64_fwd(dy, rot, bsum=(x, fp))
        return da, partial, partial.shape[0]
    da = kp.conv3x3_c64_s2_dgrad(dy, rot, (x.shape[2], x.shape[3])) if stride == 2 else kp.conv3x3_c64_fwd(dy, rot)
    partial, Sn = kp.bn_bwd_reduce(da, x, None, layout, N, C, HW, fp, True)
    return da, partial, Sn


class _BnReluConvFn(torch.autograd.Function):
    """conv(relu(bn(x))) for the 3x3 layers our kernels cover, with the normalised activation never stored: the
    convolution (tsg_conv3x3_c64_*_fwd for 64 -> 64, tsg_conv3x3_gen_fwd for every other stride-1 layer) and its weight
    gradient (tsg_conv3x3_wrw_*_norm) apply a x + b, ReLU while they stage x; the data gradient of the convolution feeds
    the ordinary SyncBN backward, which needs dy and x only.  Second output: the statistics partial of y (general
    kernel) or an empty tensor."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn, use_batch_stats, group, hint, weight, wb, stride, wrt, gen=False, out_stats=False):
        from . import syncbn as S
        kp = K.provider()
        layout, N, C, HW = K.bn_layout(x)
        world = S._world(group) if use_batch_stats else 1
        count_dev = None
        g32 = gamma.float() if gamma is not None else None
        b32 = beta.float() if beta is not None else None
        if use_batch_stats:
            invstd, fp, count_dev = S._batch_statistics(kp, x, layout, N, C, HW, bn, g32, b32, group, world, hint)
        else:
            mean = bn.running_mean.float()
            invstd = torch.rsqrt(bn.running_var.float() + bn.eps)
            fp = kp.bn_affine(mean, invstd, g32, b32)
        partial = None
        if gen:
            out = kp.conv3x3_gen_fwd(x, kp.conv3x3_gen_prep_filter(weight, 0, x), weight.shape[0], with_stats=out_stats,
                                     in_ab=fp)
            y, partial = out if out_stats else (out, None)
        elif out_stats:
            y, partial = kp.conv3x3_c64_fwd(x, wb, with_stats=True, stride=stride, in_ab=fp)
        else:
            y = kp.conv3x3_c64_fwd(x, wb, stride=stride, in_ab=fp)
        if partial is None:
            partial = x.new_empty(0, dtype=torch.float32)
        ctx.save_for_backward(x, weight if gen else wb, gamma, beta, invstd, fp, count_dev)
        ctx.wparam = weight
        ctx.cfg = (layout, N, C, HW, use_batch_stats, group, world, stride, weight.dtype, gen)
        ctx.wrt = wrt
        ctx.mark_non_differentiable(partial)
        return y, partial

    @staticmethod
    def backward(ctx, dy, _dpartial):
        from . import syncbn as S
        kp = K.provider()
        x, wb, gamma, beta, invstd, fp, count_dev = ctx.saved_tensors
        layout, N, C, HW, use_batch_stats, group, world, stride, wdtype, gen = ctx.cfg
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dy = dy.contiguous(memory_format=torch.channels_last)
        dw = wrw_on_side_stream(lambda out=None: kp.conv3x3_wrw(x, dy, stride=stride, in_ab=fp, out=out), ctx.wparam, x, dy, fp,
                                defer_out=(dy.shape[1], x.shape[1]))
        if gen:                                                  # wb is the fp32 master weight here
            da = kp.conv3x3_gen_fwd(dy, kp.conv3x3_gen_prep_filter(wb, 1, dy), wb.shape[1])
            partial, Sn = kp.bn_bwd_reduce(da, x, None, layout, N, C, HW, fp, True)
        else:
            rot = ctx.wrt if ctx.wrt is not None else kp.conv3x3_weight_rot180_t(wb)
            da, partial, Sn = _c64_dgrad_and_bn_sums(kp, dy, rot, x, fp, stride, layout, N, C, HW)
        dgamma, dbeta, bp = S._backward_pack(kp, partial, Sn, C, N * HW, invstd, fp, count_dev, use_batch_stats, group,
                                             world, x.device)
        dx, _ = kp.bn_bwd_apply(da, x, None, layout, N, C, HW, bp, True, False)
        if gamma is None:
            dgamma = dbeta = None
        else:
            dgamma = dgamma.to(gamma.dtype)
            dbeta = dbeta.to(beta.dtype) if beta is not None else None
        return dx, dgamma, dbeta, None, None, None, None, dw.to(wdtype), None, None, None, None, None


class _StemBnReluConvFn(torch.autograd.Function):
    """conv_3x3(relu(bn(stem(image)))) — SpatialPath's first two ConvBnRelu (bisenet network.py:116-117) — as ONE node:
    forward as StemConv2d + _BnReluConvFn; backward additionally folds the BatchNorm backward APPLY into the staging of the
    stem's weight gradient (tsg_stem_conv_wrw_bn), its only consumer, so neither relu(bn(xc)) nor d(xc) is ever stored."""

    @staticmethod
    def forward(ctx, img, w_stem, gamma, beta, bn, use_batch_stats, group, weight, wb, stride, wrt, out_stats=False):
        from . import syncbn as S
        kp = K.provider()
        if use_batch_stats:
            xc, hint = kp.stem_conv_fwd_stats(img, w_stem)
        else:
            xc, hint = kp.stem_conv_fwd(img, w_stem), None
        layout, N, C, HW = K.bn_layout(xc)
        world = S._world(group) if use_batch_stats else 1
        count_dev = None
        g32 = gamma.float() if gamma is not None else None
        b32 = beta.float() if beta is not None else None
        if use_batch_stats:
            invstd, fp, count_dev = S._batch_statistics(kp, xc, layout, N, C, HW, bn, g32, b32, group, world, hint)
        else:
            invstd = torch.rsqrt(bn.running_var.float() + bn.eps)
            fp = kp.bn_affine(bn.running_mean.float(), invstd, g32, b32)
        if out_stats:
            y, partial = kp.conv3x3_c64_fwd(xc, wb, with_stats=True, stride=stride, in_ab=fp)
        else:
            y, partial = kp.conv3x3_c64_fwd(xc, wb, stride=stride, in_ab=fp), img.new_empty(0, dtype=torch.float32)
        ctx.save_for_backward(img, xc, wb, gamma, beta, invstd, fp, count_dev)
        ctx.wparam, ctx.wstem = weight, w_stem
        ctx.cfg = (layout, N, C, HW, use_batch_stats, group, world, stride, weight.dtype, w_stem.dtype)
        ctx.wrt = wrt
        ctx.mark_non_differentiable(partial)
        return y, partial

    @staticmethod
    def backward(ctx, dy, _dpartial=None):
        from . import syncbn as S
        kp = K.provider()
        img, xc, wb, gamma, beta, invstd, fp, count_dev = ctx.saved_tensors
        layout, N, C, HW, use_batch_stats, group, world, stride, wdtype, sdtype = ctx.cfg
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dy = dy.contiguous(memory_format=torch.channels_last)
        dw = wrw_on_side_stream(lambda out=None: kp.conv3x3_wrw(xc, dy, stride=stride, in_ab=fp, out=out), ctx.wparam, xc, dy, fp,
                                defer_out=(dy.shape[1], xc.shape[1]))
        rot = ctx.wrt if ctx.wrt is not None else kp.conv3x3_weight_rot180_t(wb)
        da, partial, Sn = _c64_dgrad_and_bn_sums(kp, dy, rot, xc, fp, stride, layout, N, C, HW)
        dgamma, dbeta, bp = S._backward_pack(kp, partial, Sn, C, N * HW, invstd, fp, count_dev, use_batch_stats, group,
                                             world, xc.device)
        dw_stem = wrw_on_side_stream(lambda: kp.stem_conv_wrw_bn(img, da, xc, bp), ctx.wstem, img, da, xc, bp)   # the BN backward apply
        #                                                                                     happens in its staging
        if gamma is None:
            dgamma = dbeta = None
        else:
            dgamma = dgamma.to(gamma.dtype)
            dbeta = dbeta.to(beta.dtype) if beta is not None else None
        return None, dw_stem.to(sdtype), dgamma, dbeta, None, None, None, dw.to(wdtype), None, None, None, None


# TSG_STEM_BN_WRW=1|0 (default 1): stem -> BN -> ReLU -> 64 -> 64 3x3 as one autograd node (see _StemBnReluConvFn)
_STEM_BN_WRW = _os.environ.get("TSG_STEM_BN_WRW", "1") != "0"


def stem_bn_relu_conv(stem, bn, relu, img, conv):
    """`conv(relu(bn(stem(img))))`: the fused node when everything is on the HIP path, None otherwise."""
    from .syncbn import SyncBatchNorm
    from .stemconv import StemConv2d, _as_bf16_image, _wants_bf16
    if not (_STEM_BN_WRW and _BN_ON_LOAD and _OWN_C64 and relu is not None and isinstance(stem, StemConv2d)
            and isinstance(bn, SyncBatchNorm) and isinstance(conv, WrwConv2d) and isinstance(img, torch.Tensor)
            and img.is_cuda and img.dim() == 4 and not img.requires_grad and stem.bias is None and conv.bias is None
            and stem.weight.dtype == torch.float32 and conv.weight.dtype == torch.float32 and stem.weight.requires_grad
            and conv.weight.requires_grad and torch.is_grad_enabled() and stem.padding_mode == "zeros"
            and _wants_bf16(img) and bn.momentum is not None and bn.num_features == 64
            and conv.in_channels == 64 and conv.out_channels == 64 and conv.kernel_size == (3, 3)
            and conv.stride in ((1, 1), (2, 2)) and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1
            and conv.weight.is_contiguous(memory_format=torch.channels_last)):
        return None
    xb = _as_bf16_image(img)
    w_stem = stem.weight if stem.weight.is_contiguous() else stem.weight.contiguous()
    if not K.provider().stem_conv_supported(xb, w_stem, stem.stride[0], stem.padding[0], stem.dilation[0], stem.groups):
        return None
    use_batch_stats = bn.training or not bn.track_running_stats
    with torch.autocast("cuda", enabled=False):
        if _SHADOW:
            from .shadow import bank
            wb, wrt = bank.get(conv.weight, want_rot=True)
        else:
            wb, wrt = conv.weight.detach().to(torch.bfloat16), None
        y, partial = _StemBnReluConvFn.apply(xb, w_stem, bn.weight, bn.bias, bn, use_batch_stats, bn.process_group,
                                             conv.weight, wb, conv.stride[0], wrt, bool(_C64_STATS and conv.training))
        if partial.numel():
            from .stemconv import attach_bn_partial
            attach_bn_partial(y, partial)
        return y


# TSG_BN_ON_LOAD=1|0 (default 1): BatchNorm + ReLU in front of a 64 -> 64 3x3 convolution applied while that convolution
# (and its weight gradient) load their input, instead of as a pass of its own
_BN_ON_LOAD = _os.environ.get("TSG_BN_ON_LOAD", "1") != "0"


def bn_relu_conv(bn, relu, x, conv):
    """`conv(relu(bn(x)))` — seg_oprs.py:39-46 followed by the next ConvBnRelu's convolution (bisenet network.py:117-118),
    BasicBlock's bn1 -> relu -> conv2 (resnet.py:36-46).  One fused autograd node on HIP tensors when `bn` is our
    SyncBatchNorm and `conv` one of the 3x3 layers our convolution kernels cover (64 -> 64 stride 1 / 2: conv64; any other
    stride-1 layer with C_in <= 512: the general kernel); the three modules otherwise."""
    from .syncbn import SyncBatchNorm
    from .furnace_glue import norm_act
    if (_BN_ON_LOAD and relu is not None and isinstance(bn, SyncBatchNorm) and isinstance(conv, WrwConv2d)
            and isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16
            and x.shape[1] == conv.in_channels and conv.bias is None
            and conv.weight.dtype == torch.float32 and conv.weight.requires_grad and torch.is_grad_enabled()
            and bn.momentum is not None and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)
            and conv.weight.is_contiguous(memory_format=torch.channels_last)):
        c64 = (_OWN_C64 and conv.in_channels == 64 and conv.out_channels == 64
               and (_OWN_C64_S1 or conv.stride[0] == 2)
               and K.provider().conv3x3_c64_supported(x, conv.weight, conv.stride[0], conv.padding[0], conv.dilation[0],
                                                      conv.groups))
        gen = (not c64) and _GEN_BN_ON_LOAD and conv.in_channels <= 512 and _gen_eligible(x, conv) \
            and K.provider().conv3x3_wrw_supported(x, conv.weight, 1, conv.padding[0], conv.dilation[0], conv.groups)
        if c64 or gen:
            bn._check_input_dim(x)
            use_batch_stats = bn.training or not bn.track_running_stats
            hint = None
            if use_batch_stats and hasattr(x, "_tsg_bn_partial"):
                from .stemconv import take_bn_partial
                hint = take_bn_partial(x)
            with torch.autocast("cuda", enabled=False):
                if gen:
                    wb, wrt = None, None
                elif _SHADOW:
                    from .shadow import bank
                    wb, wrt = bank.get(conv.weight, want_rot=True)
                else:
                    wb, wrt = conv.weight.detach().to(torch.bfloat16), None
                out_stats = ((gen and _GEN_STATS) or (c64 and _C64_STATS)) and conv.training
                y, partial = _BnReluConvFn.apply(x, bn.weight, bn.bias, bn, use_batch_stats, bn.process_group, hint,
                                                 conv.weight, wb, conv.stride[0], wrt, gen, bool(out_stats))
            if partial.numel():
                from .stemconv import attach_bn_partial
                attach_bn_partial(y, partial)
            return y
    return conv(norm_act(bn, relu, x))


def _eligible(m):
    return (type(m) is nn.Conv2d and m.in_channels % 64 == 0 and m.out_channels % 64 == 0 and m.kernel_size == (3, 3)
            and m.stride in ((1, 1), (2, 2)) and m.padding == (1, 1) and m.dilation == (1, 1) and m.groups == 1 and m.bias is None)


def install_conv_wrw(module):
    """Re-class the eligible convolutions in place; returns how many were found."""
    import os
    maxc = int(os.environ.get("TSG_CONV_WRW_MAXC", "512"))
    n = 0
    for m in module.modules():
        if _eligible(m) and max(m.in_channels, m.out_channels) <= maxc:
            m.__class__ = WrwConv2d
            n += 1
    return n
