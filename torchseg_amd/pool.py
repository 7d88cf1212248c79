"""Global average pooling on the HIP kernels (csrc/pool.hip).

`GlobalAvgPool(nn.AdaptiveAvgPool2d)` is what our furnace/seg_opr blocks build
where the reference builds `nn.AdaptiveAvgPool2d(1)` (seg_oprs.py:200,224): same
class hierarchy and no parameters, so state dicts are unchanged.  HIP tensors go
to the streaming kernel; CPU tensors (the CPU plumbing config with
nn.BatchNorm2d) take the stock module path.
"""
import torch
import torch.nn as nn

from . import kernels as K


import os as _os
# TSG_GAP_BWD_EXPAND=1|0 (default 1): the gradient of the global average pool as a broadcast view (see _GapFn.backward)
_GAP_BWD_EXPAND = _os.environ.get("TSG_GAP_BWD_EXPAND", "1") != "0"
# TSG_CAT=1|0 (default 1): FeatureFusion's channel concatenation on tsg_cat2_rows (cat_channels)
_CAT = _os.environ.get("TSG_CAT", "1") != "0"


def _scaled_pooled_gradient(dout, N, C, HW, dtype):
    """dout / HW in `dtype`: the value of `(dout.float() * (1 / HW)).to(dtype)` — float product, one rounding — in ONE launch
    when dout already has that dtype (a bf16 tensor times a Python scalar is evaluated in float and rounded once).  Under
    graph replay every launch costs >= 4.8 us whatever it does; this triplet ran five times per step."""
    d = dout.reshape(N, C)
    if d.dtype == dtype:
        return d * (1.0 / HW)
    return (d.float() * (1.0 / HW)).to(dtype)


class _GapFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        kp = K.provider()
        lay = K.bn_layout(x)
        if lay is None:
            x = x.contiguous()
            lay = K.bn_layout(x)
        layout, N, C, HW = lay
        ctx.save_for_backward(x)
        ctx.cfg = (layout, N, C, HW)
        return kp.gap_fwd(x, layout, N, C, HW).view(N, C, 1, 1)

    @staticmethod
    def backward(ctx, dout):
        kp = K.provider()
        (x,) = ctx.saved_tensors
        layout, N, C, HW = ctx.cfg
        if _GAP_BWD_EXPAND:
            # d x[n, c, h, w] = dout[n, c] / HW for every pixel: hand autograd the BROADCAST VIEW instead of a
            # materialised tensor.  The pooled map always has a second consumer here (the squeeze-excite branches of
            # seg_oprs.py:192-238 gate the map they pool), so the engine's gradient accumulation reads the other gradient
            # + C values per sample instead of two full tensors, and gap_bwd's write of the full map disappears
            # (round 4: 4 launches + a third of 9 adds' traffic per BiSeNet step).  A consumer that needs a dense tensor
            # materialises it exactly as before.
            g = _scaled_pooled_gradient(dout, N, C, HW, x.dtype)
            return g.view(N, C, 1, 1).expand(N, C, x.shape[2], x.shape[3])
        dout = dout.reshape(N, C).to(x.dtype).contiguous()
        return kp.gap_bwd(dout, x, layout, N, C, HW)


def global_avg_pool(x):
    """[N,C,H,W] -> [N,C,1,1] mean over H,W."""
    return _GapFn.apply(x)


class GlobalAvgPool(nn.AdaptiveAvgPool2d):
    def __init__(self, output_size=1):
        super().__init__(output_size)

    def forward(self, x):
        one = self.output_size in (1, (1, 1))
        if x.is_cuda and x.dim() == 4 and one and x.dtype in (torch.float32, torch.bfloat16):
            return global_avg_pool(x)
        return super().forward(x)


class _AdaptivePoolFn(torch.autograd.Function):
    """nn.AdaptiveAvgPool2d to a small grid on a channels_last map (PSPNet's pyramid pooling, pspnet network.py:75-109)."""

    @staticmethod
    def forward(ctx, x, OH, OW):
        ctx.in_hw = (x.shape[2], x.shape[3])
        return K.provider().adaptive_avgpool_fwd(x, OH, OW)

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous(memory_format=torch.channels_last)
        return K.provider().adaptive_avgpool_bwd(dout, *ctx.in_hw), None, None


def _out_hw(output_size, x):
    if isinstance(output_size, int):
        return output_size, output_size
    oh, ow = output_size
    return (x.shape[2] if oh is None else int(oh)), (x.shape[3] if ow is None else int(ow))


class AdaptiveAvgPool2d(nn.AdaptiveAvgPool2d):
    """Same module, same (empty) state dict; HIP channels_last inputs take csrc/pool.hip (the framework's NHWC kernel
    needs 2.6 ms for [2, 2048, 90, 90] -> 6 x 6).  The DDP wrapper re-classes nn.AdaptiveAvgPool2d modules to this
    (install_adaptive_pool), so an unchanged network.py gets it too."""

    def forward(self, x):
        if isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16):
            OH, OW = _out_hw(self.output_size, x)
            if OH == 1 and OW == 1:
                return global_avg_pool(x)
            if K.provider().adaptive_avgpool_supported(x, OH, OW) and not x.is_contiguous():
                return _AdaptivePoolFn.apply(x, OH, OW)
        return super().forward(x)


def install_adaptive_pool(module):
    """Re-class every plain nn.AdaptiveAvgPool2d in place; returns how many were found."""
    n = 0
    for m in module.modules():
        if type(m) is nn.AdaptiveAvgPool2d:
            m.__class__ = AdaptiveAvgPool2d
            n += 1
    return n


class _ChanScaleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s, add_identity):
        kp = K.provider()
        lay = K.bn_layout(x)
        if lay is None:
            x = x.contiguous()
            lay = K.bn_layout(x)
        layout, N, C, HW = lay
        s2 = s.reshape(N, C).to(x.dtype).contiguous()
        ctx.save_for_backward(x, s2)
        ctx.cfg = (layout, N, C, HW, bool(add_identity), s.shape, s.dtype)
        return kp.chanscale_fwd(x, s2, layout, N, C, HW, add_identity)

    @staticmethod
    def backward(ctx, dy):
        kp = K.provider()
        x, s2 = ctx.saved_tensors
        layout, N, C, HW, add_identity, s_shape, s_dtype = ctx.cfg
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        if dy.stride() != x.stride():
            t = torch.empty_like(x)
            t.copy_(dy)
            dy = t
        dx, ds = kp.chanscale_bwd(dy, x, s2, layout, N, C, HW, add_identity)
        return dx, ds.to(s_dtype).reshape(s_shape), None


def channel_scale(x, s, add_identity=False):
    """x [N,C,H,W] * s [N,C,1,1] (+ x): the squeeze-excite gate of ARM / FFM."""
    if x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16):
        return _ChanScaleFn.apply(x, s, add_identity)
    return x + x * s if add_identity else x * s


# ---- a gate computed from the pooled map it gates: `x * se(gap(x))` (+ x) ---------------------------------------------------
# TSG_GATE_SPLIT=1|0 (default 1, round 6).  AttentionRefinement / FeatureFusion (seg_oprs.py:192-238) pool a map, run a small
# branch on the pooled vector and scale the map with the result.  The map has two gradients: the gate's, dy s (+ dy), and the
# pooled branch's, g[n, c] / HW on every pixel — and g depends on ds = sum dy x, which the gate's backward produces.  Autograd
# therefore ran tsg_chanscale_bwd (reads dy and x, writes dx1) and then `dx1 += expand(g / HW)`, a pass of its own over the map
# (79 us for FeatureFusion's [16, 256, 128, 128], 22 us for an attention-refinement module).  gated_scale links the two nodes:
# the gate's backward computes ds only and leaves (dy, s) with the pool node, whose backward — which autograd can only run
# after the branch in between — writes dx = dy s (+ dy) + g / HW in one pass: 4 tensor passes instead of 5, same bits.
_GATE_SPLIT = _os.environ.get("TSG_GATE_SPLIT", "1") != "0"


class _GateLink(object):
    __slots__ = ("pending",)

    def __init__(self):
        self.pending = None


class _GapLinkedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, link):
        kp = K.provider()
        layout, N, C, HW = K.bn_layout(x)
        ctx.cfg = (layout, N, C, HW, x.dtype, x.shape[2], x.shape[3])
        ctx.link = link
        return kp.gap_fwd(x, layout, N, C, HW).view(N, C, 1, 1)

    @staticmethod
    def backward(ctx, dout):
        layout, N, C, HW, dtype, H, W = ctx.cfg
        g = _scaled_pooled_gradient(dout, N, C, HW, dtype)                 # what _GapFn.backward hands autograd, per pixel
        pend, ctx.link.pending = ctx.link.pending, None
        if pend is None:                                                   # the gate did not take part in this backward pass
            return g.view(N, C, 1, 1).expand(N, C, H, W), None
        dy, s2, add_identity = pend
        return K.provider().chanscale_bwd_dx(dy, s2, g.contiguous(), layout, N, C, HW, add_identity), None


class _ChanScaleLinkedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s, add_identity, link):
        kp = K.provider()
        layout, N, C, HW = K.bn_layout(x)
        s2 = s.reshape(N, C).to(x.dtype).contiguous()
        ctx.save_for_backward(x, s2)
        ctx.cfg = (layout, N, C, HW, bool(add_identity), s.shape, s.dtype)
        ctx.link = link
        return kp.chanscale_fwd(x, s2, layout, N, C, HW, add_identity)

    @staticmethod
    def backward(ctx, dy):
        kp = K.provider()
        x, s2 = ctx.saved_tensors
        layout, N, C, HW, add_identity, s_shape, s_dtype = ctx.cfg
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        if dy.stride() != x.stride() or dy.data_ptr() % 16:
            t = torch.empty_like(x)
            t.copy_(dy)
            dy = t
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            ds = kp.chanscale_bwd_ds(dy, x, layout, N, C, HW)
            ctx.link.pending = (dy, s2, add_identity)                      # the map's gradient is written by the pool node
            return None, ds.to(s_dtype).reshape(s_shape), None, None
        dx, ds = kp.chanscale_bwd(dy, x, s2, layout, N, C, HW, add_identity)
        return dx, ds.to(s_dtype).reshape(s_shape), None, None


def _reaches(fn, target, limit=64):
    """does the autograd graph below `fn` contain `target`? (the few nodes of a squeeze-excite branch)"""
    seen, stack = set(), [fn]
    while stack and len(seen) < limit:
        f = stack.pop()
        if f is None or id(f) in seen:
            continue
        if f is target:
            return True
        seen.add(id(f))
        stack.extend(nf for nf, _ in f.next_functions)
    return False


def _run_pooled_branch(mods, s):
    """the modules of a squeeze-excite branch behind its pool, one after the other — except that a ConvBnRelu directly followed
    by nn.Sigmoid runs with the sigmoid inside its fused pooled-layer launch (vecconv.pooled_layer) when that applies"""
    from .vecconv import pooled_layer
    i = 0
    while i < len(mods):
        m = mods[i]
        if (i + 1 < len(mods) and type(mods[i + 1]) is nn.Sigmoid and not mods[i + 1]._forward_hooks
                and not mods[i + 1]._forward_pre_hooks and hasattr(m, "conv") and hasattr(m, "has_bn")
                and isinstance(s, torch.Tensor)):
            y = pooled_layer(m, s, sigmoid_after=True)
            if y is not None:
                s = y
                i += 2
                continue
        s = m(s)
        i += 1
    return s


def gated_scale(x, branch, add_identity=False):
    """`x * branch(x)` (+ x) where `branch` is an nn.Sequential that starts with a global average pool (seg_oprs.py:199-205,
    222-231): with the linked autograd nodes described above when x is a channels_last HIP map that needs a gradient, the
    plain `channel_scale(x, branch(x), add_identity)` otherwise."""
    mods = list(branch.children()) if isinstance(branch, nn.Sequential) else []
    ok = (_GATE_SPLIT and mods and isinstance(mods[0], GlobalAvgPool) and mods[0].output_size in (1, (1, 1))
          and isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16)
          and x.requires_grad and torch.is_grad_enabled() and not branch._forward_hooks and not branch._forward_pre_hooks
          and not mods[0]._forward_hooks and not mods[0]._forward_pre_hooks)
    if ok:
        lay = K.bn_layout(x)
        kp = K.provider()
        ok = lay is not None and hasattr(kp, "chanscale_bwd_ds") and kp.chanscale_split_supported(x, lay[0], lay[2])
    if not ok:
        return channel_scale(x, branch(x), add_identity)
    link = _GateLink()
    pooled = _GapLinkedFn.apply(x, link)
    s = _run_pooled_branch(mods[1:], pooled)
    if not (isinstance(s, torch.Tensor) and s.requires_grad and s.grad_fn is not None and pooled.grad_fn is not None
            and s.numel() == x.shape[0] * x.shape[1] and _reaches(s.grad_fn, pooled.grad_fn)):
        return channel_scale(x, s, add_identity)       # (the pool node then hands autograd its broadcast view, as before)
    return _ChanScaleLinkedFn.apply(x, s, add_identity, link)


class _Cat2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.ca = a.shape[1]
        return K.provider().cat_channels(a, b)

    @staticmethod
    def backward(ctx, g):
        return g[:, :ctx.ca], g[:, ctx.ca:]                # views, as the framework's cat backward


def cat_channels(a, b):
    """torch.cat([a, b], dim=1) (seg_oprs.py:233-235) for two channels_last HIP maps of one dtype whose rows are multiples
    of 16 bytes: one streaming kernel; anything else takes torch.cat."""
    if (_CAT and a.is_cuda and b.is_cuda and a.dim() == 4 and b.dim() == 4 and a.dtype == b.dtype
            and a.shape[0] == b.shape[0] and a.shape[2:] == b.shape[2:]
            and (a.shape[1] * a.element_size()) % 16 == 0 and (b.shape[1] * b.element_size()) % 16 == 0
            and a.shape[2] * a.shape[3] > 1
            and a.is_contiguous(memory_format=torch.channels_last) and b.is_contiguous(memory_format=torch.channels_last)):
        return _Cat2Fn.apply(a, b)
    return torch.cat([a, b], dim=1)


class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, s, p):
        kp = K.provider()
        y, idx = kp.maxpool_fwd(x, k, s, p)
        ctx.save_for_backward(idx)
        ctx.cfg = (tuple(x.shape), k, s, p)
        return y

    @staticmethod
    def backward(ctx, dy):
        kp = K.provider()
        (idx,) = ctx.saved_tensors
        in_shape, k, s, p = ctx.cfg
        dy = dy.contiguous(memory_format=torch.channels_last)
        return kp.maxpool_bwd(dy, idx, in_shape, k, s, p), None, None, None


def _as_int(v):
    if isinstance(v, (tuple, list)):
        return v[0] if len(set(v)) == 1 else None
    return v


class MaxPool2d(nn.MaxPool2d):
    """nn.MaxPool2d whose channels_last HIP inputs run on csrc/pool.hip (one-byte argmax,
    deterministic gather backward); anything else takes the stock module path."""

    def forward(self, x):
        k, s, p = _as_int(self.kernel_size), _as_int(self.stride), _as_int(self.padding)
        vec = 8 if x.dtype == torch.bfloat16 else 4
        if (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16)
                and k is not None and s is not None and p is not None and _as_int(self.dilation) == 1
                and not self.ceil_mode and not self.return_indices and 2 * p <= k and k <= 15
                and x.shape[1] % vec == 0 and not x.is_contiguous()
                and x.is_contiguous(memory_format=torch.channels_last)):
            return _MaxPoolFn.apply(x, k, s, p)
        return super().forward(x)
