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
            g = (dout.reshape(N, C).float() * (1.0 / HW)).to(x.dtype)
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
