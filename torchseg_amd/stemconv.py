"""Stem convolution on our MFMA kernels.

The two 7x7 / stride-2 / 3->64 convolutions of BiSeNet (ResNet conv1,
furnace/base_model/resnet.py:96-97; SpatialPath.conv_7x7, bisenet network.py:116)
read the raw image, which needs no gradient: the training step is forward +
weight gradient.  MIOpen runs them through NCHW<->NHWC transposes and a C_in = 3
implicit GEMM (1.62 ms per stem and step at 16 x 3 x 1024^2, tools/probe_stem.py);
`tsg_stem_conv_fwd/_wrw` do the same arithmetic (bf16 operands, fp32
accumulation, what autocast gives the reference) and emit the activation
channels_last, so the SyncBN that follows runs its NHWC kernels.

The DDP wrapper re-classes matching nn.Conv2d modules to `StemConv2d` (same
parameters, same state-dict keys).  Anything the kernel does not cover — fp32
compute, other hyper-parameters, an input that requires grad — stays on the
stock convolution; that is a different GPU implementation of the same op, not a
CPU fallback.  TSG_STEM_CONV=0 disables the swap.
"""
import weakref

import torch
import torch.nn as nn

from . import kernels as K

_cast_cache = [None]          # (weakref to the fp32 image, its version, bf16 copy): both stems read one image


def _as_bf16_image(x):
    if x.dtype == torch.bfloat16 and x.is_contiguous():
        return x
    c = _cast_cache[0]
    if c is not None and c[0]() is x and c[1] == x._version:
        return c[2]
    xb = x.to(torch.bfloat16).contiguous()
    _cast_cache[0] = (weakref.ref(x), x._version, xb)
    return xb


class _StemConvFn(torch.autograd.Function):
    """Returns (y, partial): `partial` holds the per-block sums / square sums of y that the kernel's epilogue collected
    (tsg_stem_conv_fwd_stats), i.e. the statistics pass of the BatchNorm that follows (None when not requested)."""

    @staticmethod
    def forward(ctx, x, weight, with_stats):
        kp = K.provider()
        if with_stats:
            y, partial = kp.stem_conv_fwd_stats(x, weight)
            ctx.mark_non_differentiable(partial)
        else:
            y, partial = kp.stem_conv_fwd(x, weight), None
        ctx.save_for_backward(x)
        ctx.wparam = weight
        ctx.wdtype = weight.dtype
        return y, partial

    @staticmethod
    def backward(ctx, dy, _dpartial):
        (x,) = ctx.saved_tensors
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dy = dy.contiguous(memory_format=torch.channels_last)
        from .convwrw import wrw_on_side_stream           # nothing but the optimizer waits for a stem's weight gradient
        dw = wrw_on_side_stream(lambda: K.provider().stem_conv_wrw(x, dy), ctx.wparam, x, dy)
        return None, dw.to(ctx.wdtype), None


import os as _os
_STEM_STATS = _os.environ.get("TSG_STEM_STATS", "1") != "0"


def attach_bn_partial(y, partial):
    """Leave the statistics the producer of `y` already has for the SyncBatchNorm that consumes it (syncbn.py reads
    them back with take_bn_partial; the tensor's version counter guards against an in-place change in between)."""
    y._tsg_bn_partial = (partial, y._version)


def take_bn_partial(x):
    h = getattr(x, "_tsg_bn_partial", None)
    if h is None or h[1] != x._version or h[0].shape[2] != x.shape[1]:
        return None
    return h[0]


def _wants_bf16(x):
    if x.dtype == torch.bfloat16:
        return True
    return torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16 and x.dtype == torch.float32


class StemConv2d(nn.Conv2d):
    def forward(self, x):
        if (x.is_cuda and self.bias is None and not x.requires_grad and self.weight.dtype == torch.float32
                and self.padding_mode == "zeros" and x.dim() == 4 and _wants_bf16(x)):
            xb = _as_bf16_image(x)
            w = self.weight if self.weight.is_contiguous() else self.weight.contiguous()
            if K.provider().stem_conv_supported(xb, w, self.stride[0], self.padding[0], self.dilation[0], self.groups):
                with_stats = _STEM_STATS and self.training and torch.is_grad_enabled()
                y, partial = _StemConvFn.apply(xb, w, with_stats)
                if partial is not None:
                    attach_bn_partial(y, partial)
                return y
        return super().forward(x)


def _is_stem(m):
    return (type(m) is nn.Conv2d and m.in_channels == 3 and m.out_channels == 64 and m.kernel_size == (7, 7)
            and m.stride == (2, 2) and m.padding == (3, 3) and m.dilation == (1, 1) and m.groups == 1
            and m.bias is None)


def install_stem_conv(module):
    """Re-class the 7x7/2 image stems (in place); returns how many were found."""
    n = 0
    for m in module.modules():
        if _is_stem(m):
            m.__class__ = StemConv2d
            n += 1
    return n
