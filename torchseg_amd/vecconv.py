"""1x1 convolutions of globally pooled maps on their own kernels (csrc/vecconv.hip).

The channel-attention branches of BiSeNet / DFN and BiSeNet's global context (furnace/seg_opr/seg_oprs.py:199-205, :222-231;
bisenet network.py:34-39) run `ConvBnRelu(C_in, C_out, 1, 1, 0, has_bias=False)` on `nn.AdaptiveAvgPool2d(1)` outputs:
[B, C, 1, 1] tensors.  Five of them per BiSeNet-R18 step cost 0.27 ms in the vendor library (11-18 us per forward, 31-44 us
per backward: naive kernels, zero fills, gradient casts) plus an autocast copy of each fp32 weight
(profiles/r04_eager_ops.txt).  `PooledConv2d` keeps the module's parameters and state-dict keys; on bf16 [B, C, 1, 1] HIP
inputs its forward is one launch that reads the fp32 master weight, its backward one launch for dx and dw (fp32, no cast).
Any other input takes the module's stock forward.  TSG_VEC_CONV=1|0 (default 1)."""
import os

import torch
import torch.nn as nn

from . import kernels as K

ENABLED = os.environ.get("TSG_VEC_CONV", "1") != "0"


class _PooledConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        y = K.provider().conv1x1_vec_fwd(x, weight)
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dx, dw = K.provider().conv1x1_vec_bwd(dy, x, weight, need_dx=ctx.needs_input_grad[0])
        return dx, dw


class PooledConv2d(nn.Conv2d):
    def forward(self, x):
        if (ENABLED and x.is_cuda and x.dim() == 4 and x.shape[2] == 1 and x.shape[3] == 1
                and self.weight.dtype == torch.float32
                and (x.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and x.dtype == torch.float32
                                                    and torch.get_autocast_dtype("cuda") == torch.bfloat16))):
            xb = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
            if K.provider().conv1x1_vec_supported(xb, self.weight):
                with torch.autocast("cuda", enabled=False):
                    return _PooledConvFn.apply(xb, self.weight)
        return super().forward(x)


def _eligible(m):
    return (isinstance(m, nn.Conv2d) and type(m).__name__ == "Conv2d" and m.kernel_size == (1, 1) and m.stride == (1, 1)
            and m.padding == (0, 0) and m.dilation == (1, 1) and m.groups == 1 and m.bias is None
            and m.padding_mode == "zeros" and m.in_channels % 16 == 0 and m.out_channels % 16 == 0)


def _is_global_pool(m):
    if isinstance(m, nn.AdaptiveAvgPool2d):
        o = m.output_size
        return o == 1 or o == (1, 1)
    return type(m).__name__ in ("GlobalAvgPool", "GlobalAvgPool2d")


def install_pooled_conv(module):
    """Re-class, in place, the bias-free 1x1 convolutions that FOLLOW A GLOBAL POOL inside an nn.Sequential (`nn.Sequential(
    AdaptiveAvgPool2d(1) | GlobalAvgPool, ConvBnRelu(.., 1, 1, 0), ...)`: seg_oprs.py:199-205, :222-231, bisenet
    network.py:34-39) — the only places their input is a [B, C, 1, 1] map.  Round 4 re-classed EVERY bias-free 1x1
    convolution (Bottleneck conv1 / conv3, shortcuts): they fell through to the stock forward, but each call paid the
    extra checks, and the class swap hid them from later `type(m) is nn.Conv2d` installers.  Returns how many
    were re-classed."""
    n = 0
    for seq in module.modules():
        if not isinstance(seq, nn.Sequential):
            continue
        pooled = False
        for child in seq.children():
            if _is_global_pool(child):
                pooled = True
                continue
            if not pooled:
                continue
            convs = [child] if isinstance(child, nn.Conv2d) else [c for c in child.children() if isinstance(c, nn.Conv2d)]
            for m in convs:
                if _eligible(m):
                    m.__class__ = PooledConv2d
                    n += 1
            if isinstance(child, (nn.Conv2d,)) and child.kernel_size != (1, 1):
                pooled = False                         # a spatial operator: what follows is no pooled vector any more
    return n
