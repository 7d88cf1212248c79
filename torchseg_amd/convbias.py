"""Conv2d bias handled outside the convolution.

Eager PyTorch-ROCm computes a conv's bias gradient as `grad_output.sum((0,2,3))`
on the channels_last gradient; for the 19-channel classifier convs of the
segmentation heads (bisenet network.py:151-156) that reduce_kernel takes ~525 us
per head per step (profiles/r01).  The DDP wrapper therefore re-classes every
plain nn.Conv2d that has a bias to `BiasSplitConv2d` (same parameters, same
state-dict keys): the convolution runs bias-free and the bias add / bias
gradient use our streaming column-sum kernel (the SyncBN statistics kernel).
"""
import torch
import torch.nn as nn

from . import kernels as K


class _AddBiasFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, bias):
        ctx.bias_dtype = bias.dtype
        return y + bias.to(y.dtype).view(1, -1, 1, 1)

    @staticmethod
    def backward(ctx, dy):
        kp = K.provider()
        lay = K.bn_layout(dy)
        if lay is None:
            dy = dy.contiguous()
            lay = K.bn_layout(dy)
        layout, N, C, HW = lay
        partial, S = kp.bn_stats(dy, layout, N, C, HW)          # row 0: per-channel sum over N, H, W
        sums = torch.empty(2 * C, dtype=torch.float32, device=dy.device)
        kp.bn_collapse(partial, S, C, sums)
        return dy, sums[:C].to(ctx.bias_dtype)


class BiasSplitConv2d(nn.Conv2d):
    def forward(self, x):
        if self.bias is None or not x.is_cuda:
            return super().forward(x)
        y = self._conv_forward(x, self.weight, None)
        if y.dtype not in (torch.float32, torch.bfloat16):
            return y + self.bias.to(y.dtype).view(1, -1, 1, 1)
        return _AddBiasFn.apply(y, self.bias)


def split_conv_bias(module):
    """Re-class plain nn.Conv2d layers that carry a bias (in place)."""
    n = 0
    for m in module.modules():
        if type(m) is nn.Conv2d and m.bias is not None:
            m.__class__ = BiasSplitConv2d
            n += 1
    return n
