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
