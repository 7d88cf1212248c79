"""The classifier convolution of a segmentation head on its own kernels (csrc/clshead.hip).

`nn.Conv2d(C_in, n_classes, kernel_size=1)` with bias — bisenet network.py:151-161 (`BiSeNetHead.conv_1x1`, 256 / 64 ->
19), the 19-class heads of dfn — is a stream over the feature map with a matrix that fits in registers.  The vendor
library's implicit-GEMM kernels plus the layout copies around them (channels_last logits -> the planar logits the
criterion kernels read, and back for the gradient) and the bias passes cost 0.38 ms of BiSeNet's 14 ms step
(profiles/r04_eager_ops.txt).  `ClsHeadConv2d` keeps the module's parameters and state-dict keys; on bf16 channels_last
HIP activations its forward writes PLANAR (NCHW-contiguous) logits straight from the MFMA accumulators and its backward
consumes the planar logit gradient the criterion produces.  TSG_CLS_HEAD=1|0 (default 1)."""
import os

import torch
import torch.nn as nn

from . import kernels as K

ENABLED = os.environ.get("TSG_CLS_HEAD", "1") != "0"


class _ClsHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        z = K.provider().cls_head_fwd(x, weight, bias)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.bias_dtype = bias.dtype if bias is not None else None
        return z

    @staticmethod
    def backward(ctx, dz):
        x, weight = ctx.saved_tensors
        if dz.dtype != torch.bfloat16:
            dz = dz.to(torch.bfloat16)
        dz = dz.contiguous()                                 # planar: what tsg_ohem_up_bwd / tsg_ohem_bwd hand back
        dx, dw, db = K.provider().cls_head_bwd(dz, x, weight, need_dx=ctx.needs_input_grad[0], need_db=ctx.has_bias)
        return dx, dw.to(weight.dtype), (db.to(ctx.bias_dtype) if db is not None else None)


class ClsHeadConv2d(nn.Conv2d):
    def forward(self, x):
        if (ENABLED and x.is_cuda and x.dim() == 4 and self.weight.dtype == torch.float32
                and (x.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and x.dtype == torch.float32
                                                    and torch.get_autocast_dtype("cuda") == torch.bfloat16))):
            xb = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
            if xb.is_contiguous(memory_format=torch.channels_last) and not xb.is_contiguous() \
                    and K.provider().cls_head_supported(xb, self.weight):
                with torch.autocast("cuda", enabled=False):
                    w = self.weight if self.weight.is_contiguous() else self.weight.contiguous()
                    return _ClsHeadFn.apply(xb, w, self.bias)
        if self.bias is not None and x.is_cuda:              # what BiasSplitConv2d did for this module before
            from .convbias import BiasSplitConv2d
            return BiasSplitConv2d.forward(self, x)
        return super().forward(x)


def _eligible(m):
    return (isinstance(m, nn.Conv2d) and type(m).__name__ in ("Conv2d", "BiasSplitConv2d") and m.kernel_size == (1, 1)
            and m.stride == (1, 1) and m.padding == (0, 0) and m.dilation == (1, 1) and m.groups == 1
            and m.padding_mode == "zeros" and m.out_channels <= 32 and m.in_channels in (32, 64, 128, 256))


def install_cls_head(module):
    """Re-class the classifier convolutions in place; returns how many were found."""
    n = 0
    for m in module.modules():
        if _eligible(m):
            m.__class__ = ClsHeadConv2d
            n += 1
    return n
