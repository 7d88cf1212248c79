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
# TSG_WEIGHT_SHADOW=0|1 (default 0): bf16 / rotated filters from torchseg_amd.shadow (one refresh launch per step instead
# of ~45 cast / rotate launches).  Measured neutral on one MI355X (1032.5 vs 1035.5 img/s: the 4-us launches it removes sit
# back to back in the queue and cost the GPU almost nothing), so it stays opt-in for hosts that are launch-bound.
_SHADOW = _os.environ.get("TSG_WEIGHT_SHADOW", "0") == "1"


class _ConvWrwFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, wb, stride, wrt=None):
        # x bf16 channels_last, wb = weight rounded to bf16 (what autocast feeds the convolution)
        ctx.own = _OWN_C64 and K.provider().conv3x3_c64_supported(x, wb, stride, 1, 1, 1) \
            and wb.is_contiguous(memory_format=torch.channels_last)
        ctx.in_hw = (x.shape[2], x.shape[3])
        if ctx.own:
            y = K.provider().conv3x3_c64_fwd(x, wb, stride=stride)     # 64 -> 64: our kernels (csrc/conv64.hip)
        else:
            y = F.conv2d(x, wb, None, stride, 1)
        ctx.stride = stride
        ctx.wrt = wrt                                  # not a graph tensor: a shadow owned by torchseg_amd.shadow
        ctx.save_for_backward(x, wb)
        ctx.wdtype = weight.dtype
        ctx.need_dx = x.requires_grad
        ctx.dgrad_fwd = _DGRAD_FWD and stride == 1 and (_DGRAD_ANY or weight.shape[0] == weight.shape[1])
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wb = ctx.saved_tensors
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = None
        rot = (lambda: ctx.wrt) if ctx.wrt is not None else (lambda: K.provider().conv3x3_weight_rot180_t(wb))
        if ctx.need_dx:
            if ctx.own and ctx.stride == 2:
                dx = K.provider().conv3x3_c64_s2_dgrad(dy, rot(), ctx.in_hw)
            elif ctx.own:
                dx = K.provider().conv3x3_c64_fwd(dy, rot())
            elif ctx.dgrad_fwd:
                # dx = conv(dy, rot180(w)^T): the library's forward kernels beat its backward-data kernels on the
                # symmetric layers (tools/probe_conv2.py); same bf16 operands, fp32 accumulation
                dx = F.conv2d(dy, rot(), None, 1, 1)
            else:
                st = ctx.stride
                dx = torch.ops.aten.convolution_backward(dy, x, wb, None, [st, st], [1, 1], [1, 1], False, [0, 0], 1,
                                                         [True, False, False])[0]
        dw = K.provider().conv3x3_wrw(x, dy, stride=ctx.stride)
        return dx, dw.to(ctx.wdtype), None, None, None


class WrwConv2d(nn.Conv2d):
    def forward(self, x):
        if (x.is_cuda and self.bias is None and self.weight.dtype == torch.float32 and x.dim() == 4
                and self.padding_mode == "zeros" and torch.is_grad_enabled() and self.weight.requires_grad
                and (x.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and x.dtype == torch.float32
                                                    and torch.get_autocast_dtype("cuda") == torch.bfloat16))):
            xb = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
            xb = xb.contiguous(memory_format=torch.channels_last)
            if K.provider().conv3x3_wrw_supported(xb, self.weight, self.stride[0], self.padding[0], self.dilation[0],
                                                  self.groups):
                with torch.autocast("cuda", enabled=False):
                    if _SHADOW and self.weight.is_contiguous(memory_format=torch.channels_last):
                        from .shadow import bank           # bf16 (and rotated) filters kept fresh once per step
                        wb, wrt = bank.get(self.weight, want_rot=True)
                    else:
                        wb, wrt = self.weight.detach().to(torch.bfloat16), None
                    return _ConvWrwFn.apply(xb, self.weight, wb, self.stride[0], wrt)
        return super().forward(x)


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
