"""Full-map 1x1 convolutions with a REPRODUCIBLE weight gradient (round 6).

The five bias-free 1x1 convolutions of BiSeNet-R18 that see whole feature maps — the three ResNet shortcut convolutions
(furnace/base_model/resnet.py:139-146, stride 2), SpatialPath.conv_1x1 (bisenet network.py:118) and FeatureFusion's
conv_1x1 (furnace/seg_opr/seg_oprs.py:220-223) — were the last place in the step where two runs on the same data disagree:
the vendor library computes dw = dy^T x, a reduction over 65 536 .. 262 144 pixels with 64 x 128 .. 256 x 256 outputs, as a
split-K GEMM whose slices meet in fp32 ATOMICS and whose result is rounded to bf16 (tools/r6/probe_pw_wgrad.py: never
run-to-run equal, 1.7e-3 .. 9.8e-3 from the float64 value).  Here the pixel axis is cut into `nb` chunks by a VIEW, one batched
GEMM (hipBLASLt through torch.bmm, bf16 operands, fp32 result) writes the nb partial [C_out, C_in] matrices and one sum over
the chunk axis folds them in a fixed order: bit-identical from run to run, 1e-7 .. 4e-7 from the float64 value, and no
slower over the five layers (FeatureFusion 86 -> 61 us, the others +0 .. +23 us: profiles/r06_pointwise_wgrad.txt).
The stride-1 layers also take their data gradient as a plain GEMM (dy [M, C_out] x w [C_out, C_in]: 67 against 99 us for
FeatureFusion); the stride-2 data gradient (a scatter into every other pixel) and every forward stay on the vendor library.

`PointwiseConv2d` keeps the module's parameters and state-dict keys; any input other than a bf16 channels_last HIP map takes
the stock forward.  TSG_PW_CONV=1|0 (default 1)."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

ENABLED = os.environ.get("TSG_PW_CONV", "1") != "0"
# TSG_PW_DGRAD_MM=1|0 (default 1): data gradient of the stride-1 layers as a GEMM on the [pixels, channels] views
_DGRAD_MM = os.environ.get("TSG_PW_DGRAD_MM", "1") != "0"
# TSG_PW_SUB_FWD_MM=1|0 (default 1): the shortcut convolution on the sub-sampled map (forward_subsampled) as a GEMM as well
_SUB_FWD_MM = os.environ.get("TSG_PW_SUB_FWD_MM", "1") != "0"
# TSG_SEG_DEFER_PW=1|0 (default 1): the weight gradient joins a deferred-launch list when one is open (convwrw._DEFER)
_DEFER_OK = os.environ.get("TSG_SEG_DEFER_PW", "1") != "0"
_MIN_CHUNK = 128        # rows of a chunk: below this the batched GEMM's tiles run half empty


def _chunks(M):
    """number of chunks the pixel axis is cut into: the largest power of two <= 128 that divides M and leaves >= 128 rows"""
    nb = 128
    while nb > 1 and (M % nb or M // nb < _MIN_CHUNK):
        nb //= 2
    return nb


def _rows(t):
    """[B, C, H, W] channels_last -> its [B * H * W, C] view (no copy)"""
    B, C, H, W = t.shape
    return t.permute(0, 2, 3, 1).reshape(B * H * W, C)


def pointwise_wgrad(x, dy, stride, out=None):
    """dw [C_out, C_in, 1, 1] fp32 of a 1x1 / padding 0 convolution: x [B, C_in, H, W], dy [B, C_out, OH, OW], both bf16
    channels_last.  Chunked batched GEMM + ordered fold (see the module text).  out: a dense fp32 [C_out, C_in, 1, 1] tensor
    to write the result into (a deferred launch, convwrw._DEFER: autograd already holds that tensor)."""
    if stride != 1:
        x = x[:, :, ::stride, ::stride].contiguous(memory_format=torch.channels_last)     # the pixels the convolution read
    xr, dr = _rows(x), _rows(dy)
    M, K = xr.shape
    N = dr.shape[1]
    nb = _chunks(M)
    part = torch.bmm(dr.view(nb, M // nb, N).transpose(1, 2), xr.view(nb, M // nb, K), out_dtype=torch.float32)
    if out is not None:
        o2 = out.view(N, K)
        if nb > 1:
            torch.sum(part, dim=0, out=o2)
        else:
            o2.copy_(part[0])
        return out
    dw = part.sum(0) if nb > 1 else part[0]
    return dw.view(N, K, 1, 1)


class _PointwiseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, stride, fwd_mm=False):
        from .convwrw import _SHADOW
        if _SHADOW and weight.is_leaf and weight.is_cuda:
            from .shadow import bank                   # bf16 copy kept fresh once per optimizer step (no cast launch per call)
            wb, _ = bank.get(weight)
        else:
            wb = weight.detach().to(torch.bfloat16)
        if fwd_mm and stride == 1:
            # y [pixels, C_out] = x [pixels, C_in] w^T: the shortcut on the sub-sampled map has shapes the vendor library's
            # tuned database does not hold (its immediate-mode pick for [16, 256, 32, 32] -> 512 was a split-K kernel with
            # an fp32 result and a cast pass: 97 us for 4 GFLOP, profiles/r06_shortcut_on_subsampled_map.txt)
            B, K, H, W = x.shape
            y = torch.mm(_rows(x), wb.reshape(wb.shape[0], K).t()).view(B, H, W, wb.shape[0]).permute(0, 3, 1, 2)
        else:
            y = F.conv2d(x, wb, None, stride, 0)
        ctx.save_for_backward(x, wb)
        ctx.stride = stride
        ctx.wdtype = weight.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wb = ctx.saved_tensors
        st = ctx.stride
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = None
        if ctx.needs_input_grad[0]:
            if st == 1 and _DGRAD_MM:
                B, K, H, W = x.shape
                dx = torch.mm(_rows(dy), wb.reshape(wb.shape[0], K)).view(B, H, W, K).permute(0, 3, 1, 2)
            else:
                dx = torch.ops.aten.convolution_backward(dy, x, wb, None, [st, st], [0, 0], [1, 1], False, [0, 0], 1,
                                                         [True, False, False])[0]
        dw = None
        if ctx.needs_input_grad[1]:
            from . import convwrw
            if convwrw._DEFER is not None and _DEFER_OK and ctx.wdtype == torch.float32:
                # bench.SegmentedStep: listed, launched later from a graph of its own beside the next part of the backward;
                # autograd gets the still unwritten result (an alias of its own: see convwrw.wrw_on_side_stream)
                buf = torch.empty((dy.shape[1], x.shape[1], 1, 1), dtype=torch.float32, device=dy.device)
                def later():                                   # (outside backward: grad mode is on again, x may require grad)
                    with torch.no_grad():
                        pointwise_wgrad(x, dy, st, out=buf)
                convwrw._DEFER.append((later, (x, dy), buf))
                dw = buf.detach()
            else:
                dw = pointwise_wgrad(x, dy, st)
                if dw.dtype != ctx.wdtype:
                    dw = dw.to(ctx.wdtype)
        return dx, dw, None, None


class PointwiseConv2d(nn.Conv2d):
    def takes(self, x):
        """would forward(x) run the functions of this module (rather than the stock convolution)?"""
        return (ENABLED and isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4 and x.shape[2] * x.shape[3] > 1
                and torch.is_grad_enabled() and self.weight.requires_grad and self.weight.dtype == torch.float32
                and (x.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and x.dtype == torch.float32
                                                    and torch.get_autocast_dtype("cuda") == torch.bfloat16)))

    def forward(self, x):
        if self.takes(x):
            xb = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
            xb = xb.contiguous(memory_format=torch.channels_last)
            with torch.autocast("cuda", enabled=False):
                return _PointwiseFn.apply(xb, self.weight, self.stride[0])
        return super().forward(x)

    def forward_subsampled(self, xs):
        """this (stride-s) convolution of x, given xs = x[:, :, ::s, ::s] as a compact bf16 channels_last tensor: a stride-1
        convolution of xs (convwrw.conv_with_skip(subsample=True) hands a residual block's shortcut exactly that)"""
        with torch.autocast("cuda", enabled=False):
            return _PointwiseFn.apply(xs.contiguous(memory_format=torch.channels_last), self.weight, 1, _SUB_FWD_MM)


def _eligible(m):
    return (type(m) is nn.Conv2d and m.kernel_size == (1, 1) and m.stride in ((1, 1), (2, 2)) and m.padding == (0, 0)
            and m.dilation == (1, 1) and m.groups == 1 and m.bias is None and m.padding_mode == "zeros"
            and m.in_channels % 8 == 0 and m.out_channels % 8 == 0)


def install_pointwise_conv(module):
    """Re-class, in place, the bias-free 1x1 convolutions no other installer has taken (run AFTER install_cls_head and
    install_pooled_conv: `type(m) is nn.Conv2d` leaves their classes alone); returns how many were re-classed."""
    n = 0
    for m in module.modules():
        if _eligible(m):
            m.__class__ = PointwiseConv2d
            n += 1
    return n
