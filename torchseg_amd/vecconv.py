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


# TSG_POOLED_LAYER=1|0 (default 1, round 6): a ConvBnRelu on a pooled [B, C, 1, 1] map — convolution, BatchNorm over the batch,
# ReLU, and the nn.Sigmoid that ends an attention branch — as ONE launch per direction (tsg_conv1x1_vec_bnact_*).  Under graph
# replay every launch costs >= 4.8 us; these layers ran 9-10 launches each for 16 x C numbers.
POOLED_LAYER = os.environ.get("TSG_POOLED_LAYER", "1") != "0"


class _PooledLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, gamma, beta, bn, bnmode, act):
        kp = K.provider()
        if bnmode:
            track = bn.track_running_stats
            out, yc, stats = kp.conv1x1_vec_bnact_fwd(
                x, weight, bnmode, act, gamma.float() if gamma is not None else None,
                beta.float() if beta is not None else None,
                bn.running_mean if track else None, bn.running_var if track else None,
                bn.num_batches_tracked if (track and bnmode == 1) else None, bn.eps,
                0.0 if bn.momentum is None else bn.momentum)
        else:
            out, yc, stats = kp.conv1x1_vec_bnact_fwd(x, weight, 0, act)
        ctx.save_for_backward(x, weight, out, yc, stats, gamma, beta)
        ctx.cfg = (bnmode, act)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, out, yc, stats, gamma, beta = ctx.saved_tensors
        bnmode, act = ctx.cfg
        if dout.dtype != torch.bfloat16:
            dout = dout.to(torch.bfloat16)
        dx, dw, dgamma, dbeta = K.provider().conv1x1_vec_bnact_bwd(dout, out, yc, stats, x, weight, bnmode, act,
                                                                   need_dx=ctx.needs_input_grad[0])
        if gamma is None or not bnmode:
            dgamma = None
        else:
            dgamma = dgamma.to(gamma.dtype)
        if beta is None or not bnmode:
            dbeta = None
        else:
            dbeta = dbeta.to(beta.dtype)
        return dx, dw, dgamma, dbeta, None, None, None


def pooled_layer(m, x, sigmoid_after=False):
    """`m(x)` (and `torch.sigmoid` of it) for a ConvBnRelu-like module `m` (attributes conv, has_bn [bn], has_relu) on a pooled
    [B, C, 1, 1] HIP map as one fused node, or None when anything about it is not the plain case (then the caller runs the
    modules): conv a PooledConv2d, bn our single-process SyncBatchNorm, no hooks, bf16 path."""
    if not (POOLED_LAYER and ENABLED and isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4 and x.shape[2] == 1
            and x.shape[3] == 1):
        return None
    conv = getattr(m, "conv", None)
    if not isinstance(conv, PooledConv2d) or conv.weight.dtype != torch.float32:
        return None
    has_bn, has_relu = bool(getattr(m, "has_bn", False)), bool(getattr(m, "has_relu", False))
    if has_relu and sigmoid_after:
        return None
    if not (x.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and x.dtype == torch.float32
                                          and torch.get_autocast_dtype("cuda") == torch.bfloat16)):
        return None
    if m._forward_hooks or m._forward_pre_hooks or conv._forward_hooks or conv._forward_pre_hooks:
        return None
    bn, bnmode = None, 0
    if has_bn:
        from .syncbn import SyncBatchNorm, _world
        bn = m.bn
        if not isinstance(bn, SyncBatchNorm) or bn._forward_hooks or bn._forward_pre_hooks \
                or bn.num_features != conv.out_channels:
            return None
        use_batch_stats = bn.training or not bn.track_running_stats
        if use_batch_stats and (_world(bn.process_group) != 1 or x.shape[0] < 2 or bn.momentum is None):
            return None                                # statistics cross ranks / the stock path raises its own error
        bnmode = 1 if use_batch_stats else 2
        if has_relu and (m.relu._forward_hooks or m.relu._forward_pre_hooks):
            return None
    xb = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
    kp = K.provider()
    if not (hasattr(kp, "conv1x1_vec_bnact_fwd") and kp.conv1x1_vec_supported(xb, conv.weight)):
        return None
    act = 2 if sigmoid_after else (1 if has_relu else 0)
    with torch.autocast("cuda", enabled=False):
        return _PooledLayerFn.apply(xb, conv.weight, bn.weight if bn is not None else None,
                                    bn.bias if bn is not None else None, bn, bnmode, act)


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
