"""Composite blocks with the constructor signatures and sub-module attribute
names of furnace/seg_opr/seg_oprs.py:24-238 (the attribute names are the
checkpoint keys of the released models).

MI355X path: when the injected `norm_layer` is torchseg_amd's SyncBatchNorm the
BN -> ReLU pair (seg_oprs.py:39-46) runs as ONE normalise kernel (and one fused
backward) instead of two element-wise passes; any other norm layer (e.g.
nn.BatchNorm2d for the CPU plumbing config) takes the plain module sequence.
"""
import torch
import torch.nn as nn

from torchseg_amd import fusion as _fusion
from torchseg_amd.pool import GlobalAvgPool as _GlobalAvgPool, cat_channels, channel_scale, gated_scale
from torchseg_amd.syncbn import SyncBatchNorm as _FusedBN


def norm_act(bn, relu, x, residual=None):
    """bn(x) [+ residual] [-> relu] with the fused kernel when available."""
    if isinstance(bn, _FusedBN):
        return bn(x, residual=residual, relu=relu is not None)
    x = bn(x)
    if residual is not None:
        x = x + residual
    return relu(x) if relu is not None else x


def one_hot(index_tensor, cls_num):
    """[B,H,W] integer map -> [B,cls,H,W] float one-hot on the input's device (seg_oprs.py:14-21)."""
    b, h, w = index_tensor.size()
    out = torch.zeros(b, cls_num, h, w, dtype=torch.float32, device=index_tensor.device)
    return out.scatter_(1, index_tensor.view(b, 1, h, w).long(), 1)


class _ConvNormAct(nn.Module):
    """shared body of ConvBnRelu / DeConvBnRelu: attributes conv, bn, relu."""

    def _finish(self, out_planes, has_bn, norm_layer, bn_eps, has_relu, inplace):
        self.has_bn = has_bn
        if has_bn:
            self.bn = norm_layer(out_planes, eps=bn_eps)
        self.has_relu = has_relu
        if has_relu:
            self.relu = nn.ReLU(inplace=inplace)

    tsg_accepts_pending = True     # fusion.FuseMode's forward pre-hook leaves a PendingCbr argument to us

    @_fusion.outside_mode(keeps_chain=True)
    def forward(self, x):
        if isinstance(x, torch.Tensor) and x.dim() == 4 and x.shape[2] == 1 and x.shape[3] == 1 and x.is_cuda:
            from torchseg_amd.vecconv import pooled_layer
            y = pooled_layer(self, x)                   # a pooled map: convolution + BatchNorm + ReLU as one launch
            if y is not None:
                return y
        mode = _fusion.CHAIN_ACTIVE
        if mode or isinstance(x, _fusion.PendingCbr):
            return self._forward_chain(x, mode)
        x = self.conv(x)
        relu = self.relu if self.has_relu else None
        if self.has_bn:
            return norm_act(self.bn, relu, x)
        return relu(x) if relu is not None else x

    def _forward_chain(self, x, mode):
        """Under fusion.FuseMode(chain=True), i.e. around an unchanged network.py that calls ConvBnRelu modules one after
        the other (bisenet network.py:131-137): a module whose BatchNorm + ReLU the NEXT convolution could apply on its
        load path (64 output channels: convwrw.bn_relu_conv) returns its result pending; a pending input is consumed
        here.  What runs is what `cbr_chain` below runs for our own builders."""
        can_defer = (bool(mode) and self.has_bn and self.has_relu and isinstance(self.bn, _FusedBN)
                     and isinstance(self.conv, nn.Conv2d) and self.conv.out_channels == 64
                     and not self._forward_hooks and not self._forward_pre_hooks)
        if isinstance(x, _fusion.PendingCbr):
            y = x.feed(self)
        elif can_defer and isinstance(x, torch.Tensor) and _fusion._on_device(x):
            return mode.defer_cbr(self, x=x)
        else:
            y = self.conv(x)
        if can_defer and _fusion._on_device(y):
            return mode.defer_cbr(self, y=y)
        relu = self.relu if self.has_relu else None
        if self.has_bn:
            return norm_act(self.bn, relu, y)
        return relu(y) if relu is not None else y


def cbr_chain(mods, x):
    """`mods[-1](... mods[1](mods[0](x)))` for consecutive ConvBnRelu modules (bisenet network.py:133-137, SpatialPath).
    Between two of them the BatchNorm + ReLU of the first is handed to the convolution of the second
    (torchseg_amd.convwrw.bn_relu_conv: applied while that convolution loads its input when both are on the HIP path, the
    plain module sequence otherwise); results are those of calling the modules one after the other."""
    if not (x.is_cuda and all(isinstance(m, ConvBnRelu) and m.has_bn and m.has_relu for m in mods[:-1])):
        for m in mods:
            x = m(x)
        return x
    from torchseg_amd.convwrw import bn_relu_conv, stem_bn_relu_conv
    first = 1
    y = stem_bn_relu_conv(mods[0].conv, mods[0].bn, mods[0].relu, x, mods[1].conv) if len(mods) > 1 else None
    if y is not None:                                 # image stem + its BN + the next convolution as one autograd node
        x, first = y, 2
    else:
        x = mods[0].conv(x)
    for prev, m in zip(mods[first - 1:-1], mods[first:]):
        x = bn_relu_conv(prev.bn, prev.relu, x, m.conv)
    last = mods[-1]
    relu = last.relu if last.has_relu else None
    if last.has_bn:
        return norm_act(last.bn, relu, x)
    return relu(x) if relu is not None else x


class ConvBnRelu(_ConvNormAct):
    def __init__(self, in_planes, out_planes, ksize, stride, pad, dilation=1, groups=1, has_bn=True,
                 norm_layer=nn.BatchNorm2d, bn_eps=1e-5, has_relu=True, inplace=True, has_bias=False):
        super(ConvBnRelu, self).__init__()
        self.conv = nn.Conv2d(in_planes, out_planes, kernel_size=ksize, stride=stride, padding=pad,
                              dilation=dilation, groups=groups, bias=has_bias)
        self._finish(out_planes, has_bn, norm_layer, bn_eps, has_relu, inplace)


class DeConvBnRelu(_ConvNormAct):
    def __init__(self, in_planes, out_planes, ksize, stride, pad, output_pad, dilation=1, groups=1,
                 has_bn=True, norm_layer=nn.BatchNorm2d, bn_eps=1e-5, has_relu=True, inplace=True,
                 has_bias=False):
        super(DeConvBnRelu, self).__init__()
        self.conv = nn.ConvTranspose2d(in_planes, out_planes, kernel_size=ksize, stride=stride,
                                       padding=pad, output_padding=output_pad, dilation=dilation,
                                       groups=groups, bias=has_bias)
        self._finish(out_planes, has_bn, norm_layer, bn_eps, has_relu, inplace)


class SeparableConvBnRelu(nn.Module):
    """depthwise conv -> bn -> point-wise ConvBnRelu (seg_oprs.py:76-94)."""

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=0, dilation=1,
                 has_relu=True, norm_layer=nn.BatchNorm2d):
        super(SeparableConvBnRelu, self).__init__()
        self.conv1 = nn.Conv2d(in_channels, in_channels, kernel_size, stride, padding, dilation,
                               groups=in_channels, bias=False)
        self.bn = norm_layer(in_channels)
        self.point_wise_cbr = ConvBnRelu(in_channels, out_channels, 1, 1, 0, has_bn=True,
                                         norm_layer=norm_layer, has_relu=has_relu, has_bias=False)

    def forward(self, x):
        return self.point_wise_cbr(norm_act(self.bn, None, self.conv1(x)))


class GlobalAvgPool2d(nn.Module):
    """[B,C,H,W] -> [B,C,1,1] mean (seg_oprs.py:97-107)."""

    def forward(self, inputs):
        if inputs.is_cuda and inputs.dim() == 4:
            from torchseg_amd.pool import global_avg_pool
            return global_avg_pool(inputs)
        b, c = inputs.size(0), inputs.size(1)
        return inputs.reshape(b, c, -1).mean(dim=2).view(b, c, 1, 1)


class SELayer(nn.Module):
    def __init__(self, in_planes, out_planes, reduction=16):
        super(SELayer, self).__init__()
        self.avg_pool = _GlobalAvgPool(1)
        self.fc = nn.Sequential(nn.Linear(in_planes, out_planes // reduction), nn.ReLU(inplace=True),
                                nn.Linear(out_planes // reduction, out_planes), nn.Sigmoid())
        self.out_planes = out_planes

    def forward(self, x):
        b, c = x.size(0), x.size(1)
        gate = self.fc(self.avg_pool(x).view(b, c))
        return gate.view(b, self.out_planes, 1, 1)


class ChannelAttention(nn.Module):
    """DFN CAB: x1 * SE(cat(x1,x2)) + x2 (seg_oprs.py:130-140)."""

    def __init__(self, in_planes, out_planes, reduction):
        super(ChannelAttention, self).__init__()
        self.channel_attention = SELayer(in_planes, out_planes, reduction)

    def forward(self, x1, x2):
        gate = self.channel_attention(torch.cat([x1, x2], 1))
        return x1 * gate + x2


class BNRefine(nn.Module):
    def __init__(self, in_planes, out_planes, ksize, has_bias=False, has_relu=False,
                 norm_layer=nn.BatchNorm2d, bn_eps=1e-5):
        super(BNRefine, self).__init__()
        self.conv_bn_relu = ConvBnRelu(in_planes, out_planes, ksize, 1, ksize // 2, has_bias=has_bias,
                                       norm_layer=norm_layer, bn_eps=bn_eps)
        self.conv_refine = nn.Conv2d(out_planes, out_planes, kernel_size=ksize, stride=1,
                                     padding=ksize // 2, dilation=1, bias=has_bias)
        self.has_relu = has_relu
        if has_relu:
            self.relu = nn.ReLU(inplace=False)

    def forward(self, x):
        y = self.conv_refine(self.conv_bn_relu(x)) + x
        return self.relu(y) if self.has_relu else y


class RefineResidual(nn.Module):
    """DFN RRB: 1x1 conv, then residual (CBR -> conv) branch (seg_oprs.py:165-188)."""

    def __init__(self, in_planes, out_planes, ksize, has_bias=False, has_relu=False,
                 norm_layer=nn.BatchNorm2d, bn_eps=1e-5):
        super(RefineResidual, self).__init__()
        self.conv_1x1 = nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=1, padding=0,
                                  dilation=1, bias=has_bias)
        self.cbr = ConvBnRelu(out_planes, out_planes, ksize, 1, ksize // 2, has_bias=has_bias,
                              norm_layer=norm_layer, bn_eps=bn_eps)
        self.conv_refine = nn.Conv2d(out_planes, out_planes, kernel_size=ksize, stride=1,
                                     padding=ksize // 2, dilation=1, bias=has_bias)
        self.has_relu = has_relu
        if has_relu:
            self.relu = nn.ReLU(inplace=False)

    def forward(self, x):
        x = self.conv_1x1(x)
        y = self.conv_refine(self.cbr(x)) + x
        return self.relu(y) if self.has_relu else y


class AttentionRefinement(nn.Module):
    """BiSeNet ARM: 3x3 CBR, gated by sigmoid(BN(1x1 conv(GAP))) (seg_oprs.py:192-212)."""

    def __init__(self, in_planes, out_planes, norm_layer=nn.BatchNorm2d):
        super(AttentionRefinement, self).__init__()
        self.conv_3x3 = ConvBnRelu(in_planes, out_planes, 3, 1, 1, has_bn=True, norm_layer=norm_layer,
                                   has_relu=True, has_bias=False)
        self.channel_attention = nn.Sequential(
            _GlobalAvgPool(1),
            ConvBnRelu(out_planes, out_planes, 1, 1, 0, has_bn=True, norm_layer=norm_layer,
                       has_relu=False, has_bias=False),
            nn.Sigmoid())

    @_fusion.outside_mode
    def forward(self, x):
        fm = self.conv_3x3(x)
        return gated_scale(fm, self.channel_attention)            # fm * fm_se (seg_oprs.py:209-210)


class FeatureFusion(nn.Module):
    """BiSeNet FFM: 1x1 CBR of cat(x1,x2), then fm + fm * SE(fm) (seg_oprs.py:215-238)."""

    def __init__(self, in_planes, out_planes, reduction=1, norm_layer=nn.BatchNorm2d):
        super(FeatureFusion, self).__init__()
        self.conv_1x1 = ConvBnRelu(in_planes, out_planes, 1, 1, 0, has_bn=True, norm_layer=norm_layer,
                                   has_relu=True, has_bias=False)
        self.channel_attention = nn.Sequential(
            _GlobalAvgPool(1),
            ConvBnRelu(out_planes, out_planes // reduction, 1, 1, 0, has_bn=False, norm_layer=norm_layer,
                       has_relu=True, has_bias=False),
            ConvBnRelu(out_planes // reduction, out_planes, 1, 1, 0, has_bn=False, norm_layer=norm_layer,
                       has_relu=False, has_bias=False),
            nn.Sigmoid())

    @_fusion.outside_mode
    def forward(self, x1, x2):
        fm = self.conv_1x1(cat_channels(x1, x2))           # torch.cat([x1, x2], dim=1) on HIP channels_last maps
        return gated_scale(fm, self.channel_attention, add_identity=True)    # fm + fm * fm_se (seg_oprs.py:236-237)
