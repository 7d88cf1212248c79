"""DFN (ResNet-101 v1c): smooth network (RRB + CAB) + border network — BASELINE config 4.

Architecture of model/dfn/cityscapes.dfn.R101_v1c/network.py:14-172 (DFN :14-152,
DFNHead :155-172) on the furnace surface, attribute names / construction order as
the reference.  Four cross-entropy heads + four sigmoid-focal border heads; five
parameters (border_aft_rrbs.0.*) are never used in forward."""
import torch.nn as nn
import torch.nn.functional as F

from . import ensure_furnace_on_path, head_loss
from .. import workloads as _w

ensure_furnace_on_path()
from base_model import resnet101  # noqa: E402
from seg_opr.seg_oprs import ChannelAttention, ConvBnRelu, RefineResidual  # noqa: E402


def _up(x, size=None, scale=None):
    return F.interpolate(x, size=size, scale_factor=scale, mode='bilinear', align_corners=True)


class DFNHead(nn.Module):
    def __init__(self, in_planes, out_planes, scale, norm_layer=nn.BatchNorm2d):
        super(DFNHead, self).__init__()
        self.rrb = RefineResidual(in_planes, out_planes * 9, 3, has_bias=False, has_relu=False, norm_layer=norm_layer)
        self.conv = nn.Conv2d(out_planes * 9, out_planes, kernel_size=1, stride=1, padding=0)
        self.scale = scale

    def forward(self, x):
        return _up(self.conv(self.rrb(x)), scale=self.scale)


class DFN(nn.Module):
    tsg_native_fusions = True      # calls the fused operators itself (workloads/__init__.py)

    def __init__(self, out_planes, criterion, aux_criterion, alpha, pretrained_model=None,
                 norm_layer=nn.BatchNorm2d, bn_eps=1e-5, bn_momentum=0.1, backbone=resnet101):
        super(DFN, self).__init__()
        self.backbone = backbone(pretrained_model, norm_layer=norm_layer, bn_eps=bn_eps, bn_momentum=bn_momentum,
                                 deep_stem=True, stem_width=64)
        inner = 512
        self.global_context = nn.Sequential(
            nn.AdaptiveAvgPool2d(1),
            ConvBnRelu(2048, inner, 1, 1, 0, has_bn=True, has_relu=True, has_bias=False, norm_layer=norm_layer))
        stage = [2048, 1024, 512, 256]
        rrb = dict(has_bias=False, has_relu=True, norm_layer=norm_layer)
        pre, cabs, aft, heads = [], [], [], []
        for i, ch in enumerate(stage):
            pre.append(RefineResidual(ch, inner, 3, **rrb))
            cabs.append(ChannelAttention(inner * 2, inner, 1))
            aft.append(RefineResidual(inner, inner, 3, **rrb))
            heads.append(DFNHead(inner, out_planes, 2 ** (5 - i), norm_layer=norm_layer))
        border = 21
        bpre, baft, bheads = [], [], []
        for ch in reversed(stage):
            bpre.append(RefineResidual(ch, border, 3, **rrb))
            baft.append(RefineResidual(border, border, 3, **rrb))
            bheads.append(DFNHead(border, 1, 4, norm_layer=norm_layer))
        self.smooth_pre_rrbs = nn.ModuleList(pre)
        self.cabs = nn.ModuleList(cabs)
        self.smooth_aft_rrbs = nn.ModuleList(aft)
        self.smooth_heads = nn.ModuleList(heads)
        self.border_pre_rrbs = nn.ModuleList(bpre)
        self.border_aft_rrbs = nn.ModuleList(baft)
        self.border_heads = nn.ModuleList(bheads)
        self.business_layer = [self.global_context, self.smooth_pre_rrbs, self.cabs, self.smooth_aft_rrbs,
                               self.smooth_heads, self.border_pre_rrbs, self.border_aft_rrbs, self.border_heads]
        self.criterion = criterion
        self.aux_criterion = aux_criterion
        self.alpha = alpha

    def forward(self, data, label=None, aux_label=None):
        blocks = self.backbone(data)
        deep_first = blocks[::-1]
        last_fm = _up(self.global_context(deep_first[0]), size=deep_first[0].shape[2:])
        pred_out = []
        for i, (fm, pre, cab, aft, head) in enumerate(zip(deep_first, self.smooth_pre_rrbs, self.cabs,
                                                          self.smooth_aft_rrbs, self.smooth_heads)):
            fm = aft(cab(pre(fm), last_fm))
            pred_out.append(head(fm))
            if i != 3:
                last_fm = _up(fm, scale=2)
        last_fm = None
        border_out = []
        for i, (fm, pre, aft, head) in enumerate(zip(blocks, self.border_pre_rrbs, self.border_aft_rrbs,
                                                     self.border_heads)):
            fm = pre(fm)
            if last_fm is not None:
                if fm.is_cuda and _w.NATIVE_FUSIONS:            # dfn network.py:130-133 as one kernel: up(fm) + last_fm
                    from ..upsample import upsample_bilinear_ac
                    last_fm = aft(upsample_bilinear_ac(fm, scale_factor=2 ** i, add=last_fm))
                else:
                    last_fm = aft(last_fm + _up(fm, scale=2 ** i))
            else:
                last_fm = fm
            border_out.append(head(last_fm))
        if label is not None and aux_label is not None:
            loss = sum(head_loss(self.criterion, p, label) for p in pred_out)       # network.py:140-143
            aux = sum(self.aux_criterion(b, aux_label) for b in border_out)
            return loss + self.alpha * aux                                      # network.py:150-152
        return F.log_softmax(pred_out[-1], dim=1)
