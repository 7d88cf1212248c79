"""ResNet backbones with the surface of furnace/base_model/resnet.py:104-224:
`resnetNN(pretrained_model=None, norm_layer=, bn_eps=, bn_momentum=, deep_stem=,
stem_width=, inplace=)`, forward returns the four stage outputs (resnet.py:168-184),
parameter names (conv1/bn1/layerN.M.convK/bnK/downsample.0/1) unchanged so
ImageNet checkpoints load.

MI355X path: with torchseg_amd's SyncBatchNorm every BN -> ReLU and the block
tail BN -> (+identity) -> ReLU (resnet.py:44-51, 92-99) is one fused normalise
kernel forward and one fused pair of kernels backward.
"""
import torch.nn as nn

from seg_opr.seg_oprs import norm_act
from torchseg_amd.fusion import outside_mode as _outside_mode
from torchseg_amd.pool import MaxPool2d as _MaxPool2d
from utils.pyt_utils import load_model

__all__ = ['ResNet', 'resnet18', 'resnet34', 'resnet50', 'resnet101', 'resnet152']


def _shortcut(inplanes, outplanes, stride, norm_layer, bn_eps, bn_momentum):
    return nn.Sequential(nn.Conv2d(inplanes, outplanes, kernel_size=1, stride=stride, bias=False),
                         norm_layer(outplanes, eps=bn_eps, momentum=bn_momentum))


class _Block(nn.Module):
    def _identity(self, x):
        if self.downsample is None:
            return x
        conv, bn = self.downsample[0], self.downsample[1]
        return norm_act(bn, None, conv(x))


class BasicBlock(_Block):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, norm_layer=None, bn_eps=1e-5, bn_momentum=0.1,
                 downsample=None, inplace=True):
        super(BasicBlock, self).__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = norm_layer(planes, eps=bn_eps, momentum=bn_momentum)
        self.relu = nn.ReLU(inplace=inplace)
        self.relu_inplace = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = norm_layer(planes, eps=bn_eps, momentum=bn_momentum)
        self.downsample = downsample
        self.stride = stride
        self.inplace = inplace

    def forward(self, x):
        if x.is_cuda:                                   # bn1 + relu applied while conv2 loads its input, where covered
            from torchseg_amd.convwrw import bn_relu_conv, conv_with_skip
            skip = None
            if self.downsample is None and self.stride == 1:
                # the skip connection is x itself: route it through conv1's autograd node, whose data-gradient kernel
                # then adds the skip path's gradient in its epilogue (no separate pass over three tensors)
                c1, skip = conv_with_skip(self.conv1, x)
                residual = skip if skip is not None else x
            elif self.downsample is not None and self.stride == 2:
                # the shortcut branch (1x1 / stride 2 convolution + BN) reads the alias: its gradient reaches conv1's
                # node and joins the stride-2 data gradient in that kernel's epilogue.  Where the shortcut convolution is
                # torchseg_amd.pwconv's, the alias is the COMPACT x[:, :, ::2, ::2] and the shortcut a stride-1 convolution
                # of it: its gradient comes back on the small grid and is added at the even pixels (no zero-filled map)
                sc = self.downsample[0]
                sub = type(sc).__name__ == "PointwiseConv2d" and sc.stride == (2, 2) and sc.takes(x)
                c1, skip = conv_with_skip(self.conv1, x, subsample=sub)
                if skip is not None and sub and skip.shape[2:] != x.shape[2:]:
                    residual = norm_act(self.downsample[1], None, sc.forward_subsampled(skip))
                else:
                    residual = self._identity(skip if skip is not None else x)
            else:
                c1 = self.conv1(x)
                residual = self._identity(x)
            out = bn_relu_conv(self.bn1, self.relu, c1, self.conv2)
            return norm_act(self.bn2, self.relu_inplace, out, residual=residual)
        out = self.conv2(norm_act(self.bn1, self.relu, self.conv1(x)))
        return norm_act(self.bn2, self.relu_inplace, out, residual=self._identity(x))


class Bottleneck(_Block):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, norm_layer=None, bn_eps=1e-5, bn_momentum=0.1,
                 downsample=None, inplace=True):
        super(Bottleneck, self).__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = norm_layer(planes, eps=bn_eps, momentum=bn_momentum)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = norm_layer(planes, eps=bn_eps, momentum=bn_momentum)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, kernel_size=1, bias=False)
        self.bn3 = norm_layer(planes * self.expansion, eps=bn_eps, momentum=bn_momentum)
        self.relu = nn.ReLU(inplace=inplace)
        self.relu_inplace = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride
        self.inplace = inplace

    def forward(self, x):
        out = norm_act(self.bn1, self.relu, self.conv1(x))
        out = norm_act(self.bn2, self.relu, self.conv2(out))
        return norm_act(self.bn3, self.relu_inplace, self.conv3(out), residual=self._identity(x))


class ResNet(nn.Module):
    def __init__(self, block, layers, norm_layer=nn.BatchNorm2d, bn_eps=1e-5, bn_momentum=0.1,
                 deep_stem=False, stem_width=32, inplace=True):
        super(ResNet, self).__init__()
        self.inplanes = stem_width * 2 if deep_stem else 64
        if deep_stem:  # "v1c" stem: three 3x3 convs (resnet.py:110-124)
            self.conv1 = nn.Sequential(
                nn.Conv2d(3, stem_width, kernel_size=3, stride=2, padding=1, bias=False),
                norm_layer(stem_width, eps=bn_eps, momentum=bn_momentum),
                nn.ReLU(inplace=inplace),
                nn.Conv2d(stem_width, stem_width, kernel_size=3, stride=1, padding=1, bias=False),
                norm_layer(stem_width, eps=bn_eps, momentum=bn_momentum),
                nn.ReLU(inplace=inplace),
                nn.Conv2d(stem_width, stem_width * 2, kernel_size=3, stride=1, padding=1, bias=False))
        else:
            self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(self.inplanes, eps=bn_eps, momentum=bn_momentum)
        self.relu = nn.ReLU(inplace=inplace)
        self.maxpool = _MaxPool2d(kernel_size=3, stride=2, padding=1)
        cfg = dict(bn_eps=bn_eps, bn_momentum=bn_momentum)
        self.layer1 = self._make_layer(block, norm_layer, 64, layers[0], inplace, **cfg)
        self.layer2 = self._make_layer(block, norm_layer, 128, layers[1], inplace, stride=2, **cfg)
        self.layer3 = self._make_layer(block, norm_layer, 256, layers[2], inplace, stride=2, **cfg)
        self.layer4 = self._make_layer(block, norm_layer, 512, layers[3], inplace, stride=2, **cfg)

    def _make_layer(self, block, norm_layer, planes, blocks, inplace=True, stride=1, bn_eps=1e-5,
                    bn_momentum=0.1):
        out_planes = planes * block.expansion
        downsample = None
        if stride != 1 or self.inplanes != out_planes:
            downsample = _shortcut(self.inplanes, out_planes, stride, norm_layer, bn_eps, bn_momentum)
        stages = [block(self.inplanes, planes, stride, norm_layer, bn_eps, bn_momentum, downsample, inplace)]
        self.inplanes = out_planes
        for _ in range(1, blocks):
            stages.append(block(self.inplanes, planes, norm_layer=norm_layer, bn_eps=bn_eps,
                                bn_momentum=bn_momentum, inplace=inplace))
        return nn.Sequential(*stages)

    def _stem(self, x):
        if isinstance(self.conv1, nn.Sequential):
            c = self.conv1
            if x.is_cuda:
                from torchseg_amd.convwrw import bn_relu_conv
                x = bn_relu_conv(c[1], c[2], c[0](x), c[3])
            else:
                x = c[3](norm_act(c[1], c[2], c[0](x)))
            x = norm_act(c[4], c[5], x)
            x = c[6](x)
        else:
            if x.is_cuda:
                from torchseg_amd.syncbn import stem_bn_relu_maxpool
                y = stem_bn_relu_maxpool(self.conv1, self.bn1, x, self.maxpool)   # the whole stem as one recomputing node
                if y is not None:
                    return y
            x = self.conv1(x)
        if x.is_cuda:
            from torchseg_amd.syncbn import bn_relu_maxpool
            y = bn_relu_maxpool(self.bn1, x, self.maxpool)          # one fused pass per direction on HIP tensors
            if y is not None:
                return y
        return self.maxpool(norm_act(self.bn1, self.relu, x))

    @_outside_mode
    def forward(self, x):
        x = self._stem(x)
        blocks = []
        for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
            x = stage(x)
            blocks.append(x)
        return blocks


def _build(block, layers, pretrained_model, kwargs):
    model = ResNet(block, layers, **kwargs)
    if pretrained_model is not None:
        model = load_model(model, pretrained_model)
    return model


def resnet18(pretrained_model=None, **kwargs):
    return _build(BasicBlock, [2, 2, 2, 2], pretrained_model, kwargs)


def resnet34(pretrained_model=None, **kwargs):
    return _build(BasicBlock, [3, 4, 6, 3], pretrained_model, kwargs)


def resnet50(pretrained_model=None, **kwargs):
    return _build(Bottleneck, [3, 4, 6, 3], pretrained_model, kwargs)


def resnet101(pretrained_model=None, **kwargs):
    return _build(Bottleneck, [3, 4, 23, 3], pretrained_model, kwargs)


def resnet152(pretrained_model=None, **kwargs):
    return _build(Bottleneck, [3, 8, 36, 3], pretrained_model, kwargs)
