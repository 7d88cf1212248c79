"""TEST INFRASTRUCTURE (oracle): CPU restatement of the stem convolution.

The reference's stem is `nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)`
(furnace/base_model/resnet.py:96-97; model/bisenet/cityscapes.bisenet.R18/network.py:116 via
ConvBnRelu, furnace/seg_opr/seg_oprs.py:27-31), i.e. cross-correlation as torch defines it:

    y[b, o, i, j] = sum_{c, kh, kw} w[o, c, kh, kw] * x[b, c, s*i + kh - p, s*j + kw - p]

restated below tap by tap on zero-padded input (float64 accumulation), together with its weight
gradient dw[o, c, kh, kw] = sum_{b, i, j} dy[b, o, i, j] * x[b, c, s*i + kh - p, s*j + kw - p].
Pinned against torch's own CPU convolution and autograd in tests/test_oracles_cpu.py.
`bf16_round` reproduces the operand rounding autocast applies before the convolution."""
import torch
import torch.nn.functional as F


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float64)


def _taps(x, kh, kw, stride, pad, oh, ow):
    xp = F.pad(x, (pad, pad, pad, pad))
    return xp[:, :, kh: kh + stride * (oh - 1) + 1: stride, kw: kw + stride * (ow - 1) + 1: stride]


def conv2d_ref(x, w, stride=2, pad=3):
    x, w = x.double(), w.double()
    B, C, H, W = x.shape
    O, _, KH, KW = w.shape
    oh, ow = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    y = torch.zeros(B, O, oh, ow, dtype=torch.float64)
    for kh in range(KH):
        for kw in range(KW):
            y += torch.einsum("bchw,oc->bohw", _taps(x, kh, kw, stride, pad, oh, ow), w[:, :, kh, kw])
    return y


def conv2d_wgrad_ref(x, dy, ksize=7, stride=2, pad=3):
    x, dy = x.double(), dy.double()
    oh, ow = dy.shape[2:]
    dw = torch.zeros(dy.shape[1], x.shape[1], ksize, ksize, dtype=torch.float64)
    for kh in range(ksize):
        for kw in range(ksize):
            dw[:, :, kh, kw] = torch.einsum("bohw,bchw->oc", dy, _taps(x, kh, kw, stride, pad, oh, ow))
    return dw
