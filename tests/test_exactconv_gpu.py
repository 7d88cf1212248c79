"""GPU parity of the reference-accuracy fp32 convolution (csrc/convf32.hip, through the C-ABI) with oracle/conv_ref.py
(fp64, tap by tap) and torch's fp64 autograd: forward, data gradient and weight gradient must be the CORRECTLY ROUNDED
fp32 value of the exact result (products exact, fp64 accumulation, one rounding: <= 1 ulp), for every convolution form
the four networks use — 7x7 / stride 2 / padding 3 on a 3-channel NCHW image (resnet.py:96-97), 3x3 stride 1 and 2
(resnet.py:24-29), 1x1 stride 1 and 2 (the shortcuts, resnet.py:127-133), dilated 3x3 (pspnet network.py:62-72), ragged
sizes, NCHW and channels_last operands, C_out = 19 classifier heads — and it must be what an unchanged nn.Conv2d computes
inside ExactConvMode, bias included."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import conv_ref

pytestmark = pytest.mark.gpu

# (B, Cin, H, W, Cout, K, stride, pad, dil, channels_last)
CASES = [
    (2, 3, 33, 41, 64, 7, 2, 3, 1, False),
    (2, 64, 20, 27, 64, 3, 1, 1, 1, True),
    (1, 64, 21, 30, 128, 3, 2, 1, 1, True),
    (2, 96, 9, 11, 19, 1, 1, 0, 1, True),
    (2, 64, 16, 18, 128, 1, 2, 0, 1, False),
    (1, 80, 17, 15, 72, 3, 1, 2, 2, True),
    (1, 130, 12, 12, 70, 3, 1, 4, 4, False),
    (3, 5, 8, 8, 7, 5, 3, 1, 1, False),
    (2, 64, 64, 72, 64, 3, 1, 1, 1, True),                 # 9216 pixels: the weight gradient runs in 4 pixel slices
    (1, 3, 96, 100, 32, 7, 2, 3, 1, False),
]


def _ulp_ok(got, want64, what):
    got = got.double().cpu()
    err = (got - want64).abs()
    bound = want64.abs() * 2.0 ** -23 + 1e-30 + 2.0 ** -149
    bad = int((err > bound).sum())
    assert bad == 0, (what, bad, err.max().item(), want64.abs().max().item())


@pytest.mark.parametrize("case", CASES)
def test_forward_dgrad_wgrad_are_correctly_rounded(cuda, case):
    from torchseg_amd import kernels as K
    kp = K.provider()
    B, Cin, H, W, Cout, k, s, p, d, cl = case
    g = torch.Generator().manual_seed(sum(case[:9]))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (k * k * Cin)) ** 0.5
    fmt = torch.channels_last if cl else torch.contiguous_format
    xd = x.to(cuda).contiguous(memory_format=fmt)
    wd = w.to(cuda).contiguous(memory_format=fmt)
    y = kp.conv2d_f32_exact_fwd(xd, wd, (s, s), (p, p), (d, d))
    x64 = x.double().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, None, s, p, d)
    assert tuple(y.shape) == tuple(y64.shape)
    _ulp_ok(y, y64.detach(), "fwd")
    if d == 1:                                              # the tap-by-tap restatement (no dilation argument)
        _ulp_ok(y, conv_ref.conv2d_ref(x, w, stride=s, pad=p), "fwd vs conv_ref")
    dy = torch.randn(y64.shape, generator=g)
    y64.backward(dy.double())
    dyd = dy.to(cuda).contiguous(memory_format=fmt)
    dx = kp.conv2d_f32_exact_dgrad(dyd, wd, xd, (s, s), (p, p), (d, d))
    _ulp_ok(dx, x64.grad, "dgrad")
    dw = kp.conv2d_f32_exact_wgrad(xd, dyd, wd, (s, s), (p, p), (d, d))
    _ulp_ok(dw, w64.grad, "wgrad")
    assert torch.equal(y, kp.conv2d_f32_exact_fwd(xd, wd, (s, s), (p, p), (d, d)))       # run-to-run


def test_unchanged_modules_inside_the_mode(cuda):
    """nn.Conv2d with and without bias, and a stock F.conv2d call, inside ExactConvMode: forward and every gradient equal
    the fp64 evaluation to 1 ulp; outside the mode the vendor library runs (looser)."""
    from torchseg_amd.exactconv import ExactConvMode
    torch.manual_seed(0)
    net = nn.Sequential(nn.Conv2d(3, 32, 7, 2, 3, bias=False), nn.ReLU(), nn.Conv2d(32, 19, 3, 1, 1, bias=True)).to(cuda)
    ref = nn.Sequential(nn.Conv2d(3, 32, 7, 2, 3, bias=False), nn.ReLU(), nn.Conv2d(32, 19, 3, 1, 1, bias=True)).double()
    ref.load_state_dict({k_: v.double().cpu() for k_, v in net.state_dict().items()})
    x = torch.randn(2, 3, 40, 52)
    with ExactConvMode():
        y = net(x.to(cuda))
        y.square().sum().backward()
    y64 = ref(x.double())
    y64.square().sum().backward()
    # two layers deep: the second convolution sees inputs that already carry one rounding
    assert (y.double().cpu() - y64).abs().max().item() <= 4e-7 * y64.abs().max().item()
    for (n, p), q in zip(net.named_parameters(), ref.parameters()):
        rel = (p.grad.double().cpu() - q.grad).abs().max().item() / q.grad.abs().max().item()
        assert rel <= 2e-6, (n, rel)
