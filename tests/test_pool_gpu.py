"""GPU parity: global average pool kernels (both layouts, fwd + bwd) vs torch CPU mean."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(2, 64, 7, 5), (4, 128, 32, 32), (16, 256, 128, 128), (3, 19, 9, 9), (2, 4096, 3, 3),
                                   (2, 512, 1, 1)])
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_global_avg_pool(cuda, shape, layout, dtype):
    from torchseg_amd.pool import GlobalAvgPool
    g = torch.Generator().manual_seed(shape[1])
    x = torch.randn(shape, generator=g).to(dtype)
    dy = torch.randn(shape[0], shape[1], 1, 1, generator=g).to(dtype)
    fmt = torch.channels_last if layout == "nhwc" else torch.contiguous_format
    xd = x.to(cuda).contiguous(memory_format=fmt).requires_grad_(True)
    y = GlobalAvgPool(1)(xd)
    assert y.shape == (shape[0], shape[1], 1, 1)
    y.backward(dy.to(cuda))
    xr = x.double().requires_grad_(True)
    yr = xr.mean((2, 3), keepdim=True)
    yr.backward(dy.double())
    tol = dict(rtol=1e-5, atol=1e-6) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(y.detach().cpu().double(), yr.detach(), **tol)
    torch.testing.assert_close(xd.grad.cpu().double(), xr.grad, **tol)
    assert xd.grad.stride() == xd.stride()


@pytest.mark.parametrize("shape", [(2, 64, 7, 5), (4, 128, 32, 32), (16, 256, 128, 128), (3, 19, 9, 9), (2, 4096, 3, 3)])
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("ident", [False, True])
def test_channel_scale(cuda, shape, layout, dtype, ident):
    from torchseg_amd.pool import channel_scale
    g = torch.Generator().manual_seed(shape[1] + 1)
    x = torch.randn(shape, generator=g).to(dtype)
    s = torch.sigmoid(torch.randn(shape[0], shape[1], 1, 1, generator=g)).to(dtype)
    dy = torch.randn(shape, generator=g).to(dtype)
    fmt = torch.channels_last if layout == "nhwc" else torch.contiguous_format
    xd = x.to(cuda).contiguous(memory_format=fmt).requires_grad_(True)
    sd = s.to(cuda).requires_grad_(True)
    y = channel_scale(xd, sd, ident)
    y.backward(dy.to(cuda).contiguous(memory_format=fmt))
    xr, sr = x.double().requires_grad_(True), s.double().requires_grad_(True)
    yr = xr + xr * sr if ident else xr * sr
    yr.backward(dy.double())
    n = shape[2] * shape[3]
    if dtype == torch.float32:
        torch.testing.assert_close(y.detach().cpu().double(), yr.detach(), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(xd.grad.cpu().double(), xr.grad, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(sd.grad.cpu().double(), sr.grad, rtol=1e-4, atol=1e-4 * n ** 0.5)
    else:
        torch.testing.assert_close(y.detach().cpu().double(), yr.detach(), rtol=1e-2, atol=1e-2)
        torch.testing.assert_close(xd.grad.cpu().double(), xr.grad, rtol=1e-2, atol=1e-2)
        torch.testing.assert_close(sd.grad.cpu().double(), sr.grad, rtol=2e-2, atol=2e-2 * n ** 0.5)


def test_bias_split_conv_matches_plain_conv(cuda):
    import torch.nn as nn
    from torchseg_amd.convbias import split_conv_bias
    torch.manual_seed(0)
    a = nn.Conv2d(32, 19, 1).to(cuda)
    b = nn.Conv2d(32, 19, 1).to(cuda)
    b.load_state_dict(a.state_dict())
    assert split_conv_bias(b) == 1
    x = torch.randn(4, 32, 24, 40, device=cuda).contiguous(memory_format=torch.channels_last)
    g = torch.randn(4, 19, 24, 40, device=cuda)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = a(xa), b(xb)
    ya.backward(g); yb.backward(g)
    torch.testing.assert_close(yb, ya, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(xb.grad, xa.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(b.weight.grad, a.weight.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(b.bias.grad, a.bias.grad, rtol=1e-4, atol=1e-3)
    assert list(a.state_dict().keys()) == list(b.state_dict().keys())


@pytest.mark.parametrize("shape,k,s,p", [((2, 64, 32, 32), 3, 2, 1), ((2, 64, 33, 47), 3, 2, 1), ((4, 8, 16, 16), 2, 2, 0),
                                         ((1, 128, 20, 12), 3, 1, 1), ((16, 64, 512, 512), 3, 2, 1)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_maxpool_channels_last(cuda, shape, k, s, p, dtype):
    from torchseg_amd.pool import MaxPool2d
    big = shape[0] * shape[1] * shape[2] * shape[3] > 5e7
    if big and dtype == torch.float32:
        pytest.skip("large case once is enough")
    g = torch.Generator().manual_seed(shape[2])
    x = torch.randn(shape, generator=g)
    x[0, :, 0, 0] = x[0, :, 0, 1]                     # a tie: the first maximum must take the gradient
    x = x.to(dtype)
    xd = x.to(cuda).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = MaxPool2d(k, s, p)(xd)
    assert y.is_contiguous(memory_format=torch.channels_last)
    dy = torch.randn(y.shape, generator=g).to(dtype)
    y.backward(dy.to(cuda))
    xr = x.float().requires_grad_(True)
    yr = torch.nn.functional.max_pool2d(xr, k, s, p)
    yr.backward(dy.float())
    torch.testing.assert_close(y.detach().float().cpu(), yr.detach(), rtol=0, atol=0)
    tol = dict(rtol=1e-6, atol=1e-6) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(xd.grad.float().cpu(), xr.grad, **tol)


@pytest.mark.parametrize("case", [(2, 2048, 90, 90, 6), (2, 2048, 90, 90, 3), (2, 2048, 90, 90, 2), (2, 512, 60, 60, 6),
                                  (3, 24, 7, 10, 3), (1, 64, 33, 47, (5, 4)), (2, 128, 8, 8, 8)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_adaptive_avg_pool_channels_last(cuda, case, dtype):
    """nn.AdaptiveAvgPool2d(s) on channels_last maps (PSPNet's pyramid pooling, pspnet network.py:75-109) against the
    CPU op the reference calls, in fp64 on the same (bf16-rounded) input: forward and backward, windows that overlap
    (90 -> 6 ... 7 -> 3) included.  Also through the module the DDP wrapper installs."""
    import torch.nn as nn
    import torch.nn.functional as F
    from torchseg_amd.pool import AdaptiveAvgPool2d, install_adaptive_pool
    N, C, H, W, out = case
    if dtype == torch.bfloat16 and C % 8:
        pytest.skip("bf16 vectors are 8 channels")
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(N, C, H, W, generator=g).to(dtype)
    m = nn.Sequential(nn.AdaptiveAvgPool2d(out))
    assert install_adaptive_pool(m) == 1 and isinstance(m[0], AdaptiveAvgPool2d)
    xd = x.to(cuda).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = m(xd)
    xr = x.double().requires_grad_(True)
    yr = F.adaptive_avg_pool2d(xr, out)
    assert tuple(y.shape) == tuple(yr.shape) and y.is_contiguous(memory_format=torch.channels_last)
    dy = torch.randn(yr.shape, generator=g).to(dtype)
    y.backward(dy.to(cuda).contiguous(memory_format=torch.channels_last))
    yr.backward(dy.double())
    tol = dict(rtol=1e-5, atol=1e-6) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(y.detach().cpu().double(), yr.detach(), **tol)
    torch.testing.assert_close(xd.grad.cpu().double(), xr.grad, **tol)
    assert xd.grad.is_contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("shape", [(16, 128, 128, 32, 32), (2, 64, 256, 7, 9), (1, 8, 24, 5, 3), (3, 128, 128, 1, 2)])
def test_channel_concatenation_equals_torch_cat_forward_and_backward(cuda, shape):
    """cat_channels (tsg_cat2_rows) for FeatureFusion's torch.cat([x1, x2], dim=1) (seg_oprs.py:233-235): bit-equal output and
    gradients, channels_last result; inputs it does not take (NCHW-contiguous, 1x1 maps) go to torch.cat."""
    from torchseg_amd.pool import cat_channels
    B, Ca, Cb, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    a0 = torch.randn(B, Ca, H, W, generator=g).to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    b0 = torch.randn(B, Cb, H, W, generator=g).to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, Ca + Cb, H, W, generator=g).to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    res = []
    for fn in (lambda a, b: cat_channels(a, b), lambda a, b: torch.cat([a, b], dim=1)):
        a, b = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        y = fn(a, b)
        y.backward(dy)
        res.append((y.detach(), a.grad, b.grad))
    assert res[0][0].is_contiguous(memory_format=torch.channels_last)
    for u, v in zip(res[0], res[1]):
        assert torch.equal(u, v)
    n = torch.randn(2, 16, 4, 4, device=cuda)               # fp32 NCHW-contiguous: the stock path
    assert torch.equal(cat_channels(n, n), torch.cat([n, n], 1))


# ---- round 6: a gate computed from the pooled map it gates (pool.gated_scale) ---------------------------------------------
@pytest.mark.parametrize("shape", [(2, 128, 24, 40), (3, 256, 16, 16), (2, 64, 33, 47), (1, 8, 5, 7)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("ident", [False, True])
def test_gated_scale_equals_pool_branch_and_channel_scale(cuda, shape, dtype, ident, monkeypatch):
    """`x * se(gap(x))` (+ x) with the linked autograd nodes (the gate's backward computes ds only, the pool node writes
    dx = dy s (+ dy) + g / HW in one pass) against the plain composition, whose map gradient autograd sums in a pass of its
    own: output, the map's gradient and the branch's parameter gradients are BIT-equal."""
    from torchseg_amd import kernels as K, pool
    kp = K.provider()
    N, C, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x0 = torch.randn(shape, generator=g).to(dtype)
    dy0 = torch.randn(shape, generator=g).to(dtype)
    torch.manual_seed(4)
    class Affine(torch.nn.Module):                      # (element-wise: a vendor convolution here is not run-to-run reproducible
        def __init__(self):                             #  at shapes outside its tuned database, which fp32 equality would see)
            super().__init__()
            self.w = torch.nn.Parameter(torch.randn(1, C, 1, 1) * 0.5 + 1.0)
            self.b = torch.nn.Parameter(torch.randn(1, C, 1, 1) * 0.3)

        def forward(self, v):
            return v * self.w + self.b

    branch = torch.nn.Sequential(pool.GlobalAvgPool(1), Affine(), torch.nn.Sigmoid()).to(cuda).to(dtype)

    def run(split):
        monkeypatch.setattr(pool, "_GATE_SPLIT", split)
        for p in branch.parameters():
            p.grad = None
        x = x0.to(cuda).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        cnt = K.CallCounter(kp)
        try:
            y = pool.gated_scale(x * 1.0 if False else x, branch, add_identity=ident)
            y.backward(dy0.to(cuda).contiguous(memory_format=torch.channels_last))
        finally:
            calls = cnt.stop()
        torch.cuda.synchronize()
        return [y.detach(), x.grad] + [p.grad.clone() for p in branch.parameters()], calls

    ref, c0 = run(False)
    got, c1 = run(True)
    assert c0.get("chanscale_bwd") == 1 and "chanscale_bwd_ds" not in c0
    assert c1.get("chanscale_bwd_ds") == 1 and c1.get("chanscale_bwd_dx") == 1 and "chanscale_bwd" not in c1, c1
    for name, a, b in zip(("y", "dx", "dw", "db"), got, ref):
        assert torch.equal(a, b), name


def test_gated_scale_falls_back_when_the_scale_does_not_come_from_the_pool(cuda):
    """a branch that cuts the graph between the pooled vector and the scale (detach): the pool node will never run in the
    backward pass, so the gate must write the map's gradient itself"""
    from torchseg_amd import pool

    class Cut(torch.nn.Module):
        def forward(self, v):
            return torch.sigmoid(v.detach() * 0.5 + self.w)

        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(1, 16, 1, 1))

    branch = torch.nn.Sequential(pool.GlobalAvgPool(1), Cut()).to(cuda)
    x = torch.randn(2, 16, 6, 10, device=cuda).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = pool.gated_scale(x, branch)
    y.sum().backward()
    s = torch.sigmoid(x.detach().mean((2, 3), keepdim=True) * 0.5)
    assert torch.allclose(x.grad, s.expand_as(x), atol=1e-6) and branch[1].w.grad is not None
