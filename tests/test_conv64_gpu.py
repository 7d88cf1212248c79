"""GPU parity of tsg_conv3x3_c64_fwd (through the C-ABI) with oracle/conv_ref.py on the same bf16-rounded operands
(fp64 accumulation): y is bf16 -> one bf16 ulp of the fp64 result (2^-8 relative) plus 1e-3 of the output scale for
cancellation.  The statistics epilogue must fold to the sums of exactly the activation that was written; fed
(dy, rot180(w)^T) the kernel is the data gradient of the same convolution."""
import numpy as np
import pytest
import torch

from oracle import conv_ref

pytestmark = pytest.mark.gpu

SHAPES = [(2, 8, 32), (1, 5, 37), (3, 4, 64), (1, 33, 70), (2, 64, 64), (1, 1, 1)]


def _run(cuda, B, H, W, seed, with_stats=False):
    from torchseg_amd import kernels as K
    kp = K.provider()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 64, H, W, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    xb = x.to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    wb = w.to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    assert kp.conv3x3_c64_supported(xb, wb, 1, 1, 1, 1)
    out = kp.conv3x3_c64_fwd(xb, wb, with_stats)
    y = out[0] if with_stats else out
    assert y.shape == xb.shape and y.is_contiguous(memory_format=torch.channels_last) and y.dtype == torch.bfloat16
    y_ref = conv_ref.conv2d_ref(conv_ref.bf16_round(x), conv_ref.bf16_round(w), stride=1, pad=1)
    err = (y.double().cpu() - y_ref).abs()
    bound = y_ref.abs() * 2.0 ** -8 + 1e-3 * y_ref.abs().max()
    assert bool((err <= bound).all()), (err.max().item(), y_ref.abs().max().item())
    return (x, w, xb, wb) + (out if with_stats else (y,))


@pytest.mark.parametrize("shape", SHAPES)
def test_conv64_forward_vs_oracle(cuda, shape):
    _run(cuda, *shape, seed=sum(shape))


def test_conv64_statistics_epilogue_and_determinism(cuda):
    from torchseg_amd import kernels as K
    kp = K.provider()
    x, w, xb, wb, y, partial = _run(cuda, 2, 70, 96, seed=3, with_stats=True)
    assert torch.equal(y, kp.conv3x3_c64_fwd(xb, wb))
    sums = partial.double().sum(0).cpu()
    yf = y.double().cpu()
    ref = torch.stack([yf.sum((0, 2, 3)), (yf * yf).sum((0, 2, 3))])
    np.testing.assert_allclose(sums.numpy(), ref.numpy(), rtol=2e-5, atol=1e-3)
    y2, p2 = kp.conv3x3_c64_fwd(xb, wb, True)
    assert torch.equal(y, y2) and torch.equal(partial, p2)


def test_conv64_is_the_data_gradient_with_the_rotated_filter(cuda):
    from torchseg_amd import kernels as K
    kp = K.provider()
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 64, 24, 40, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    dy = torch.randn(2, 64, 24, 40, generator=g)
    wb = w.to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    dyb = dy.to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    dx = kp.conv3x3_c64_fwd(dyb, kp.conv3x3_weight_rot180_t(wb))
    xr = x.double().requires_grad_(True)
    torch.nn.functional.conv2d(xr, conv_ref.bf16_round(w), None, 1, 1).backward(conv_ref.bf16_round(dy))
    err = (dx.double().cpu() - xr.grad).abs()
    assert bool((err <= xr.grad.abs() * 2.0 ** -8 + 1e-3 * xr.grad.abs().max()).all()), err.max().item()


def test_conv64_full_size(cuda):
    """BASELINE config 2 geometry of layer1 ([B, 64, 256, 256]; B = 2 keeps the fp64 oracle to seconds)."""
    _run(cuda, 2, 256, 256, seed=5)


S2_SHAPES = [(2, 8, 32), (1, 5, 37), (2, 9, 64), (1, 33, 70), (2, 64, 64), (1, 1, 1), (1, 2, 2), (1, 130, 66)]


@pytest.mark.parametrize("shape", S2_SHAPES)
def test_conv64_stride2_forward_and_data_gradient_vs_fp64(cuda, shape):
    from torchseg_amd import kernels as K
    kp = K.provider()
    B, H, W = shape
    g = torch.Generator().manual_seed(sum(shape) + 1)
    x = torch.randn(B, 64, H, W, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    dy = torch.randn(B, 64, OH, OW, generator=g)
    xb = x.to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    wb = w.to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    dyb = dy.to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    assert kp.conv3x3_c64_supported(xb, wb, 2, 1, 1, 1)
    y, partial = kp.conv3x3_c64_fwd(xb, wb, True, stride=2)
    assert tuple(y.shape) == (B, 64, OH, OW) and y.is_contiguous(memory_format=torch.channels_last)
    xr = conv_ref.bf16_round(x).requires_grad_(True)
    y_ref = torch.nn.functional.conv2d(xr, conv_ref.bf16_round(w), None, 2, 1)
    err = (y.double().cpu() - y_ref.detach()).abs()
    assert bool((err <= y_ref.detach().abs() * 2.0 ** -8 + 1e-3 * y_ref.detach().abs().max()).all()), err.max().item()
    np.testing.assert_allclose(conv_ref.conv2d_ref(conv_ref.bf16_round(x), conv_ref.bf16_round(w), stride=2, pad=1).numpy(),
                               y_ref.detach().numpy(), rtol=1e-10, atol=1e-10)       # the oracle states the same thing
    yf = y.double().cpu()
    np.testing.assert_allclose(partial.double().sum(0).cpu().numpy(),
                               torch.stack([yf.sum((0, 2, 3)), (yf * yf).sum((0, 2, 3))]).numpy(), rtol=2e-5, atol=1e-3)
    assert torch.equal(y, kp.conv3x3_c64_fwd(xb, wb, stride=2))
    y_ref.backward(conv_ref.bf16_round(dy))
    dx = kp.conv3x3_c64_s2_dgrad(dyb, kp.conv3x3_weight_rot180_t(wb), (H, W))
    assert tuple(dx.shape) == (B, 64, H, W) and dx.is_contiguous(memory_format=torch.channels_last)
    err = (dx.double().cpu() - xr.grad).abs()
    assert bool((err <= xr.grad.abs() * 2.0 ** -8 + 1e-3 * xr.grad.abs().max()).all()), err.max().item()


# ---- round 6: the BatchNorm backward sums in the epilogue of the data gradient ------------------------------------------
BSUM_SHAPES = [(2, 8, 32), (1, 5, 37), (2, 9, 64), (1, 33, 70), (2, 64, 64), (1, 1, 1), (1, 2, 2), (1, 130, 66), (3, 96, 160)]


def _bsum_case(cuda, B, H, W, stride, seed):
    """dy of a 64 -> 64 convolution of the given stride whose INPUT (size H x W) was relu(bn(x)); returns everything the
    fused launch takes plus the BatchNorm forward pack [3, 64] = a, b, mean."""
    g = torch.Generator().manual_seed(seed)
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    x = (torch.randn(B, 64, H, W, generator=g) * 1.5 + 0.3)
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    dy = torch.randn(B, 64, OH, OW, generator=g)
    a = torch.rand(64, generator=g) + 0.5
    a[::7] *= -1.0                                               # negative gamma: the mask flips
    b = torch.randn(64, generator=g) * 0.5
    mean = torch.randn(64, generator=g) * 0.2 + 0.3
    fp = torch.stack([a, b, mean]).to(cuda).contiguous()
    cl = dict(memory_format=torch.channels_last)
    return (x.to(cuda).bfloat16().contiguous(**cl), w.to(cuda).bfloat16().contiguous(**cl),
            dy.to(cuda).bfloat16().contiguous(**cl), fp)


@pytest.mark.parametrize("stride", [1, 2])
@pytest.mark.parametrize("shape", BSUM_SHAPES)
def test_conv64_data_gradient_with_bn_backward_sums(cuda, shape, stride):
    """The fused launch writes the SAME gradient as the plain one (bit-equal) and its partial rows fold to the sums
    tsg_bn_bwd_reduce computes from that stored gradient in a pass of its own: same terms (bf16-rounded gradient, mask
    a x + b > 0, x - mean in fp32), other summation order -> 2e-5 of sum |term| per channel; and to the fp64 sums of the
    oracle's definition (syncbn_ref: sum dy m, sum dy m (x - mean))."""
    from torchseg_amd import kernels as K
    kp = K.provider()
    B, H, W = shape
    if not kp.conv3x3_c64_bnsums_supported(B, H, W, stride):
        pytest.skip("shape not covered by the fused form")
    xb, wb, dyb, fp = _bsum_case(cuda, B, H, W, stride, seed=sum(shape) + stride)
    rot = kp.conv3x3_weight_rot180_t(wb)
    if stride == 1:
        plain = kp.conv3x3_c64_fwd(dyb, rot)
        dx, partial = kp.conv3x3_c64_fwd(dyb, rot, bsum=(xb, fp))
        dx2, partial2 = kp.conv3x3_c64_fwd(dyb, rot, bsum=(xb, fp))
    else:
        plain = kp.conv3x3_c64_s2_dgrad(dyb, rot, (H, W))
        dx, partial = kp.conv3x3_c64_s2_dgrad(dyb, rot, (H, W), bsum=(xb, fp))
        dx2, partial2 = kp.conv3x3_c64_s2_dgrad(dyb, rot, (H, W), bsum=(xb, fp))
    assert torch.equal(dx, plain)
    assert torch.equal(dx, dx2) and torch.equal(partial, partial2)            # run-to-run reproducible
    layout, N, C, HW = K.bn_layout(xb)
    sep, S = kp.bn_bwd_reduce(dx, xb, None, layout, N, C, HW, fp, True)
    got = partial.double().sum(0).cpu()
    want = sep[:S].double().sum(0).cpu()
    # fp64 definition on the stored values
    d = dx.double().cpu()
    xv = xb.double().cpu()
    a, b, mu = (fp[i].double().cpu().view(1, 64, 1, 1) for i in range(3))
    m = ((xb.float().cpu() * fp[0].cpu().view(1, 64, 1, 1) + fp[1].cpu().view(1, 64, 1, 1)) > 0).double()   # the fp32 fma's sign
    t1, t2 = d * m, d * m * (xv - mu)
    ref = torch.stack([t1.sum((0, 2, 3)), t2.sum((0, 2, 3))])
    scale = torch.stack([t1.abs().sum((0, 2, 3)), t2.abs().sum((0, 2, 3))]) + 1e-6
    assert bool(((got - want).abs() <= 2e-5 * scale).all()), ((got - want).abs() / scale).max().item()
    assert bool(((got - ref).abs() <= 2e-5 * scale).all()), ((got - ref).abs() / scale).max().item()


@pytest.mark.parametrize("stride", [1, 2])
def test_bn_relu_conv_node_takes_the_fused_sums(cuda, stride, monkeypatch):
    """conv(relu(bn(x))) as one autograd node (convwrw.bn_relu_conv): with the sums from the data gradient's epilogue the
    node returns the gradients of the path with the separate pass up to the fp32 summation order of 2 x 64 numbers
    (dx: one bf16 ulp where the coefficients differ in the last bit; dgamma / dbeta: 1e-5), and no bn_bwd_reduce launch."""
    from torchseg_amd import convwrw, kernels as K
    from torchseg_amd.syncbn import SyncBatchNorm
    kp = K.provider()
    g = torch.Generator().manual_seed(5 + stride)
    x = torch.randn(2, 64, 40, 72, generator=g).to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    dy_shape = (2, 64, (40 - 1) // stride + 1, (72 - 1) // stride + 1)
    dy = torch.randn(dy_shape, generator=g).to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)

    def run(fused):
        monkeypatch.setattr(convwrw, "_BN_BSUM", fused)
        torch.manual_seed(3)
        bn = SyncBatchNorm(64).to(cuda)
        with torch.no_grad():
            bn.weight.copy_(torch.rand(64, generator=torch.Generator().manual_seed(1)) + 0.5)
            bn.bias.copy_(torch.randn(64, generator=torch.Generator().manual_seed(2)) * 0.3)
        conv = torch.nn.Conv2d(64, 64, 3, stride, 1, bias=False).to(cuda).to(memory_format=torch.channels_last)
        convwrw.install_conv_wrw(conv)
        xin = x.clone().requires_grad_(True)
        cnt = K.CallCounter(kp)
        try:
            y = convwrw.bn_relu_conv(bn, torch.nn.ReLU(), xin, conv)
            y.backward(dy)
        finally:
            counts = cnt.stop()
        torch.cuda.synchronize()
        return xin.grad, bn.weight.grad, bn.bias.grad, conv.weight.grad, counts

    gx0, gg0, gb0, gw0, c0 = run(False)
    gx1, gg1, gb1, gw1, c1 = run(True)
    assert c0.get("bn_bwd_reduce", 0) == 1 and c1.get("bn_bwd_reduce", 0) == 0
    assert torch.equal(gw0, gw1)
    np.testing.assert_allclose(gg1.cpu().numpy(), gg0.cpu().numpy(), rtol=1e-5, atol=1e-5 * gg0.abs().max().item())
    np.testing.assert_allclose(gb1.cpu().numpy(), gb0.cpu().numpy(), rtol=1e-5, atol=1e-5 * gb0.abs().max().item())
    d = (gx1.float() - gx0.float()).abs()
    assert bool((d <= gx0.float().abs() * 2.0 ** -7 + 1e-6).all()), d.max().item()
