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
