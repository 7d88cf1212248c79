"""The recomputing ResNet stem (csrc/stemconv.hip, round 6; furnace/base_model/resnet.py:96-100,131-133): every pass re-evaluates
the 7x7/2 convolution instead of reading its 537 MB output.  Held against the materialising path it replaces
(tsg_stem_conv_fwd_stats -> tsg_bn_relu_pool_fwd / _bwd_reduce / _bwd_apply -> tsg_stem_conv_wrw), which in turn is held
against the oracle by tests/test_stemconv_gpu.py and tests/test_pool_gpu.py."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cuda():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


SHAPES = [(2, 64, 64), (1, 70, 96), (2, 22, 130), (1, 128, 256), (3, 36, 68), (2, 256, 512)]


def _case(cuda, shape):
    from torchseg_amd import kernels as K
    kp = K.provider()
    B, H, W = shape
    g = torch.Generator().manual_seed(3 * H + W)
    img = torch.randn(B, 3, H, W, generator=g).to(cuda).bfloat16()
    w = (torch.randn(64, 3, 7, 7, generator=g) * 0.1).to(cuda)
    y, partial = kp.stem_conv_fwd_stats(img, w)
    gamma = (torch.randn(64, generator=g) * 0.5 + 1.0).to(cuda)
    gamma[::9] *= -1.0
    beta = (torch.randn(64, generator=g) * 0.3).to(cuda)
    OH, OW = y.shape[2], y.shape[3]
    _, invstd, fp = kp.bn_finalize(partial, partial.shape[0], 64, float(B * OH * OW), None, 1e-5, 0.1, gamma, beta, None, None, None)
    return kp, img, w, y, partial, invstd, fp, g


@pytest.mark.parametrize("shape", SHAPES)
def test_statistics_without_the_activation(cuda, shape):
    kp, img, w, y, partial, invstd, fp, g = _case(cuda, shape)
    assert torch.equal(kp.stem_conv_stats(img, w), partial)


@pytest.mark.parametrize("shape", SHAPES)
def test_forward_equals_conv_then_bn_relu_pool(cuda, shape):
    kp, img, w, y, partial, invstd, fp, g = _case(cuda, shape)
    want_y, want_idx = kp.bn_relu_pool_fwd(y, fp)
    got_y, got_idx = kp.stem_conv_bn_relu_pool_fwd(img, w, fp)
    assert got_y.shape == want_y.shape and got_y.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got_y, want_y)
    assert torch.equal(got_idx, want_idx)


@pytest.mark.parametrize("shape", SHAPES)
def test_backward_sums_and_weight_gradient(cuda, shape):
    from torchseg_amd import kernels as K
    kp, img, w, y, partial, invstd, fp, g = _case(cuda, shape)
    B = img.shape[0]
    OH, OW = y.shape[2], y.shape[3]
    ypool, idx = kp.bn_relu_pool_fwd(y, fp)
    dpool = torch.randn(ypool.shape, generator=g).to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    # sums: the same terms in another order -> compare the fp64 folds, against the magnitude of the terms
    p_ref, S_ref = kp.bn_relu_pool_bwd_reduce(dpool, idx, y, fp)
    p_got, S_got = kp.stem_conv_bn_relu_pool_bwd_reduce(img, w, dpool, idx, fp)
    assert p_got.shape == (S_got, 2, 64)
    ref = p_ref[:S_ref].double().sum(0)
    got = p_got[:S_got].double().sum(0)
    scale = p_ref[:S_ref].double().abs().sum(0).clamp_min(1e-6)
    assert ((got - ref).abs() / scale).max().item() < 2e-5
    # weight gradient: the unfused pair at ITS tile partition (768 blocks) against ours (512): the same products, fp32 partial
    # sums in another grouping; both against the float64 product of the same bf16 operands
    _, _, bp = kp.bn_bwd_coeffs(p_ref, S_ref, 64, float(B * OH * OW), None, True, invstd, fp, True, True)
    dy = kp.bn_relu_pool_bwd_apply(dpool, idx, y, bp)
    dw_ref = kp.stem_conv_wrw(img, dy)
    dw_got = kp.stem_conv_wrw_bn_pool(img, w, dpool, idx, bp)              # y re-evaluated
    dw_rd = kp.stem_conv_wrw_bn_pool(img, w, dpool, idx, bp, xc=y)          # y read: the unfused pair's tile partition
    assert torch.equal(dw_rd, dw_ref)
    dw64 = torch.nn.grad.conv2d_weight(img.double(), (64, 3, 7, 7), dy.double().contiguous(), stride=2, padding=3)
    den = dw64.abs().max().item()
    e_ref = (dw_ref.double() - dw64).abs().max().item() / den
    e_got = (dw_got.double() - dw64).abs().max().item() / den
    assert e_got < 2e-6 and e_got <= 4 * e_ref + 1e-7, (e_got, e_ref)


def test_module_level_node_equals_the_three_modules(cuda):
    """ResNet._stem as the one node (TSG_STEM_RECOMPUTE = 2: gradient never stored; 1: nothing stored) against the three
    modules with their own nodes (0): pooled output and running statistics bit for bit, gradients of conv1 / bn1 to the
    rounding of another summation order."""
    import torch.nn as nn
    from torchseg_amd import syncbn
    from torchseg_amd.stemconv import StemConv2d
    from torchseg_amd.syncbn import SyncBatchNorm
    res = {}
    for mode in (2, 1, 0):
        torch.manual_seed(5)
        conv = StemConv2d(3, 64, 7, 2, 3, bias=False).to(cuda)
        bn = SyncBatchNorm(64).to(cuda)
        with torch.no_grad():
            bn.weight.copy_(torch.randn(64, device=cuda) * 0.5 + 1.0)
            bn.bias.copy_(torch.randn(64, device=cuda) * 0.2)
        pool = nn.MaxPool2d(3, 2, 1)
        img = torch.randn(2, 3, 128, 192, device=cuda)
        old = syncbn._STEM_RECOMPUTE
        syncbn._STEM_RECOMPUTE = mode
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = syncbn.stem_bn_relu_maxpool(conv, bn, img, pool)
                assert (out is not None) == (mode != 0)
                if out is None:
                    x = conv(img)
                    out = syncbn.bn_relu_maxpool(bn, x, pool)
                    assert out is not None
            gout = torch.randn(out.shape, device=cuda, generator=torch.Generator(device=cuda).manual_seed(9)).to(out.dtype)
            out.backward(gout.contiguous(memory_format=torch.channels_last))
        finally:
            syncbn._STEM_RECOMPUTE = old
        torch.cuda.synchronize()
        res[mode] = (out.detach().clone(), bn.running_mean.clone(), bn.running_var.clone(), conv.weight.grad.clone(),
                     bn.weight.grad.clone(), bn.bias.grad.clone())
    b = res[0]
    for mode in (2, 1):
        a = res[mode]
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
        for i in (3, 4, 5):
            err = ((a[i].double() - b[i].double()).norm() / b[i].double().norm().clamp_min(1e-12)).item()
            assert err < 2e-3, (mode, i, err)
    # mode 2 reads the same stored activation for its sums as mode 0: BN gradients bit for bit
    assert torch.equal(res[2][4], b[4]) and torch.equal(res[2][5], b[5])
