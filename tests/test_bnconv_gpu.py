"""GPU parity of normalise-on-load (torchseg_amd.convwrw.bn_relu_conv): conv(relu(bn(x))) for the 64 -> 64 3x3 layers with
the BatchNorm + ReLU applied while the convolution and its weight gradient load x.
 * kernel level: tsg_conv3x3_c64_*_fwd(in_ab) and tsg_conv3x3_wrw_*_norm equal the same kernels fed the materialised
   tsg_bn_apply_fwd output, bit for bit;
 * module level: the fused autograd node equals the module sequence (bn -> relu -> conv) in forward, in every gradient and
   in the running statistics, and both equal nn.BatchNorm2d + ReLU + Conv2d in fp64 on the CPU."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("stride", [1, 2])
@pytest.mark.parametrize("shape", [(2, 24, 40), (1, 33, 70), (2, 64, 64)])
def test_kernels_with_affine_on_load_equal_the_materialised_path(cuda, shape, stride):
    from torchseg_amd import kernels as K
    kp = K.provider()
    B, H, W = shape
    g = torch.Generator().manual_seed(H * 3 + W + stride)
    x = torch.randn(B, 64, H, W, generator=g).to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    fp = torch.stack([torch.randn(64, generator=g) * 0.5 + 1.0, torch.randn(64, generator=g) * 0.3,
                      torch.randn(64, generator=g) * 0.1]).to(cuda).contiguous()
    fp[0, ::7] *= -1.0
    layout, n, c, hw = K.bn_layout(x)
    a = kp.bn_apply_fwd(x, None, layout, n, c, hw, fp, True)
    y_ref = kp.conv3x3_c64_fwd(a, w, stride=stride)
    y, partial = kp.conv3x3_c64_fwd(x, w, True, stride=stride, in_ab=fp)
    assert torch.equal(y, y_ref)
    assert torch.equal(kp.conv3x3_c64_fwd(x, w, stride=stride, in_ab=fp), y_ref)
    dy = torch.randn(y.shape, generator=g).to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    dw_ref = kp.conv3x3_wrw(a, dy, stride=stride)
    assert torch.equal(kp.conv3x3_wrw(x, dy, stride=stride, in_ab=fp), dw_ref)
    if stride == 1:
        assert torch.equal(kp.conv3x3_wrw(x, dy, variant="gen", stride=1, in_ab=fp), kp.conv3x3_wrw(a, dy, variant="gen", stride=1))


@pytest.mark.parametrize("cin,cout,stride", [(64, 64, 1), (64, 64, 2),                  # conv64 kernels
                                             (128, 128, 1), (128, 256, 1), (256, 64, 1)])  # the general kernel (round 3)
def test_fused_node_equals_module_sequence_and_fp64(cuda, cin, cout, stride):
    from torchseg_amd import convwrw
    from torchseg_amd.convwrw import bn_relu_conv, install_conv_wrw
    from torchseg_amd.syncbn import SyncBatchNorm
    torch.manual_seed(3)

    class Pair(nn.Module):
        def __init__(self, bn_cls):
            super().__init__()
            self.bn = bn_cls(cin)
            self.relu = nn.ReLU()
            self.conv = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)

    ref = Pair(nn.BatchNorm2d).double()
    with torch.no_grad():
        ref.bn.weight.copy_(torch.randn(cin) * 0.4 + 1.0)
        ref.bn.bias.copy_(torch.randn(cin) * 0.2)
        ref.conv.weight.copy_(ref.conv.weight.float().bfloat16().double())
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, cin, 40, 48, generator=g).bfloat16().float()
    xr = x.double().requires_grad_(True)
    out_ref = ref.conv(ref.relu(ref.bn(xr)))
    dout = torch.randn(out_ref.shape, generator=g).bfloat16().float()
    out_ref.backward(dout.double())

    res = {}
    for fused in (True, False):
        net = Pair(SyncBatchNorm).to(cuda).to(memory_format=torch.channels_last)
        net.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
        net.bn.running_mean.zero_(); net.bn.running_var.fill_(1.0); net.bn.num_batches_tracked.zero_()
        assert install_conv_wrw(net) == 1
        xg = x.to(cuda).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        old, old_gen = convwrw._BN_ON_LOAD, convwrw._GEN_BN_ON_LOAD
        convwrw._BN_ON_LOAD = fused
        convwrw._GEN_BN_ON_LOAD = True                     # opt-in for the general kernel (TSG_CONV_GEN_BN_ON_LOAD=1)
        calls = []
        kp = convwrw.K.provider()
        orig = kp.bn_apply_fwd
        kp.bn_apply_fwd = lambda *a, **k: (calls.append("bn_apply_fwd"), orig(*a, **k))[1]
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = bn_relu_conv(net.bn, net.relu, xg, net.conv)
            out.backward(dout.to(cuda).to(out.dtype))
        finally:
            convwrw._BN_ON_LOAD, convwrw._GEN_BN_ON_LOAD = old, old_gen
            del kp.bn_apply_fwd
        assert calls == ([] if fused else ["bn_apply_fwd"])          # the normalised activation was never written
        res[fused] = [t.float().cpu() for t in (out, xg.grad, net.conv.weight.grad, net.bn.weight.grad, net.bn.bias.grad,
                                                net.bn.running_mean, net.bn.running_var)]
    names = ["out", "dx", "dw", "dgamma", "dbeta", "running_mean", "running_var"]
    refs = [out_ref.detach(), xr.grad, ref.conv.weight.grad, ref.bn.weight.grad, ref.bn.bias.grad, ref.bn.running_mean,
            ref.bn.running_var]
    for name, a, b in zip(names, res[True], res[False]):
        assert torch.equal(a, b), name                               # same kernels on the same values
    for name, got, want in zip(names, res[True], refs):
        err = ((got.double() - want).norm() / want.norm()).item()
        assert err <= (2e-2 if name in ("out", "dx", "dw", "dgamma", "dbeta") else 1e-3), (name, err)


@pytest.mark.parametrize("shape", [(2, 64, 64), (1, 70, 96), (2, 22, 130)])
def test_stem_weight_gradient_with_bn_backward_on_load(cuda, shape):
    """tsg_stem_conv_wrw_bn(img, da, xc, bp) == tsg_stem_conv_wrw(img, tsg_bn_bwd_apply(da, xc, bp)), bit for bit."""
    from torchseg_amd import kernels as K
    kp = K.provider()
    B, H, W = shape
    g = torch.Generator().manual_seed(H + 2 * W)
    img = torch.randn(B, 3, H, W, generator=g).to(cuda).bfloat16()
    w = (torch.randn(64, 3, 7, 7, generator=g) * 0.1).to(cuda)
    xc = kp.stem_conv_fwd(img, w)
    da = torch.randn(xc.shape, generator=g).to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    bp = torch.stack([torch.randn(64, generator=g) * 0.5 + 1.0, torch.randn(64, generator=g) * 0.3,
                      torch.randn(64, generator=g) * 0.1, torch.randn(64, generator=g) * 0.05,
                      torch.randn(64, generator=g) * 0.05]).to(cuda).contiguous()
    bp[0, ::9] *= -1.0
    layout, n, c, hw = K.bn_layout(xc)
    dy, _ = kp.bn_bwd_apply(da, xc, None, layout, n, c, hw, bp, True, False)
    assert torch.equal(kp.stem_conv_wrw_bn(img, da, xc, bp), kp.stem_conv_wrw(img, dy))


def test_spatial_path_chain_with_and_without_the_stem_node(cuda):
    """cbr_chain over [7x7/2 stem, 3x3/2, 3x3/2, 1x1] ConvBnRelu modules under the wrapper's bf16 autocast: the one-node
    stem -> BN -> conv form (TSG_STEM_BN_WRW) equals the module-by-module form in the output (bit for bit) and in every
    gradient (the kernel-level test above pins the fused weight gradient bit for bit)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "torchseg_amd", "furnace"))
    from seg_opr.seg_oprs import ConvBnRelu, cbr_chain
    from torchseg_amd import convwrw
    from torchseg_amd.convwrw import install_conv_wrw
    from torchseg_amd.stemconv import install_stem_conv
    from torchseg_amd.syncbn import SyncBatchNorm
    torch.manual_seed(5)
    proto = nn.ModuleList([ConvBnRelu(3, 64, 7, 2, 3, norm_layer=SyncBatchNorm), ConvBnRelu(64, 64, 3, 2, 1, norm_layer=SyncBatchNorm),
                           ConvBnRelu(64, 64, 3, 2, 1, norm_layer=SyncBatchNorm),
                           ConvBnRelu(64, 128, 1, 1, 0, norm_layer=SyncBatchNorm)])
    x = torch.randn(2, 3, 96, 128).to(cuda)
    out = {}
    for flag in (True, False):
        import copy
        mods = copy.deepcopy(proto).to(cuda).to(memory_format=torch.channels_last)
        assert install_stem_conv(mods) == 1 and install_conv_wrw(mods) == 2
        calls = []
        kp = convwrw.K.provider()
        orig = kp.stem_conv_wrw_bn
        kp.stem_conv_wrw_bn = lambda *a, **k: (calls.append("wrw_bn"), orig(*a, **k))[1]
        old = convwrw._STEM_BN_WRW
        convwrw._STEM_BN_WRW = flag
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = cbr_chain(list(mods), x)
            y.float().square().mean().backward()
        finally:
            convwrw._STEM_BN_WRW = old
            del kp.stem_conv_wrw_bn
        assert calls == (["wrw_bn"] if flag else [])
        out[flag] = [y.detach().float()] + [p.grad.float() for p in mods.parameters()] + \
                    [b.clone() for n_, b in mods.named_buffers() if "running" in n_]
    assert torch.equal(out[True][0], out[False][0])                 # forward: the same kernels on the same values
    for a, b in zip(out[True][1:], out[False][1:]):                 # backward passes through the library's 1x1 kernels,
        assert ((a - b).norm() / b.norm().clamp_min(1e-12)).item() <= 1e-3   # which are not run-to-run deterministic
