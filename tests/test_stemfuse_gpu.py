"""GPU parity of the stem fusions (through the C-ABI):
 * tsg_stem_conv_fwd_stats: the activation of tsg_stem_conv_fwd, bit for bit, plus per-block sums that fold to the sums of
   exactly that activation (what tsg_bn_stats would have read it again for);
 * tsg_bn_relu_pool_fwd: values bit-equal to tsg_maxpool_nhwc_fwd(tsg_bn_apply_fwd(x, relu)), argmax bytes equal in fp32
   and equal up to rounding ties in bf16;
 * tsg_bn_relu_pool_bwd_reduce / _apply: the BN backward of the gradient tsg_maxpool_nhwc_bwd would have written, without
   writing it (fp32: to rounding of the sums; bf16: the unfused path additionally rounds that gradient to bf16);
 * the ResNet stem as a module: fused == unfused, and both == nn.BatchNorm2d + ReLU + MaxPool2d in fp64 on the CPU."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _fold(partial, S):
    return partial[:S].double().sum(0)          # [2, C]


@pytest.mark.parametrize("shape", [(2, 64, 64), (1, 70, 96), (3, 22, 130), (2, 256, 256)])
def test_stem_conv_stats_epilogue(cuda, shape):
    from torchseg_amd import kernels as K
    kp = K.provider()
    B, H, W = shape
    g = torch.Generator().manual_seed(H + W)
    x = torch.randn(B, 3, H, W, generator=g).to(cuda).bfloat16()
    w = (torch.randn(64, 3, 7, 7, generator=g) * 0.1).to(cuda)
    y0 = kp.stem_conv_fwd(x, w)
    y1, partial = kp.stem_conv_fwd_stats(x, w)
    assert torch.equal(y0, y1)
    sums = _fold(partial, partial.shape[0]).cpu()
    yf = y1.double().cpu()
    ref = torch.stack([yf.sum((0, 2, 3)), (yf * yf).sum((0, 2, 3))])
    np.testing.assert_allclose(sums.numpy(), ref.numpy(), rtol=2e-5, atol=1e-3)
    # and they are what tsg_bn_finalize makes of tsg_bn_stats on the same tensor
    lay = K.bn_layout(y1)
    p2, S2 = kp.bn_stats(y1, *lay)
    np.testing.assert_allclose(sums.numpy(), _fold(p2, S2).cpu().numpy(), rtol=2e-5, atol=1e-3)


def _pack(C, cuda, seed):
    g = torch.Generator().manual_seed(seed)
    a = (torch.randn(C, generator=g) * 0.5 + 1.0)
    a[::5] *= -1.0                                  # negative gamma: the max is not monotone in x
    b = torch.randn(C, generator=g) * 0.3
    mean = torch.randn(C, generator=g) * 0.1
    return torch.stack([a, b, mean]).to(cuda).contiguous()


CASES = [(2, 16, 9, 11), (1, 64, 32, 32), (2, 8, 7, 30), (1, 128, 33, 18), (2, 64, 64, 48)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("case", CASES)
def test_bn_relu_pool_forward_equals_the_unfused_kernels(cuda, case, dtype):
    from torchseg_amd import kernels as K
    kp = K.provider()
    N, C, IH, IW = case
    g = torch.Generator().manual_seed(C + IH)
    x = torch.randn(N, C, IH, IW, generator=g).to(cuda).to(dtype).contiguous(memory_format=torch.channels_last)
    fp = _pack(C, cuda, 1)
    layout, n, c, hw = K.bn_layout(x)
    y_bn = kp.bn_apply_fwd(x, None, layout, n, c, hw, fp, True)
    y_ref, idx_ref = kp.maxpool_fwd(y_bn, 3, 2, 1)
    y, idx = kp.bn_relu_pool_fwd(x, fp)
    assert y.shape == y_ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(y, y_ref)
    if dtype == torch.float32:
        assert torch.equal(idx, idx_ref)
    else:
        # the fused kernel ranks the UNROUNDED values (like the fp32 reference); the unfused pair ranks bf16-rounded ones
        # and sees ties where two elements round alike.  Every argmax byte must name an element that holds the maximum.
        win = torch.nn.functional.unfold(torch.nn.functional.pad(y_bn.float(), (1, 1, 1, 1), value=float("-inf")),
                                         3, stride=2).view(N, C, 9, y.shape[2], y.shape[3])
        pick = win.gather(2, idx.permute(0, 3, 1, 2).long().unsqueeze(2)).squeeze(2)
        assert torch.equal(pick, y.float())
        assert (idx != idx_ref).float().mean().item() < 0.02


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("case", CASES)
def test_bn_relu_pool_backward_equals_the_unfused_kernels(cuda, case, dtype):
    from torchseg_amd import kernels as K
    kp = K.provider()
    N, C, IH, IW = case
    g = torch.Generator().manual_seed(C * 3 + IW)
    x = torch.randn(N, C, IH, IW, generator=g).to(cuda).to(dtype).contiguous(memory_format=torch.channels_last)
    fp = _pack(C, cuda, 2)
    y, idx = kp.bn_relu_pool_fwd(x, fp)
    dpool = torch.randn(y.shape, generator=g).to(cuda).to(dtype).contiguous(memory_format=torch.channels_last)
    layout, n, c, hw = K.bn_layout(x)
    # unfused: the pool gradient is written (rounded to the element type), then the BN backward reads it twice
    dy_full = kp.maxpool_bwd(dpool, idx, tuple(x.shape), 3, 2, 1)
    p_ref, S_ref = kp.bn_bwd_reduce(dy_full, x, None, layout, n, c, hw, fp, True)
    p, S = kp.bn_relu_pool_bwd_reduce(dpool, idx, x, fp)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    ref_s, got_s = _fold(p_ref, S_ref).cpu().numpy(), _fold(p, S).cpu().numpy()
    np.testing.assert_allclose(got_s, ref_s, rtol=tol, atol=tol * max(1.0, float(np.abs(ref_s).max())))
    invstd = torch.rand(C, generator=g).to(cuda) + 0.5
    _, _, bp = kp.bn_bwd_coeffs(p_ref, S_ref, C, float(n * hw), None, True, invstd, fp, True, True)
    dx_ref, _ = kp.bn_bwd_apply(dy_full, x, None, layout, n, c, hw, bp, True, False)
    dx = kp.bn_relu_pool_bwd_apply(dpool, idx, x, bp)
    assert dx.is_contiguous(memory_format=torch.channels_last)
    scale = dx_ref.float().abs().max().item()
    err = (dx.float() - dx_ref.float()).abs().max().item()
    assert err <= (1e-6 if dtype == torch.float32 else 2e-2) * scale, (err, scale)
    if dtype == torch.float32:
        # exact fp64 statement of the same thing on the CPU
        xd = x.double().cpu().requires_grad_(True)
        a, b = fp[0].double().cpu().view(1, -1, 1, 1), fp[1].double().cpu().view(1, -1, 1, 1)
        yd = torch.nn.functional.max_pool2d(torch.relu(xd * a + b), 3, 2, 1)
        (gx,) = torch.autograd.grad(yd, xd, dpool.double().cpu())
        dyd = gx / a                                    # d/dx = a * dy'  =>  dy' (zero where the ReLU is off)
        np.testing.assert_allclose(_fold(p, S)[0].cpu().numpy(), dyd.sum((0, 2, 3)).numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_resnet_stem_fused_equals_unfused_and_the_cpu(cuda, dtype):
    """conv1 -> bn1 -> relu -> maxpool of furnace/base_model/resnet.py through our modules, with and without the fusions
    (statistics in the convolution's epilogue, BN + ReLU + pool in one pass), against stock modules in fp64 on the CPU."""
    from torchseg_amd import syncbn, stemconv
    from torchseg_amd.pool import MaxPool2d
    from torchseg_amd.stemconv import install_stem_conv
    from torchseg_amd.syncbn import SyncBatchNorm, bn_relu_maxpool
    torch.manual_seed(0)

    class Stem(nn.Module):
        def __init__(self, bn_cls, pool_cls):
            super().__init__()
            self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
            self.bn1 = bn_cls(64)
            self.relu = nn.ReLU()
            self.maxpool = pool_cls(3, 2, 1)

        def forward(self, x, fused):
            x = self.conv1(x)
            if fused:
                y = bn_relu_maxpool(self.bn1, x, self.maxpool)
                assert y is not None
                return y
            return self.maxpool(self.relu(self.bn1(x)))

    ref = Stem(nn.BatchNorm2d, nn.MaxPool2d).double()
    with torch.no_grad():
        ref.bn1.weight.copy_(torch.randn(64) * 0.5 + 1.0)
        ref.bn1.bias.copy_(torch.randn(64) * 0.2)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 96, 80, generator=g)
    dout = torch.randn(2, 64, 24, 20, generator=g)
    xr = x.double()
    if dtype == torch.bfloat16:                         # what autocast feeds the convolution
        xr = x.bfloat16().double()
        with torch.no_grad():
            ref.conv1.weight.copy_(ref.conv1.weight.float().bfloat16().double())
    out_ref = ref(xr, False)
    out_ref.backward(dout.double())

    results = {}
    for fused in (True, False):
        net = Stem(SyncBatchNorm, MaxPool2d).to(cuda)
        net.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
        net.bn1.running_mean.zero_(); net.bn1.running_var.fill_(1.0); net.bn1.num_batches_tracked.zero_()
        if dtype == torch.bfloat16:
            assert install_stem_conv(net) == 1
        old = stemconv._STEM_STATS
        stemconv._STEM_STATS = fused
        calls = []
        kp = syncbn.K.provider()
        orig = kp.bn_stats
        kp.bn_stats = lambda *a, **k: (calls.append("bn_stats"), orig(*a, **k))[1]
        try:
            xin = x.to(cuda)
            if dtype == torch.bfloat16:
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    out = net(xin, fused)
            else:
                out = net(xin.contiguous(memory_format=torch.channels_last), fused)
            out.backward(dout.to(cuda).to(out.dtype))
        finally:
            stemconv._STEM_STATS = old
            del kp.bn_stats
        if dtype == torch.bfloat16:
            assert calls == ([] if fused else ["bn_stats"])     # the statistics rode along with the convolution
        results[fused] = (out.float().cpu(), net.conv1.weight.grad.float().cpu(), net.bn1.weight.grad.float().cpu(),
                          net.bn1.bias.grad.float().cpu(), net.bn1.running_mean.cpu(), net.bn1.running_var.cpu())
    names = ["out", "dw", "dgamma", "dbeta", "running_mean", "running_var"]
    refs = [out_ref.detach(), ref.conv1.weight.grad, ref.bn1.weight.grad, ref.bn1.bias.grad, ref.bn1.running_mean,
            ref.bn1.running_var]
    # against fp64: fp32 to 1e-3 of each tensor's scale (the MIOpen fp32 convolution sets that floor).  In bf16 the values
    # agree to bf16 rounding, but near-ties of the pooling windows pick another pixel than fp64 does, which re-routes
    # whole gradient entries: the gradients are compared in relative L2 (measured 0.05-0.1, the same for both paths).
    report = {}
    for fused in (True, False):
        for name, got, want in zip(names, results[fused], refs):
            if dtype == torch.bfloat16 and name in ("dw", "dgamma", "dbeta"):
                err, bound = ((got.double() - want).norm() / want.norm()).item(), 0.2
            else:
                err = (got.double() - want).abs().max().item()
                bound = (1e-3 if dtype == torch.float32 else 3e-2) * max(want.abs().max().item(), 1e-3)
            report[(fused, name)] = (err, bound)
    print(report)
    for key, (err, bound) in report.items():
        assert err <= bound, (key, err, bound, report)
    # the two HIP paths: identical activations; in fp32 identical routing, hence gradients equal to rounding; in bf16 the
    # fused forward ranks unrounded values, so a few rounding ties route differently (see the forward test)
    for name, a, b in zip(names, results[True], results[False]):
        if dtype == torch.float32 or name in ("out", "running_mean", "running_var"):
            scale = b.abs().max().item()
            assert (a - b).abs().max().item() <= 1e-5 * max(scale, 1e-3), name
        else:
            assert ((a - b).norm() / b.norm()).item() <= 0.1, name
