"""GPU parity: bilinear align_corners=True fwd/bwd (+fused add), nearest, and the
aten override that unchanged network.py call sites hit."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import upsample_ref as R

pytestmark = pytest.mark.gpu

SIZES = [(1, 1, 32, 32), (32, 32, 64, 64), (8, 8, 64, 64), (7, 5, 13, 9), (16, 12, 128, 96),
         (6, 6, 6, 6), (9, 9, 4, 3), (64, 64, 1024, 1024), (3, 4, 1, 1), (60, 60, 480, 480), (6, 6, 60, 60)]


@pytest.mark.parametrize("ih,iw,oh,ow", SIZES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bilinear_fwd_bwd(cuda, ih, iw, oh, ow, dtype):
    from torchseg_amd.upsample import upsample_bilinear_ac
    nc = (2, 3) if oh * ow < 500000 else (1, 2)
    g = torch.Generator().manual_seed(ih + ow)
    x = torch.randn(*nc, ih, iw, generator=g).to(dtype)
    dy = torch.randn(*nc, oh, ow, generator=g).to(dtype)
    xd = x.to(cuda).requires_grad_(True)
    y = upsample_bilinear_ac(xd, size=(oh, ow))
    y.backward(dy.to(cuda))
    y_ref = R.upsample_bilinear_ac(x.float().numpy(), oh, ow)
    dx_ref = R.upsample_bilinear_ac_backward(dy.float().numpy(), ih, iw)
    if dtype == torch.float32:
        np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(xd.grad.cpu().numpy(), dx_ref, rtol=1e-4, atol=1e-4 * max(1.0, oh / ih))
    else:
        np.testing.assert_allclose(y.detach().float().cpu().numpy(), y_ref, rtol=8e-3, atol=8e-3)
        np.testing.assert_allclose(xd.grad.float().cpu().numpy(), dx_ref, rtol=1e-2, atol=1e-2 * np.abs(dx_ref).max())


def test_bilinear_fused_add(cuda):
    from torchseg_amd.upsample import upsample_bilinear_ac
    x = torch.randn(2, 8, 16, 16, device=cuda, requires_grad=True)
    a = torch.randn(2, 8, 32, 32, device=cuda, requires_grad=True)
    y = upsample_bilinear_ac(x, scale_factor=2, add=a)
    ref = F.interpolate(x.detach().cpu(), scale_factor=2, mode="bilinear", align_corners=True) + a.detach().cpu()
    torch.testing.assert_close(y.detach().cpu(), ref, rtol=1e-5, atol=1e-5)
    g = torch.randn_like(y)
    y.backward(g)
    torch.testing.assert_close(a.grad, g)


def test_adjointness_full_size(cuda):
    """config-2 head size 19 x 128^2 -> 1024^2: <up(x), g> == <x, up^T(g)>."""
    from torchseg_amd import kernels as K
    kp = K.provider()
    x = torch.randn(4, 19, 128, 128, device=cuda)
    gq = torch.randn(4, 19, 1024, 1024, device=cuda)
    y = kp.upsample_fwd(x, None, 1024, 1024)
    dx = kp.upsample_bwd(gq, 128, 128)
    lhs = (y.double() * gq.double()).sum().item()
    rhs = (x.double() * dx.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0) + 1e-2


def test_nearest(cuda):
    from torchseg_amd import kernels as K
    x = torch.randint(0, 255, (2, 3, 40, 60), dtype=torch.uint8)
    for oh, ow in [(20, 30), (80, 90), (5, 7)]:
        y = K.provider().upsample_nearest(x.to(cuda), oh, ow).cpu()
        np.testing.assert_array_equal(y.numpy(), R.upsample_nearest(x.numpy(), oh, ow))
    xl = torch.randint(0, 19, (2, 96, 192), dtype=torch.int64)
    y = K.provider().upsample_nearest(xl.to(cuda), 12, 24).cpu()
    np.testing.assert_array_equal(y.numpy(), R.upsample_nearest(xl.numpy(), 12, 24))


def test_aten_override_used_by_interpolate(cuda):
    """Unchanged reference code calls F.interpolate directly: after install the
    HIP kernel must be what runs (checked by counting provider calls) and
    autograd must flow through it."""
    from torchseg_amd import kernels as K
    from torchseg_amd.upsample import install_aten_overrides
    install_aten_overrides()
    kp = K.provider()
    calls = {"f": 0, "b": 0}
    of, ob = kp.upsample_fwd, kp.upsample_bwd
    kp.upsample_fwd = lambda *a: (calls.__setitem__("f", calls["f"] + 1), of(*a))[1]
    kp.upsample_bwd = lambda *a: (calls.__setitem__("b", calls["b"] + 1), ob(*a))[1]
    try:
        x = torch.randn(2, 5, 9, 7, device=cuda, requires_grad=True)
        y = F.interpolate(x, scale_factor=8, mode="bilinear", align_corners=True)
        y.sum().backward()
    finally:
        kp.upsample_fwd, kp.upsample_bwd = of, ob
    assert calls == {"f": 1, "b": 1}
    xr = x.detach().cpu().requires_grad_(True)
    yr = F.interpolate(xr, scale_factor=8, mode="bilinear", align_corners=True)
    yr.sum().backward()
    torch.testing.assert_close(y.detach().cpu(), yr.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(x.grad.cpu(), xr.grad, rtol=1e-4, atol=1e-4)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        yb = F.interpolate(x.bfloat16(), size=(20, 20), mode="bilinear", align_corners=True)
    assert yb.dtype == torch.bfloat16


@pytest.mark.parametrize("c,ih,iw,oh,ow", [(128, 1, 1, 32, 32), (128, 32, 32, 64, 64), (128, 64, 64, 128, 128),
                                           (8, 7, 5, 13, 9), (16, 9, 9, 4, 3), (256, 6, 6, 60, 60),
                                           # PSPNet's pyramid pooling at 720^2 (pspnet network.py:101-106): small sources,
                                           # footprints of up to 90 rows -> the row-split backward (up_bwd_nhwc_split)
                                           (512, 2, 2, 90, 90), (512, 3, 3, 90, 90), (512, 6, 6, 90, 90), (24, 2, 3, 40, 17)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bilinear_channels_last(cuda, c, ih, iw, oh, ow, dtype):
    from torchseg_amd.upsample import upsample_bilinear_ac
    g = torch.Generator().manual_seed(c + oh)
    x = torch.randn(2, c, ih, iw, generator=g).to(dtype)
    dy = torch.randn(2, c, oh, ow, generator=g).to(dtype)
    xd = x.to(cuda).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = upsample_bilinear_ac(xd, size=(oh, ow))
    if ih * iw > 1:
        assert y.is_contiguous(memory_format=torch.channels_last)
    y.backward(dy.to(cuda).contiguous(memory_format=torch.channels_last))
    y_ref = R.upsample_bilinear_ac(x.float().numpy(), oh, ow)
    dx_ref = R.upsample_bilinear_ac_backward(dy.float().numpy(), ih, iw)
    if dtype == torch.float32:
        np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(xd.grad.cpu().numpy(), dx_ref, rtol=1e-4, atol=1e-4 * max(1.0, oh / ih))
    else:
        np.testing.assert_allclose(y.detach().float().cpu().numpy(), y_ref, rtol=8e-3, atol=8e-3)
        np.testing.assert_allclose(xd.grad.float().cpu().numpy(), dx_ref, rtol=1e-2, atol=1e-2 * np.abs(dx_ref).max())


def test_backward_of_a_channel_slice_stays_channels_last(cuda):
    """torch.cat's backward hands the pyramid-pooling branches channel SLICES of a channels_last gradient (pixel stride
    = all channels): they must take the NHWC gather (with its small-source path), not a copy to NCHW."""
    from torchseg_amd.upsample import upsample_bilinear_ac
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 64, 3, 3, generator=g)
    big = torch.randn(2, 160, 45, 45, generator=g)
    xd = x.to(cuda).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = upsample_bilinear_ac(xd, size=(45, 45))
    dyd = big.to(cuda).contiguous(memory_format=torch.channels_last)[:, 32:96]
    assert dyd.stride(1) == 1 and not dyd.is_contiguous(memory_format=torch.channels_last)
    y.backward(dyd)
    dx_ref = R.upsample_bilinear_ac_backward(big[:, 32:96].numpy(), 3, 3)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), dx_ref, rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("channels_last", [False, True])
@pytest.mark.parametrize("shape", [(2, 128, 32, 32, 64, 64), (2, 128, 64, 64, 128, 128), (1, 16, 5, 7, 13, 9)])
def test_presum_upsample_matches_add_then_interpolate(cuda, dtype, channels_last, shape):
    """SURVEY 8 row a8 as the reference orders it (bisenet network.py:91-95): `fm += last_fm` then F.interpolate.
    Oracle: round_T(a + b) -> numpy bilinear; both addends get the transposed operator applied to dy."""
    from torchseg_amd.fusion import upsample_presum
    n, c, ih, iw, oh, ow = shape
    g = torch.Generator().manual_seed(ih * ow)
    a = torch.randn(n, c, ih, iw, generator=g).to(dtype)
    b = torch.randn(n, c, ih, iw, generator=g).to(dtype)
    dy = torch.randn(n, c, oh, ow, generator=g).to(dtype)
    fmt = torch.channels_last if channels_last else torch.contiguous_format
    ad = a.to(cuda).contiguous(memory_format=fmt).requires_grad_(True)
    bd = b.to(cuda).contiguous(memory_format=fmt).requires_grad_(True)
    y = upsample_presum(ad, bd, size=(oh, ow))
    assert y.is_contiguous(memory_format=fmt)
    y.backward(dy.to(cuda).contiguous(memory_format=fmt))
    s = (a.float() + b.float()).to(dtype).float().numpy()          # what the eager in-place add stores
    y_ref = R.upsample_bilinear_ac(s, oh, ow)
    d_ref = R.upsample_bilinear_ac_backward(dy.float().numpy(), ih, iw)
    if dtype == torch.float32:
        np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(ad.grad.cpu().numpy(), d_ref, rtol=1e-4, atol=1e-4 * max(1.0, oh / ih))
    else:
        np.testing.assert_allclose(y.detach().float().cpu().numpy(), y_ref, rtol=8e-3, atol=8e-3)
        np.testing.assert_allclose(ad.grad.float().cpu().numpy(), d_ref, rtol=1e-2, atol=1e-2 * np.abs(d_ref).max())
    assert torch.equal(ad.grad, bd.grad)


def test_iadd_interpolate_pattern_is_fused_under_the_ddp_wrapper(cuda):
    """An unchanged network.py's `fm += last_fm; F.interpolate(fm, ...)`: no aten add kernel, one presum launch,
    numbers equal to eager on the CPU."""
    import torch.nn as nn
    from torchseg_amd import kernels as K
    from torchseg_amd.ddp import DistributedDataParallel

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.c1 = nn.Conv2d(8, 8, 1)
            self.c2 = nn.Conv2d(8, 8, 1)

        def forward(self, x):
            fm = self.c1(x)
            last_fm = self.c2(x)
            fm += last_fm
            return F.interpolate(fm, size=(24, 24), mode='bilinear', align_corners=True).square().mean()

    torch.manual_seed(2)
    ref = Net()
    net = Net()
    net.load_state_dict(ref.state_dict())
    net = DistributedDataParallel(net.to(cuda), compute_dtype=torch.float32, channels_last=False)
    assert net.fuse_add_up
    x = torch.randn(2, 8, 12, 12)
    lr = ref(x)
    lr.backward()
    kp = K.provider()
    calls = []
    orig = kp.upsample_presum_fwd
    kp.upsample_presum_fwd = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        l = net(x.to(cuda))
        l.backward()
    finally:
        del kp.upsample_presum_fwd
    assert calls == [1]
    assert abs(l.item() - lr.item()) <= 1e-5 * max(1.0, abs(lr.item()))
    for (n, p), (_, q) in zip(net.module.named_parameters(), ref.named_parameters()):
        torch.testing.assert_close(p.grad.cpu(), q.grad, rtol=1e-4, atol=1e-6, msg=n)


def test_override_leaves_uncovered_cases_to_aten(cuda):
    """The aten override is process-wide: bilinear resizes our kernels do not cover (align_corners=False, fp16 / fp64) must
    keep working for unrelated code in the process, forward and backward."""
    import torch.nn.functional as F
    from torchseg_amd.upsample import install_aten_overrides
    install_aten_overrides()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 3, 5, 7, generator=g)
    for dtype, ac, tol in ((torch.float32, False, 1e-5), (torch.float64, True, 1e-12), (torch.float16, True, 2e-2)):
        xr = x.clone().to(dtype).requires_grad_(True)          # clone: .to() of the same dtype returns x itself
        ref = F.interpolate(xr, size=(11, 13), mode="bilinear", align_corners=ac)
        ref.sum().backward()
        xg = x.detach().to(cuda).to(dtype).requires_grad_(True)
        out = F.interpolate(xg, size=(11, 13), mode="bilinear", align_corners=ac)
        (out * 1.0).sum().backward()
        torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=tol, atol=tol)
        torch.testing.assert_close(xg.grad.cpu(), xr.grad, rtol=tol, atol=tol)
