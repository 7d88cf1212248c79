"""GPU parity of tsg_conv3x3_gen_fwd (csrc/conv3g.hip, through the C-ABI) with oracle/conv_ref.py on the same bf16-rounded
operands (fp64 accumulation): y is bf16 -> one bf16 ulp of the fp64 result (2^-8 relative) plus 1e-3 of the output scale
for cancellation.  Covers both oc-tile widths (C_out = 64 k: 64-wide; 128 k: 128-wide), several oc tiles, C_in from one
chunk to many, partial pixel tiles (H, W not multiples of 8 / 32), the statistics epilogue, normalise-on-load, the
mode-1 filter (= the data gradient of the same convolution) and the autograd node the DDP wrapper installs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import conv_ref

pytestmark = pytest.mark.gpu

# (B, Cin, Cout, H, W)
SHAPES = [(2, 128, 128, 8, 32), (1, 16, 64, 5, 37), (2, 64, 128, 19, 70), (1, 128, 256, 16, 64), (3, 256, 64, 9, 33),
          (1, 512, 512, 8, 32), (1, 32, 192, 1, 1), (2, 128, 128, 24, 96)]


def _operands(cuda, B, Cin, Cout, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    xb = x.to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    wd = w.to(cuda).contiguous(memory_format=torch.channels_last)          # fp32 master, channels_last
    return x, w, xb, wd


def _check(y, y_ref):
    err = (y.double().cpu() - y_ref).abs()
    bound = y_ref.abs() * 2.0 ** -8 + 1e-3 * y_ref.abs().max()
    assert bool((err <= bound).all()), (err.max().item(), y_ref.abs().max().item())


@pytest.fixture(params=["64", "128"])
def tile_width(request, monkeypatch):
    """Both oc-tile widths at every shape (the library picks 128 only for launches with >= 192 work items; C_out that is
    not a multiple of 128 takes 64 either way)."""
    from torchseg_amd import kernels as K
    monkeypatch.setenv("TSG_CONV3G_BN", request.param)
    K.provider()._npart.clear()                            # cached geometry answers depend on the override
    yield request.param
    K.provider()._npart.clear()


@pytest.mark.parametrize("shape", SHAPES)
def test_forward_vs_oracle(cuda, shape, tile_width):
    from torchseg_amd import kernels as K
    kp = K.provider()
    B, Cin, Cout, H, W = shape
    x, w, xb, wd = _operands(cuda, *shape, seed=sum(shape))
    assert kp.conv3x3_gen_supported(xb, wd, 1, 1, 1, 1)
    wf = kp.conv3x3_gen_prep_filter(wd, 0, xb)
    y = kp.conv3x3_gen_fwd(xb, wf, Cout)
    assert tuple(y.shape) == (B, Cout, H, W) and y.is_contiguous(memory_format=torch.channels_last)
    _check(y, conv_ref.conv2d_ref(conv_ref.bf16_round(x), conv_ref.bf16_round(w), stride=1, pad=1))
    # a bf16 master gives the same prepared filter as the fp32 master rounded by the kernel
    wf2 = kp.conv3x3_gen_prep_filter(wd.bfloat16().contiguous(memory_format=torch.channels_last), 0, xb)
    assert torch.equal(wf[0], wf2[0]) and wf[1] == wf2[1]


@pytest.mark.parametrize("shape", [(2, 128, 128, 70, 96), (1, 64, 256, 33, 40), (2, 256, 64, 16, 64)])
def test_statistics_epilogue_and_determinism(cuda, shape, tile_width):
    from torchseg_amd import kernels as K
    kp = K.provider()
    B, Cin, Cout, H, W = shape
    x, w, xb, wd = _operands(cuda, *shape, seed=5)
    wf = kp.conv3x3_gen_prep_filter(wd, 0, xb)
    y, partial = kp.conv3x3_gen_fwd(xb, wf, Cout, with_stats=True)
    assert torch.equal(y, kp.conv3x3_gen_fwd(xb, wf, Cout))
    assert partial.shape[1:] == (2, Cout)
    sums = partial.double().sum(0).cpu()
    yf = y.double().cpu()
    ref = torch.stack([yf.sum((0, 2, 3)), (yf * yf).sum((0, 2, 3))])
    np.testing.assert_allclose(sums.numpy(), ref.numpy(), rtol=2e-5, atol=2e-3)
    y2, p2 = kp.conv3x3_gen_fwd(xb, wf, Cout, with_stats=True)
    assert torch.equal(y, y2) and torch.equal(partial, p2)
    # ... and they are what tsg_bn_stats computes from y (the pass the epilogue replaces), to fp32 summation order
    layout, N, C, HW = K.bn_layout(y)
    p_ref, S = kp.bn_stats(y, layout, N, C, HW)
    np.testing.assert_allclose(sums.numpy(), p_ref[:S].double().sum(0).cpu().numpy(), rtol=2e-5, atol=2e-3)


@pytest.mark.parametrize("shape", [(2, 128, 64, 12, 40), (1, 64, 128, 20, 33), (1, 512, 128, 8, 32)])
def test_normalise_on_load_equals_bn_apply_then_conv(cuda, shape, tile_width):
    """in_ab: the convolution reads relu(a x + b); bit-equal to tsg_bn_apply_fwd's output fed to the same kernel."""
    from torchseg_amd import kernels as K
    kp = K.provider()
    B, Cin, Cout, H, W = shape
    x, w, xb, wd = _operands(cuda, *shape, seed=8)
    g = torch.Generator().manual_seed(1)
    a = (torch.rand(Cin, generator=g) + 0.5).to(cuda)
    b = (torch.randn(Cin, generator=g) * 0.3).to(cuda)
    fp = torch.stack([a, b, torch.zeros_like(a)]).contiguous()
    wf = kp.conv3x3_gen_prep_filter(wd, 0, xb)
    layout, N, C, HW = K.bn_layout(xb)
    xn = kp.bn_apply_fwd(xb, None, layout, N, C, HW, fp, True)
    want = kp.conv3x3_gen_fwd(xn, wf, Cout)
    got = kp.conv3x3_gen_fwd(xb, wf, Cout, in_ab=fp)
    assert torch.equal(got, want)
    ref_in = torch.relu(conv_ref.bf16_round(x) * a.cpu().view(1, -1, 1, 1) + b.cpu().view(1, -1, 1, 1))
    _check(got, conv_ref.conv2d_ref(conv_ref.bf16_round(ref_in), conv_ref.bf16_round(w), stride=1, pad=1))


@pytest.mark.parametrize("shape", [(2, 128, 128, 24, 40), (1, 64, 256, 9, 33), (2, 256, 64, 8, 32)])
def test_mode1_filter_is_the_data_gradient(cuda, shape, tile_width):
    from torchseg_amd import kernels as K
    kp = K.provider()
    B, Cin, Cout, H, W = shape
    x, w, xb, wd = _operands(cuda, *shape, seed=11)
    g = torch.Generator().manual_seed(2)
    dy = torch.randn(B, Cout, H, W, generator=g)
    dyb = dy.to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    dx = kp.conv3x3_gen_fwd(dyb, kp.conv3x3_gen_prep_filter(wd, 1, dyb), Cin)
    xr = conv_ref.bf16_round(x).double().requires_grad_(True)
    yr = F.conv2d(xr, conv_ref.bf16_round(w).double(), None, 1, 1)
    yr.backward(conv_ref.bf16_round(dy).double())
    _check(dx, xr.grad)


def test_autograd_node_installed_by_the_wrapper(cuda):
    """nn.Conv2d(128, 128, 3, padding=1) re-classed by install_conv_wrw: forward, dx, dw against torch in fp64 on the
    bf16-rounded operands; the output carries the statistics partial for the SyncBatchNorm behind it."""
    import torch.nn as nn
    from torchseg_amd.convwrw import WrwConv2d, install_conv_wrw
    from torchseg_amd.stemconv import take_bn_partial
    torch.manual_seed(3)
    conv = nn.Conv2d(128, 256, 3, padding=1, bias=False).to(cuda)
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    assert install_conv_wrw(conv) == 1 and isinstance(conv, WrwConv2d)
    conv.train()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 128, 16, 40, generator=g)
    dy = torch.randn(2, 256, 16, 40, generator=g)
    xd = x.to(cuda).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = conv(xd)
    part = take_bn_partial(y)
    assert part is not None and part.shape[1:] == (2, 256)
    y.backward(dy.to(cuda).bfloat16().contiguous(memory_format=torch.channels_last))
    xr = conv_ref.bf16_round(x).double().requires_grad_(True)
    wr = conv_ref.bf16_round(conv.weight.detach().cpu().float()).double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, 1, 1)
    yr.backward(conv_ref.bf16_round(dy).double())
    _check(y.detach(), yr.detach())
    _check(xd.grad, xr.grad)
    dw = conv.weight.grad.double().cpu()
    assert (dw - wr.grad).abs().max().item() <= 2e-3 * wr.grad.abs().max().item()


@pytest.mark.parametrize("planes", [64, 128])
def test_skip_connection_gradient_joins_the_data_gradient_in_the_epilogue(cuda, planes):
    """BasicBlock whose skip connection is its input (resnet.py:36-53): with conv_with_skip the gradient of the skip path
    is the `addend` of conv1's data-gradient kernel (conv64 for 64 channels, the general kernel otherwise) instead of a
    separate add over three tensors.  bf16(bf16(conv) + addend) is what the eager add computes: every gradient must be
    bit-equal to the unfused block's, and the add kernel must be gone."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "torchseg_amd", "furnace"))
    from base_model.resnet import BasicBlock
    from torchseg_amd import convwrw
    from torchseg_amd.convwrw import install_conv_wrw
    from torchseg_amd.syncbn import SyncBatchNorm
    torch.manual_seed(7)
    proto = BasicBlock(planes, planes, 1, norm_layer=SyncBatchNorm)
    g = torch.Generator().manual_seed(8)
    x0 = torch.randn(2, planes, 24, 40, generator=g)
    dy0 = torch.randn(2, planes, 24, 40, generator=g)
    res = {}
    for fused in (True, False):
        import copy
        blk = copy.deepcopy(proto).to(cuda).to(memory_format=torch.channels_last)
        assert install_conv_wrw(blk) == 2
        blk.train()
        calls = []
        kp = convwrw.K.provider()
        name = "conv3x3_c64_fwd" if planes == 64 else "conv3x3_gen_fwd"
        orig = getattr(kp, name)

        def spy(*a, **k):
            calls.append(k.get("addend") is not None)
            return orig(*a, **k)
        setattr(kp, name, spy)
        old = convwrw._FUSE_SKIP
        convwrw._FUSE_SKIP = fused
        try:
            xin = x0.to(cuda).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
            x = xin * 1.0                                   # a non-leaf block input, as inside the network
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = blk(x)
            out.backward(dy0.to(cuda).to(out.dtype).contiguous(memory_format=torch.channels_last))
        finally:
            convwrw._FUSE_SKIP = old
            delattr(kp, name)
        assert any(calls) == fused, calls                   # exactly the fused run passes an addend
        res[fused] = [out.detach().float().cpu(), xin.grad.float().cpu()] + [p.grad.float().cpu() for p in blk.parameters()]
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)


# ---------------------------------------------------------------------------------------------------------------------
# The five geometries bench.py times, at its batch: 16 x {128 @ 128^2 (layer2, refine), 256 @ 64^2
# (layer3), 512 @ 32^2 (layer4), 128 -> 256 @ 128^2 (head conv_3x3), 256 -> 64 @ 128^2}: furnace/base_model/resnet.py:24-29,
# bisenet network.py:104-106,140-156.  These run the persistent tile loop (2048 / 512 / 128 pixel tiles over 512 / 256
# blocks), the XCD block map and the LDS-DMA filter path at the tile counts of the step.  Every launch form of the step:
# forward with the statistics epilogue, the mode-1 filter (= data gradient) without and with `addend` — each against
# oracle/conv_ref.py (fp64) on two images from the middle of the batch (bf16-ulp bound) and against the vendor library's
# convolution on the whole batch.
BENCH_GEOMS = [(16, 128, 128, 128), (16, 256, 256, 64), (16, 512, 512, 32), (16, 128, 256, 128), (16, 256, 64, 128)]


def _vs_library(y, y_lib):
    """Both are bf16 roundings of fp32 accumulations of the same products in different orders."""
    a, b = y.float(), y_lib.float()
    bound = b.abs() * 2.0 ** -6 + 2e-3 * b.abs().max()
    assert bool(((a - b).abs() <= bound).all()), ((a - b).abs().max().item(), b.abs().max().item())


@pytest.mark.parametrize("geom", BENCH_GEOMS)
def test_bench_geometry_forward_stats_dgrad_addend(cuda, geom):
    from torchseg_amd import kernels as K
    kp = K.provider()
    B, Cin, Cout, S = geom
    # which kernel the step runs at this geometry: the 16-row / LDS-DMA kernel wherever its tiles fill 2 x 256 block slots
    assert kp.conv3x3_gen_variant(B, S, S, Cin, Cout) == (0 if S == 32 else 1)
    assert kp.conv3x3_gen_variant(B, S, S, Cout, Cin) == (0 if S == 32 else 1)      # the data gradient
    assert kp.conv3x3_gen_variant(B, S, S, Cin, Cout, with_in_ab=True) == 0
    g = torch.Generator(device=cuda).manual_seed(Cin + Cout + S)
    x = torch.randn(B, Cin, S, S, generator=g, device=cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g, device=cuda) * (2.0 / (9 * Cin)) ** 0.5).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, Cout, S, S, generator=g, device=cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    skip = torch.randn(B, Cin, S, S, generator=g, device=cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    wb = w.bfloat16()
    sl = slice(7, 9)                                        # two images from the middle of the persistent loop
    wr = conv_ref.bf16_round(w.cpu())
    # ---- forward + statistics epilogue
    wf = kp.conv3x3_gen_prep_filter(w, 0, x)
    y, partial = kp.conv3x3_gen_fwd(x, wf, Cout, with_stats=True)
    assert torch.equal(y, kp.conv3x3_gen_fwd(x, wf, Cout))                        # the epilogue changes nothing; run-to-run equal
    _check(y[sl], conv_ref.conv2d_ref(x[sl].double().cpu(), wr, stride=1, pad=1))
    _vs_library(y, F.conv2d(x, wb, None, 1, 1))
    yf = y.double()
    ref_stats = torch.stack([yf.sum((0, 2, 3)), (yf * yf).sum((0, 2, 3))]).cpu()
    np.testing.assert_allclose(partial.double().sum(0).cpu().numpy(), ref_stats.numpy(), rtol=2e-5, atol=2e-2)
    # ---- data gradient (mode-1 filter), without and with the skip-connection addend
    wf1 = kp.conv3x3_gen_prep_filter(w, 1, dy)
    dx = kp.conv3x3_gen_fwd(dy, wf1, Cin)
    xr = torch.zeros(2, Cin, S, S, dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, wr, None, 1, 1).backward(dy[sl].double().cpu())
    _check(dx[sl], xr.grad)
    lib = torch.ops.aten.convolution_backward(dy, x, wb, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                              [True, False, False])[0]
    _vs_library(dx, lib)
    dxa = kp.conv3x3_gen_fwd(dy, wf1, Cin, addend=skip)
    assert torch.equal(dxa, dx + skip)                      # bf16(bf16(conv) + addend): what the eager add computes


# ---------------------------------------------------------------------------------------------------------------------
# conv3h_fwd_k (16-row tiles, both operands by LDS-DMA, zero padding = out-of-range buffer offsets) forced onto small and
# ragged problems: TSG_CONV3G_V2=2.  Partial tiles in both directions, one tile column / several, fewer tiles than block
# slots (idle blocks must write zero partial rows), one oc tile / several, two chunks / many.
V2_SHAPES = [(2, 128, 128, 8, 32), (2, 64, 128, 19, 70), (1, 128, 256, 16, 64), (3, 256, 64, 9, 33), (1, 512, 512, 8, 32),
             (1, 32, 192, 1, 1), (2, 128, 128, 24, 96), (1, 32, 64, 37, 45), (2, 96, 64, 33, 31), (1, 192, 64, 17, 65)]


@pytest.fixture
def sixteen_row_kernel(monkeypatch):
    from torchseg_amd import kernels as K
    monkeypatch.setenv("TSG_CONV3G_BN", "64")
    monkeypatch.setenv("TSG_CONV3G_V2", "2")
    K.provider()._npart.clear()
    yield
    K.provider()._npart.clear()


@pytest.mark.parametrize("shape", V2_SHAPES)
def test_sixteen_row_kernel_forward_stats_dgrad_addend_vs_oracle(cuda, shape, sixteen_row_kernel):
    from torchseg_amd import kernels as K
    kp = K.provider()
    B, Cin, Cout, H, W = shape
    assert kp.conv3x3_gen_variant(B, H, W, Cin, Cout) == 1
    x, w, xb, wd = _operands(cuda, *shape, seed=sum(shape) + 1)
    wf = kp.conv3x3_gen_prep_filter(wd, 0, xb)
    y, partial = kp.conv3x3_gen_fwd(xb, wf, Cout, with_stats=True)
    y_ref = conv_ref.conv2d_ref(conv_ref.bf16_round(x), conv_ref.bf16_round(w), stride=1, pad=1)
    _check(y, y_ref)
    y2, p2 = kp.conv3x3_gen_fwd(xb, wf, Cout, with_stats=True)
    assert torch.equal(y, y2) and torch.equal(partial, p2) and torch.equal(y, kp.conv3x3_gen_fwd(xb, wf, Cout))
    yf = y.double().cpu()
    ref = torch.stack([yf.sum((0, 2, 3)), (yf * yf).sum((0, 2, 3))])
    np.testing.assert_allclose(partial.double().sum(0).cpu().numpy(), ref.numpy(), rtol=2e-5, atol=2e-3)
    # data gradient = the same kernel on dy with the mode-1 filter; with the skip connection's gradient as addend
    if Cin % 64:
        return                                              # (its output channels are this convolution's inputs)
    assert kp.conv3x3_gen_variant(B, H, W, Cout, Cin) == 1
    g = torch.Generator().manual_seed(3)
    dy = torch.randn(B, Cout, H, W, generator=g)
    dyb = dy.to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    wf1 = kp.conv3x3_gen_prep_filter(wd, 1, dyb)
    dx = kp.conv3x3_gen_fwd(dyb, wf1, Cin)
    xr = conv_ref.bf16_round(x).double().requires_grad_(True)
    F.conv2d(xr, conv_ref.bf16_round(w).double(), None, 1, 1).backward(conv_ref.bf16_round(dy).double())
    _check(dx, xr.grad)
    skip = torch.randn(B, Cin, H, W, generator=g).to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    assert torch.equal(kp.conv3x3_gen_fwd(dyb, wf1, Cin, addend=skip), dx + skip)


def test_sixteen_row_kernel_against_the_eight_row_kernel(cuda, monkeypatch):
    """The two kernels walk the nine taps of a chunk in different orders (kernel-row-major / column-shift-major), so their
    fp32 accumulations differ in the last bits: y equal to a bf16 ulp at a few rounding boundaries."""
    from torchseg_amd import kernels as K
    kp = K.provider()
    shape = (4, 128, 128, 40, 70)
    x, w, xb, wd = _operands(cuda, *shape, seed=21)
    monkeypatch.setenv("TSG_CONV3G_BN", "64")
    out = {}
    for v2 in ("0", "2"):
        monkeypatch.setenv("TSG_CONV3G_V2", v2)
        kp._npart.clear()
        assert kp.conv3x3_gen_variant(4, 40, 70, 128, 128) == (1 if v2 == "2" else 0)
        out[v2] = kp.conv3x3_gen_fwd(xb, kp.conv3x3_gen_prep_filter(wd, 0, xb), 128, with_stats=True)
    kp._npart.clear()
    _vs_library(out["2"][0], out["0"][0])
    assert (out["2"][0] != out["0"][0]).float().mean().item() < 0.05     # a rounding boundary now and then, not a different result
    for v2 in ("0", "2"):                                    # each partial belongs to its own y (which differ in those few roundings)
        yf = out[v2][0].double()
        ref = torch.stack([yf.sum((0, 2, 3)), (yf * yf).sum((0, 2, 3))]).cpu()
        np.testing.assert_allclose(out[v2][1].double().sum(0).cpu().numpy(), ref.numpy(), rtol=2e-5, atol=2e-3)
    assert out["0"][1].shape == out["2"][1].shape            # same rows whichever kernel runs


# ---------------------------------------------------------------------------------------------------------------------
# tsg_conv3x3_s2_dgrad: the data gradient of the stride-2 3x3 convolutions by output parity (conv3s2d_k).  Oracle: autograd
# of F.conv2d(stride 2) in fp64 on the bf16-rounded operands.  Even and odd input sizes (OH = (H - 1) / 2 + 1), partial
# tiles, one / several dx-channel tiles, one chunk of 32 dy channels / many, the addend; the three layers of the bench
# step against the vendor library's backward-data on the whole batch.
S2_SHAPES = [(2, 64, 128, 16, 64), (1, 32, 32, 7, 9), (2, 128, 256, 18, 66), (1, 64, 96, 33, 31), (3, 32, 64, 1, 1),
             (1, 256, 512, 16, 64), (2, 96, 32, 20, 130)]


@pytest.mark.parametrize("shape", S2_SHAPES)
def test_stride2_data_gradient_vs_oracle(cuda, shape):
    from torchseg_amd import kernels as K
    kp = K.provider()
    B, Cin, Cout, H, W = shape
    assert kp.conv3x3_s2_dgrad_supported(Cin, Cout)
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    g = torch.Generator().manual_seed(sum(shape))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cout)) ** 0.5
    dy = torch.randn(B, Cout, OH, OW, generator=g)
    skip = torch.randn(B, Cin, H, W, generator=g)
    wd = w.to(cuda).contiguous(memory_format=torch.channels_last)
    dyb = dy.to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    dx = kp.conv3x3_s2_dgrad(dyb, wd, (H, W))
    assert tuple(dx.shape) == (B, Cin, H, W) and dx.is_contiguous(memory_format=torch.channels_last)
    xr = torch.zeros(B, Cin, H, W, dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, conv_ref.bf16_round(w).double(), None, 2, 1).backward(conv_ref.bf16_round(dy).double())
    _check(dx, xr.grad)
    # the bf16 copy of the master weight gives the same result (what the autograd node passes)
    assert torch.equal(dx, kp.conv3x3_s2_dgrad(dyb, wd.bfloat16().contiguous(memory_format=torch.channels_last), (H, W)))
    sb = skip.to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    assert torch.equal(kp.conv3x3_s2_dgrad(dyb, wd, (H, W), addend=sb), dx + sb)


@pytest.mark.parametrize("geom", [(16, 64, 128, 256), (16, 128, 256, 128), (16, 256, 512, 64)])
def test_stride2_data_gradient_bench_geometry_vs_library(cuda, geom):
    from torchseg_amd import kernels as K
    kp = K.provider()
    B, Cin, Cout, S = geom
    g = torch.Generator(device=cuda).manual_seed(Cin + S)
    x = torch.randn(B, Cin, S, S, generator=g, device=cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g, device=cuda) * (2.0 / (9 * Cout)) ** 0.5).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, Cout, S // 2, S // 2, generator=g, device=cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    dx = kp.conv3x3_s2_dgrad(dy, w, (S, S))
    lib = torch.ops.aten.convolution_backward(dy, x, w.bfloat16(), None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1,
                                              [True, False, False])[0]
    _vs_library(dx, lib)
    assert torch.equal(dx, kp.conv3x3_s2_dgrad(dy, w, (S, S)))


def test_shortcut_gradient_joins_the_stride2_data_gradient(cuda, monkeypatch):
    """BasicBlock with a 1x1 / stride-2 shortcut (resnet.py:139-146): conv1's node returns the input alias the shortcut
    branch reads, so the branch's gradient is the addend of tsg_conv3x3_s2_dgrad.  Against the same block with the vendor
    library's data gradient + autograd's accumulation (TSG_CONV_S2_DGRAD off): outputs bit-equal (the forward is the same),
    every gradient equal to bf16 rounding."""
    import copy, os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "torchseg_amd", "furnace"))
    from base_model.resnet import BasicBlock, _shortcut
    from torchseg_amd import convwrw
    from torchseg_amd.convwrw import install_conv_wrw
    from torchseg_amd.syncbn import SyncBatchNorm
    torch.manual_seed(9)
    proto = BasicBlock(64, 128, 2, norm_layer=SyncBatchNorm, downsample=_shortcut(64, 128, 2, SyncBatchNorm, 1e-5, 0.1))
    g = torch.Generator().manual_seed(10)
    x0 = torch.randn(2, 64, 24, 40, generator=g)
    dy0 = torch.randn(2, 128, 12, 20, generator=g)
    res = {}
    for own in (True, False):
        blk = copy.deepcopy(proto).to(cuda).to(memory_format=torch.channels_last)
        install_conv_wrw(blk)
        blk.train()
        kp = convwrw.K.provider()
        calls = []
        orig = kp.conv3x3_s2_dgrad

        def spy(*a, **k):
            calls.append(k.get("addend") is not None)
            return orig(*a, **k)
        kp.conv3x3_s2_dgrad = spy
        monkeypatch.setattr(convwrw, "_OWN_S2_DGRAD", own)
        try:
            xin = x0.to(cuda).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
            x = xin * 1.0
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = blk(x)
            out.backward(dy0.to(cuda).to(out.dtype).contiguous(memory_format=torch.channels_last))
        finally:
            del kp.conv3x3_s2_dgrad
        assert calls == ([True] if own else []), calls
        res[own] = [out.detach().float().cpu(), xin.grad.float().cpu()] + [p.grad.float().cpu() for p in blk.parameters()]
    assert torch.equal(res[True][0], res[False][0])
    for a, b in zip(res[True][1:], res[False][1:]):
        assert (a - b).abs().max().item() <= 2.0 ** -6 * b.abs().max().item() + 1e-6


# ---- round 6: the shortcut's gradient as a COMPACT addend (tsg_conv3x3_s2_dgrad_subadd) ---------------------------------
@pytest.mark.parametrize("shape", S2_SHAPES)
def test_stride2_data_gradient_with_compact_addend(cuda, shape):
    """addend_sub [B, Cin, OH, OW] added at the even pixels of dx == the full-size addend that is zero everywhere else
    (bit-equal: both compute bf16(bf16(conv) + a), and x + 0 = x), even and odd sizes, partial tiles."""
    from torchseg_amd import kernels as K
    kp = K.provider()
    B, Cin, Cout, H, W = shape
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    g = torch.Generator().manual_seed(sum(shape) + 1)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cout)) ** 0.5).to(cuda).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, Cout, OH, OW, generator=g).to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    sub = torch.randn(B, Cin, OH, OW, generator=g).to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    full = torch.zeros(B, Cin, H, W, device=cuda, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    full[:, :, ::2, ::2] = sub
    want = kp.conv3x3_s2_dgrad(dy, w, (H, W), addend=full)
    got = kp.conv3x3_s2_dgrad(dy, w, (H, W), addend_sub=sub)
    plain = kp.conv3x3_s2_dgrad(dy, w, (H, W))
    # (-0 + 0 = +0: compare values, and the untouched pixels against the plain launch)
    assert torch.equal(got.float(), want.float())
    odd = torch.ones(H, W, dtype=torch.bool, device=cuda)
    odd[::2, ::2] = False
    assert torch.equal(got.float()[:, :, odd], plain.float()[:, :, odd])
    with pytest.raises(ValueError):
        kp.conv3x3_s2_dgrad(dy, w, (H, W), addend=full, addend_sub=sub)


def test_shortcut_on_the_subsampled_map_equals_the_stride2_shortcut(cuda, monkeypatch):
    """BasicBlock with a 1x1 / stride-2 shortcut: with the shortcut convolution on torchseg_amd.pwconv, conv1's node hands it
    x[:, :, ::2, ::2] as a compact tensor, the shortcut runs as a stride-1 convolution of that, and its gradient joins the
    stride-2 data gradient as a compact addend.  Against the same block with TSG_SKIP_SUBSAMPLE off (full-size alias,
    vendor stride-2 shortcut): same mathematics, other kernels for the shortcut -> everything within bf16 rounding."""
    import copy, os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "torchseg_amd", "furnace"))
    from base_model.resnet import BasicBlock, _shortcut
    from torchseg_amd import convwrw
    from torchseg_amd.convwrw import install_conv_wrw
    from torchseg_amd.pwconv import install_pointwise_conv
    from torchseg_amd.syncbn import SyncBatchNorm
    torch.manual_seed(19)
    proto = BasicBlock(64, 128, 2, norm_layer=SyncBatchNorm, downsample=_shortcut(64, 128, 2, SyncBatchNorm, 1e-5, 0.1))
    g = torch.Generator().manual_seed(20)
    x0 = torch.randn(2, 64, 26, 38, generator=g)
    dy0 = torch.randn(2, 128, 13, 19, generator=g)
    res = {}
    for subs in (True, False):
        blk = copy.deepcopy(proto).to(cuda).to(memory_format=torch.channels_last)
        install_conv_wrw(blk)
        assert install_pointwise_conv(blk) == 1
        blk.train()
        kp = convwrw.K.provider()
        calls = []
        orig = kp.conv3x3_s2_dgrad

        def spy(*a, **k):
            calls.append("sub" if k.get("addend_sub") is not None else "full" if k.get("addend") is not None else "none")
            return orig(*a, **k)
        kp.conv3x3_s2_dgrad = spy
        monkeypatch.setattr(convwrw, "_SKIP_SUB", subs)
        try:
            xin = x0.to(cuda).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
            x = xin * 1.0
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = blk(x)
            out.backward(dy0.to(cuda).to(out.dtype).contiguous(memory_format=torch.channels_last))
        finally:
            del kp.conv3x3_s2_dgrad
        assert calls == (["sub"] if subs else ["full"]), calls
        res[subs] = [out.detach().float().cpu(), xin.grad.float().cpu()] + [p.grad.float().cpu() for p in blk.parameters()]
    for a, b in zip(res[True], res[False]):
        assert (a - b).abs().max().item() <= 2.0 ** -6 * b.abs().max().item() + 1e-6
