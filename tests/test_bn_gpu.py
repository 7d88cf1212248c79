"""GPU parity: HIP SyncBN kernels (through the SyncBatchNorm module, i.e. the
C-ABI) vs the numpy oracle.  fp32: 1e-4 (north_star tolerance); bf16: the
oracle is fed the same bf16-rounded inputs and compared at bf16 resolution."""
import numpy as np
import pytest
import torch

from oracle import syncbn_ref as R

pytestmark = pytest.mark.gpu

SHAPES = [(2, 64, 33, 47), (4, 128, 16, 16), (16, 128, 1, 1), (3, 24, 8, 8), (2, 19, 7, 5),
          (2, 2048, 6, 6), (2, 4096, 4, 4), (2, 64, 96, 128), (1, 8, 300, 300)]


def _run(cuda, shape, layout, dtype, relu, res, seed=0):
    from torchseg_amd.syncbn import SyncBatchNorm
    N, C, H, W = shape
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g) * 1.5 + 0.4
    r = torch.randn(shape, generator=g) if res else None
    dy = torch.randn(shape, generator=g)
    gamma = torch.randn(C, generator=g) * 0.5 + 1.0
    beta = torch.randn(C, generator=g) * 0.5
    x, dy = x.to(dtype), dy.to(dtype)
    if res:
        r = r.to(dtype)
    fmt = torch.channels_last if layout == "nhwc" else torch.contiguous_format
    bn = SyncBatchNorm(C, eps=1e-5, momentum=0.1).to(cuda)
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta)
    xd = x.to(cuda).contiguous(memory_format=fmt).requires_grad_(True)
    rd = r.to(cuda).contiguous(memory_format=fmt).requires_grad_(True) if res else None
    y = bn(xd, residual=rd, relu=relu)
    assert y.dtype == dtype and y.stride() == xd.stride()
    y.backward(dy.to(cuda).contiguous(memory_format=fmt))
    torch.cuda.synchronize()

    xs = [x.float().numpy().astype(np.float64)]
    rs = [r.float().numpy().astype(np.float64)] if res else None
    ys, mean, inv_std, rm, rv = R.forward(xs, gamma.numpy(), beta.numpy(), 1e-5, 0.1,
                                          np.zeros(C), np.ones(C), rs, relu)
    y_ref = ys[0]
    if dtype == torch.bfloat16:
        # the device masks with its own (bf16-rounded) y; use the same mask source
        ys = [y.detach().float().cpu().numpy().astype(np.float64)] if res else ys
    dxs, dres, dg, db = R.backward(xs, [dy.float().numpy()], ys, gamma.numpy(), mean, inv_std, relu)
    return dict(y=y.detach().float().cpu().numpy(), y_ref=y_ref,
                dx=xd.grad.float().cpu().numpy(), dx_ref=dxs[0],
                dres=rd.grad.float().cpu().numpy() if res else None, dres_ref=dres[0],
                dg=bn.weight.grad.cpu().numpy(), dg_ref=dg[0], db=bn.bias.grad.cpu().numpy(), db_ref=db[0],
                rm=bn.running_mean.cpu().numpy(), rm_ref=rm, rv=bn.running_var.cpu().numpy(), rv_ref=rv,
                nbt=int(bn.num_batches_tracked.item()))


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
@pytest.mark.parametrize("relu,res", [(False, False), (True, False), (True, True), (False, True)])
def test_bn_fp32(cuda, shape, layout, relu, res):
    o = _run(cuda, shape, layout, torch.float32, relu, res)
    n = shape[0] * shape[2] * shape[3]
    np.testing.assert_allclose(o["y"], o["y_ref"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(o["dx"], o["dx_ref"], rtol=1e-4, atol=1e-4)
    if res:
        np.testing.assert_allclose(o["dres"], o["dres_ref"], rtol=0, atol=0)
    np.testing.assert_allclose(o["dg"], o["dg_ref"], rtol=1e-4, atol=1e-4 * np.sqrt(n))
    np.testing.assert_allclose(o["db"], o["db_ref"], rtol=1e-4, atol=1e-4 * np.sqrt(n))
    np.testing.assert_allclose(o["rm"], o["rm_ref"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(o["rv"], o["rv_ref"], rtol=1e-4, atol=1e-6)
    assert o["nbt"] == 1


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
@pytest.mark.parametrize("relu,res", [(True, False), (True, True)])
def test_bn_bf16(cuda, shape, layout, relu, res):
    o = _run(cuda, shape, layout, torch.bfloat16, relu, res)
    n = shape[0] * shape[2] * shape[3]
    # outputs are rounded to bf16 (8 bits of mantissa): 2^-8 relative
    np.testing.assert_allclose(o["y"], o["y_ref"], rtol=8e-3, atol=8e-3)
    # a ReLU mask decided on a value that rounds to 0 in bf16 may differ: allow a handful
    bad = np.abs(o["dx"] - o["dx_ref"]) > (8e-3 + 8e-3 * np.abs(o["dx_ref"]))
    assert bad.mean() < 1e-3, bad.mean()
    np.testing.assert_allclose(o["dg"], o["dg_ref"], rtol=2e-2, atol=2e-2 * np.sqrt(n))
    np.testing.assert_allclose(o["db"], o["db_ref"], rtol=2e-2, atol=2e-2 * np.sqrt(n))
    np.testing.assert_allclose(o["rm"], o["rm_ref"], rtol=1e-4, atol=1e-5)


def test_bn_eval_and_errors(cuda):
    from torchseg_amd.syncbn import SyncBatchNorm
    bn = SyncBatchNorm(16).to(cuda)
    x = torch.randn(4, 16, 9, 9, device=cuda)
    bn.train()
    for _ in range(3):
        bn(x)
    bn.eval()
    ref = torch.nn.functional.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.1, bn.eps)
    torch.testing.assert_close(bn(x), ref, rtol=1e-5, atol=1e-5)
    bn.train()
    with pytest.raises(ValueError):
        bn(torch.randn(1, 16, 1, 1, device=cuda))
    with pytest.raises(Exception):
        bn(torch.randn(4, 16, 3, 3))  # CPU tensor: no fallback


@pytest.mark.parametrize("shape", [(2, 64, 32, 32), (3, 64, 40, 24), (2, 128, 16, 16), (2, 8, 8, 12), (4, 64, 128, 128)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("relu", [True, False])
def test_bn_mixed_layout_stem(cuda, shape, dtype, relu):
    """NCHW conv output -> BN(+ReLU) -> channels_last output (and NHWC dy -> NCHW dx)
    must equal the plain NCHW path bit-for-bit in fp32 and the oracle within tolerance."""
    from torchseg_amd import syncbn
    from torchseg_amd.syncbn import SyncBatchNorm
    N, C, H, W = shape
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(shape, generator=g) * 1.3 + 0.2).to(dtype)
    dy = torch.randn(shape, generator=g).to(dtype)
    outs = []
    for prefer in (False, True):
        syncbn.PREFER_CHANNELS_LAST_OUTPUT = prefer
        try:
            bn = SyncBatchNorm(C).to(cuda)
            with torch.no_grad():
                bn.weight.copy_(torch.linspace(0.5, 1.5, C)); bn.bias.copy_(torch.linspace(-0.3, 0.3, C))
            xd = x.to(cuda).requires_grad_(True)
            y = bn(xd, relu=relu)
            y.backward(dy.to(cuda).contiguous(memory_format=torch.channels_last if prefer else torch.contiguous_format))
            outs.append((y.detach().float().cpu(), xd.grad.float().cpu(), bn.weight.grad.cpu(), bn.bias.grad.cpu(),
                         y.is_contiguous(memory_format=torch.channels_last) and not y.is_contiguous()))
        finally:
            syncbn.PREFER_CHANNELS_LAST_OUTPUT = False
    (y0, dx0, dg0, db0, cl0), (y1, dx1, dg1, db1, cl1) = outs
    assert cl1 and not cl0
    assert xd.grad.is_contiguous()
    torch.testing.assert_close(y1, y0, rtol=0, atol=0)
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(dx1, dx0, **tol)
    n = N * H * W
    torch.testing.assert_close(dg1, dg0, rtol=1e-4, atol=1e-4 * n ** 0.5)
    torch.testing.assert_close(db1, db0, rtol=1e-4, atol=1e-4 * n ** 0.5)


# ---- round 6: the block tail's ReLU mask as one bit per element ------------------------------------------------------
@pytest.mark.parametrize("shape", [(2, 64, 33, 47), (4, 128, 16, 16), (16, 128, 1, 1), (2, 64, 96, 128), (1, 8, 300, 300),
                                   (3, 24, 8, 8)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_block_tail_with_bit_mask_equals_the_stored_output_mask(cuda, shape, dtype, monkeypatch):
    """relu(bn(x) + identity) through tsg_bn_apply_fwd_maskbits / tsg_bn_bwd_*_maskbits: the output, the mask bits
    (= y > 0 of the ROUNDED output), both input gradients and the parameter gradients are BIT-equal to the path that reads the
    mask from the stored output (same arithmetic in the same order; only where the mask comes from differs).  Shapes whose
    channel count is not a multiple of the bits' group (24 for bf16: 8 per byte fits; 19 would not) fall back silently."""
    from torchseg_amd import kernels as K, syncbn
    from torchseg_amd.syncbn import SyncBatchNorm
    kp = K.provider()
    N, C, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(shape, generator=g) * 1.5 + 0.4).to(dtype)
    r = torch.randn(shape, generator=g).to(dtype)
    dy = torch.randn(shape, generator=g).to(dtype)
    cl = dict(memory_format=torch.channels_last)

    def run(bits):
        monkeypatch.setattr(syncbn, "_MASKBITS", bits)
        bn = SyncBatchNorm(C).to(cuda)
        with torch.no_grad():
            bn.weight.copy_(torch.randn(C, generator=torch.Generator().manual_seed(1)) * 0.5 + 1.0)
            bn.bias.copy_(torch.randn(C, generator=torch.Generator().manual_seed(2)) * 0.5)
        xd = x.to(cuda).contiguous(**cl).requires_grad_(True)
        rd = r.to(cuda).contiguous(**cl).requires_grad_(True)
        cnt = K.CallCounter(kp)
        try:
            y = bn(xd, residual=rd, relu=True)
            y.backward(dy.to(cuda).contiguous(**cl))
        finally:
            counts = cnt.stop()
        torch.cuda.synchronize()
        return y.detach(), xd.grad, rd.grad, bn.weight.grad, bn.bias.grad, bn.running_var.clone(), counts

    ref = run(False)
    got = run(True)
    layout, _, _, HW = K.bn_layout(x.to(cuda).contiguous(**cl))
    if kp.bn_maskbits_supported(x.to(cuda).contiguous(**cl), layout, C, HW):
        assert got[6].get("bn_apply_fwd_bits") == 1 and got[6].get("bn_bwd_reduce_bits") == 1 \
            and got[6].get("bn_bwd_apply_bits") == 1 and "bn_bwd_reduce" not in got[6], got[6]
    assert "bn_apply_fwd_bits" not in ref[6]
    for name, a, b in zip(("y", "dx", "dres", "dgamma", "dbeta", "running_var"), got[:6], ref[:6]):
        assert torch.equal(a, b), name


def test_mask_bits_are_the_sign_of_the_rounded_output(cuda):
    from torchseg_amd import kernels as K
    kp = K.provider()
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(2, 64, 24, 40, generator=g) * 1e-3).to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    r = (torch.randn(2, 64, 24, 40, generator=g) * 1e-3).to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    fp = torch.stack([torch.rand(64, generator=g) - 0.5, torch.randn(64, generator=g) * 1e-3, torch.zeros(64)]).to(cuda).contiguous()
    layout, N, C, HW = K.bn_layout(x)
    y, bits = kp.bn_apply_fwd_bits(x, r, layout, N, C, HW, fp)
    assert torch.equal(y, kp.bn_apply_fwd(x, r, layout, N, C, HW, fp, True))
    want = (y.permute(0, 2, 3, 1).reshape(-1, 8) > 0).to(torch.uint8)          # NHWC order, 8 channels per byte, bit j = channel j
    packed = (want << torch.arange(8, device=cuda, dtype=torch.uint8)).sum(1).to(torch.uint8)
    assert torch.equal(bits, packed)
    assert 0.2 < want.float().mean().item() < 0.8
