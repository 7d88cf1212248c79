"""GPU parity of the classifier-convolution kernels of a head (csrc/clshead.hip, through the C-ABI): nn.Conv2d(C_in, n, 1)
+ bias of bisenet network.py:151-161 on a channels_last bf16 map, producing PLANAR logits.  Against oracle/conv_ref.py
(fp64, tap by tap) and torch's fp64 autograd on the same bf16-rounded operands: forward to one bf16 ulp, data gradient to
one bf16 ulp, weight / bias gradient to fp32 accumulation accuracy; run-to-run bit-identical; the three bench shapes
(16 x 256 x 128^2, 16 x 64 x 128^2, 16 x 256 x 64^2 -> 19) and ragged ones; the re-classed module inside autocast."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import conv_ref

pytestmark = pytest.mark.gpu

# (B, Cin, H, W, N, bias)
CASES = [(16, 256, 128, 128, 19, True), (16, 64, 128, 128, 19, True), (16, 256, 64, 64, 19, True),
         (2, 128, 8, 14, 19, True), (1, 32, 4, 4, 7, False), (3, 32, 8, 10, 1, True), (2, 256, 12, 12, 32, True),
         (3, 64, 20, 12, 21, True)]


@pytest.mark.parametrize("case", CASES)
def test_cls_head_kernels_vs_oracle(cuda, case):
    from torchseg_amd import kernels as K
    kp = K.provider()
    B, Cin, H, W, N, has_bias = case
    g = torch.Generator().manual_seed(sum(case[:5]))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(N, Cin, 1, 1, generator=g) * (1.0 / Cin) ** 0.5
    bias = torch.randn(N, generator=g) if has_bias else None
    dz = torch.randn(B, N, H, W, generator=g)
    xd = x.to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    wd = w.to(cuda)
    bd = bias.to(cuda) if has_bias else None
    assert kp.cls_head_supported(xd, wd)
    z = kp.cls_head_fwd(xd, wd, bd)
    assert z.dtype == torch.bfloat16 and z.is_contiguous() and tuple(z.shape) == (B, N, H, W)
    assert torch.equal(z, kp.cls_head_fwd(xd, wd, bd))
    xr = conv_ref.bf16_round(x).requires_grad_(True)
    wr = conv_ref.bf16_round(w).requires_grad_(True)
    br = bias.double().requires_grad_(True) if has_bias else None
    want = conv_ref.conv2d_ref(xr.detach(), wr.detach(), stride=1, pad=0)
    if has_bias:
        want = want + br.detach().view(1, -1, 1, 1)
    err = (z.double().cpu() - want).abs()
    assert bool((err <= want.abs() * 2.0 ** -8 + 1e-3 * want.abs().max()).all()), err.max().item()
    # backward
    dzd = dz.to(cuda).bfloat16().contiguous()
    dx, dw, db = kp.cls_head_bwd(dzd, xd, wd, need_dx=True, need_db=has_bias)
    dx2, dw2, db2 = kp.cls_head_bwd(dzd, xd, wd, need_dx=True, need_db=has_bias)
    assert torch.equal(dx, dx2) and torch.equal(dw, dw2) and (not has_bias or torch.equal(db, db2))
    yr = F.conv2d(xr, wr, br, 1, 0)
    yr.backward(conv_ref.bf16_round(dz))
    assert dx.is_contiguous(memory_format=torch.channels_last) or Cin == 1
    e = (dx.double().cpu() - xr.grad).abs()
    assert bool((e <= xr.grad.abs() * 2.0 ** -8 + 1e-3 * xr.grad.abs().max()).all()), e.max().item()
    assert (dw.double().cpu() - wr.grad).abs().max().item() <= 1e-4 * wr.grad.abs().max().item() + 1e-6
    if has_bias:
        assert (db.double().cpu() - br.grad).abs().max().item() <= 1e-4 * br.grad.abs().max().item() + 1e-6


def test_reclassed_head_convolution_inside_autocast(cuda):
    """install_cls_head on a head as the reference builds it (conv_3x3 output -> conv_1x1): same parameters and state-dict
    keys, planar bf16 logits, gradients against fp64; shapes the kernels do not cover (150 classes) keep their module."""
    from torchseg_amd.clshead import ClsHeadConv2d, install_cls_head
    torch.manual_seed(1)
    head = nn.Sequential(nn.Conv2d(64, 19, 1), nn.Conv2d(64, 150, 1)).to(cuda)
    keys = list(head.state_dict().keys())
    assert install_cls_head(head) == 1 and isinstance(head[0], ClsHeadConv2d) and type(head[1]) is nn.Conv2d
    assert list(head.state_dict().keys()) == keys
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 64, 24, 20, generator=g)
    dz = torch.randn(2, 19, 24, 20, generator=g)
    xd = x.to(cuda).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        z = head[0](xd)
    assert z.is_contiguous() and z.dtype == torch.bfloat16
    z.backward(dz.to(cuda).bfloat16())
    xr = conv_ref.bf16_round(x).requires_grad_(True)
    wr = conv_ref.bf16_round(head[0].weight.detach().cpu().float()).requires_grad_(True)
    br = head[0].bias.detach().cpu().double().requires_grad_(True)
    F.conv2d(xr, wr, br).backward(conv_ref.bf16_round(dz))
    assert (xd.grad.double().cpu() - xr.grad).abs().max().item() <= 2.0 ** -7 * xr.grad.abs().max().item()
    assert (head[0].weight.grad.double().cpu() - wr.grad).abs().max().item() <= 1e-4 * wr.grad.abs().max().item()
    assert (head[0].bias.grad.double().cpu() - br.grad).abs().max().item() <= 1e-4 * br.grad.abs().max().item()
    # fp32 activations outside autocast: the module's ordinary path
    y32 = head[0](x.to(cuda))
    assert y32.dtype == torch.float32
