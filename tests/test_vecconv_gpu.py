"""GPU parity of tsg_conv1x1_vec_fwd / _bwd (csrc/vecconv.hip, through the C-ABI): the bias-free 1x1 convolutions that follow
nn.AdaptiveAvgPool2d(1) (furnace/seg_opr/seg_oprs.py:199-205, :222-231; bisenet network.py:34-39).  Oracle: the same
products in fp64 on the bf16-rounded operands (what autocast feeds the vendor library).  y and dx are bf16 -> one bf16 ulp
of the fp64 result; dw is fp32 and not rounded -> 1e-5 of its scale."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (B, Cin, Cout): the five layers of BiSeNet-R18 at the bench batch, then ragged / small / maximal batches
SHAPES = [(16, 512, 128), (16, 128, 128), (16, 256, 256), (2, 128, 128), (32, 256, 64), (5, 48, 80), (1, 16, 16), (7, 2048, 512)]


def _bf(t):
    return t.bfloat16().double()


@pytest.mark.parametrize("shape", SHAPES)
def test_forward_and_both_gradients_vs_fp64(cuda, shape):
    from torchseg_amd import kernels as K
    kp = K.provider()
    B, Cin, Cout = shape
    g = torch.Generator().manual_seed(B + Cin + Cout)
    x = torch.randn(B, Cin, 1, 1, generator=g).abs()                      # pooled activations are non-negative after ReLU
    w = torch.randn(Cout, Cin, 1, 1, generator=g) * (2.0 / Cin) ** 0.5
    dy = torch.randn(B, Cout, 1, 1, generator=g)
    xb, wd, dyb = x.to(cuda).bfloat16(), w.to(cuda), dy.to(cuda).bfloat16()
    assert kp.conv1x1_vec_supported(xb, wd)
    y = kp.conv1x1_vec_fwd(xb, wd)
    assert y.shape == (B, Cout, 1, 1) and y.dtype == torch.bfloat16
    y_ref = _bf(x).view(B, Cin) @ _bf(w).view(Cout, Cin).t()
    err = (y.double().cpu().view(B, Cout) - y_ref).abs()
    assert bool((err <= y_ref.abs() * 2.0 ** -8 + 1e-3 * y_ref.abs().max()).all()), err.max().item()
    dx, dw = kp.conv1x1_vec_bwd(dyb, xb, wd)
    dx_ref = _bf(dy).view(B, Cout) @ _bf(w).view(Cout, Cin)
    err = (dx.double().cpu().view(B, Cin) - dx_ref).abs()
    assert bool((err <= dx_ref.abs() * 2.0 ** -8 + 1e-3 * dx_ref.abs().max()).all()), err.max().item()
    dw_ref = _bf(dy).view(B, Cout).t() @ _bf(x).view(B, Cin)
    assert dw.dtype == torch.float32 and dw.shape == (Cout, Cin, 1, 1)
    assert (dw.double().cpu().view(Cout, Cin) - dw_ref).abs().max().item() <= 1e-5 * dw_ref.abs().max().item()
    # no data gradient requested: same dw, no dx
    dx2, dw2 = kp.conv1x1_vec_bwd(dyb, xb, wd, need_dx=False)
    assert dx2 is None and torch.equal(dw, dw2)
    # run-to-run equal
    assert torch.equal(y, kp.conv1x1_vec_fwd(xb, wd))


def test_module_swap_matches_the_stock_convolution_under_autocast(cuda):
    """install_pooled_conv on a channel-attention branch: same outputs / gradients as nn.Conv2d under bf16 autocast to bf16
    rounding, state-dict keys untouched, inputs that are not [B, C, 1, 1] take the stock path."""
    from torchseg_amd.vecconv import PooledConv2d, install_pooled_conv
    torch.manual_seed(3)
    ref = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(128, 64, 1, bias=False), nn.ReLU(), nn.Conv2d(64, 128, 1, bias=False),
                        nn.Sigmoid()).to(cuda)
    import copy
    ours = copy.deepcopy(ref)
    assert install_pooled_conv(ours) == 2 and isinstance(ours[1], PooledConv2d)
    assert list(ours.state_dict().keys()) == list(ref.state_dict().keys())
    x0 = torch.randn(16, 128, 12, 20, device=cuda)
    outs = []
    for m in (ref, ours):
        x = x0.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = m(x)
        out.float().square().sum().backward()
        outs.append((out.float(), x.grad, [p.grad for p in m.parameters()]))
    (o_r, gx_r, gp_r), (o_o, gx_o, gp_o) = outs
    assert (o_r - o_o).abs().max().item() <= 2.0 ** -7
    assert (gx_r - gx_o).abs().max().item() <= 2e-2 * gx_r.abs().max().item()
    for a, b in zip(gp_r, gp_o):
        assert b.dtype == torch.float32 and (a - b).abs().max().item() <= 2e-2 * a.abs().max().item()
    # a full-size map through the same module: the stock convolution
    with torch.autocast("cuda", dtype=torch.bfloat16):
        a = ours[1](x0)
        b = F.conv2d(x0, ours[1].weight)
    assert a.shape == b.shape and (a.float() - b.float()).abs().max().item() <= 2.0 ** -6 * b.float().abs().max().item()


def _branch(cuda, cin, cout, has_bn, has_relu, sigmoid, seed):
    """Sequential(GlobalAvgPool-less) ConvBnRelu [+ Sigmoid] on a pooled map, our furnace modules with the HIP installers applied"""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "torchseg_amd", "furnace"))
    from seg_opr.seg_oprs import ConvBnRelu
    from torchseg_amd.syncbn import SyncBatchNorm
    from torchseg_amd.vecconv import install_pooled_conv
    torch.manual_seed(seed)
    mods = [nn.AdaptiveAvgPool2d(1), ConvBnRelu(cin, cout, 1, 1, 0, has_bn=has_bn, norm_layer=SyncBatchNorm, has_relu=has_relu,
                                                has_bias=False)]
    if sigmoid:
        mods.append(nn.Sigmoid())
    seq = nn.Sequential(*mods).to(cuda)
    if has_bn:
        with torch.no_grad():
            seq[1].bn.weight.copy_(torch.randn(cout, device=cuda) * 0.5 + 1.0)
            seq[1].bn.bias.copy_(torch.randn(cout, device=cuda) * 0.3)
    assert install_pooled_conv(seq) == 1
    return seq


@pytest.mark.parametrize("cfg", [(128, 128, True, False, True), (512, 128, True, True, False), (256, 256, False, True, False),
                                 (256, 256, False, False, True), (64, 48, True, False, False)])
@pytest.mark.parametrize("train", [True, False])
def test_pooled_layer_in_one_launch_equals_the_module_sequence(cuda, cfg, train):
    """tsg_conv1x1_vec_bnact_* (vecconv.pooled_layer: convolution + BatchNorm over the batch + ReLU / Sigmoid as ONE launch per
    direction) against the same modules run one by one (PooledConv2d -> SyncBatchNorm -> activation), which test_bn_gpu.py /
    the tests above hold to the oracle: same rounding points, the 16-term sums in fp64 instead of fp32 partial rows."""
    from torchseg_amd import vecconv, kernels as K
    cin, cout, has_bn, has_relu, sigmoid = cfg
    B = 16
    res = {}
    for fused in (True, False):
        seq = _branch(cuda, cin, cout, has_bn, has_relu, sigmoid, seed=cin + cout)
        seq.train(train)
        if has_bn and not train:
            with torch.no_grad():
                seq[1].bn.running_mean.copy_(torch.randn(cout, device=cuda) * 0.1)
                seq[1].bn.running_var.copy_(torch.rand(cout, device=cuda) + 0.5)
        g = torch.Generator(device=cuda).manual_seed(7)
        x = (torch.randn(B, cin, 1, 1, device=cuda, generator=g) * 2).bfloat16().requires_grad_(True)
        old = vecconv.POOLED_LAYER
        vecconv.POOLED_LAYER = fused
        cnt = K.CallCounter(K.provider())
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                if sigmoid and fused:
                    from torchseg_amd.pool import _run_pooled_branch
                    y = _run_pooled_branch(list(seq.children())[1:], x)
                else:
                    y = x
                    for m in list(seq.children())[1:]:
                        y = m(y)
            gy = torch.randn(y.shape, device=cuda, generator=g).bfloat16()
            y.backward(gy)
        finally:
            vecconv.POOLED_LAYER = old
            calls = cnt.stop()
        torch.cuda.synchronize()
        assert (calls.get("conv1x1_vec_bnact_fwd", 0) == 1) == fused and (calls.get("conv1x1_vec_bnact_bwd", 0) == 1) == fused, calls
        bn = seq[1].bn if has_bn else None
        res[fused] = dict(y=y.detach().float(), dx=x.grad.float(), dw=seq[1].conv.weight.grad.clone(),
                          dg=bn.weight.grad.clone() if bn is not None and train else None,
                          db=bn.bias.grad.clone() if bn is not None and train else None,
                          rm=bn.running_mean.clone() if bn is not None else None,
                          rv=bn.running_var.clone() if bn is not None else None,
                          nbt=int(bn.num_batches_tracked) if bn is not None else None)
    a, b = res[True], res[False]
    # outputs: the same bf16 value except where the fp64 / fp32-partial statistics differ in the last place
    assert (a["y"] - b["y"]).abs().max().item() <= 2.0 ** -7 * max(1.0, b["y"].abs().max().item())
    assert (a["y"] != b["y"]).float().mean().item() < 0.02
    for k, tol in (("dx", 2e-2), ("dw", 2e-2), ("dg", 2e-2), ("db", 2e-2)):
        if b[k] is None:
            continue
        err = ((a[k].double() - b[k].double()).norm() / b[k].double().norm().clamp_min(1e-12)).item()
        assert err < tol, (k, err)
    if has_bn:
        assert torch.allclose(a["rm"], b["rm"], rtol=1e-5, atol=1e-6) and torch.allclose(a["rv"], b["rv"], rtol=1e-5, atol=1e-6)
        assert a["nbt"] == b["nbt"]
