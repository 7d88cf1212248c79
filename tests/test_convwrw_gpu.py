"""GPU parity of tsg_conv3x3_wrw (through the C-ABI) with oracle/conv_ref.py on the same bf16-rounded operands:
fp32 output, 1e-4 relative L2 (north_star's fp32 tolerance), run-to-run bit-identical; and the module swap
(WrwConv2d) against stock autograd under autocast."""
import pytest
import torch

from oracle import conv_ref

pytestmark = pytest.mark.gpu


def _check(cuda, B, H, W, seed=0, variant="tr"):
    from torchseg_amd import kernels as K
    kp = K.provider()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 64, H, W, generator=g)
    dy = torch.randn(B, 64, H, W, generator=g)
    xb = x.to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    dyb = dy.to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    dw = kp.conv3x3_wrw(xb, dyb, variant=variant)
    assert dw.shape == (64, 64, 3, 3) and dw.dtype == torch.float32
    want = conv_ref.conv2d_wgrad_ref(conv_ref.bf16_round(x), conv_ref.bf16_round(dy), ksize=3, stride=1, pad=1)
    rel = ((dw.double().cpu() - want).norm() / want.norm()).item()
    assert rel <= 1e-4, rel
    return dw


@pytest.mark.parametrize("variant", ["tr", "v1", "gen"])
@pytest.mark.parametrize("shape", [(1, 8, 32), (2, 6, 40), (1, 3, 5), (3, 17, 70), (2, 64, 64)])
def test_conv3x3_wrw_vs_oracle(cuda, shape, variant):
    _check(cuda, *shape, variant=variant)


def test_conv3x3_wrw_layer1_size_and_determinism(cuda):
    """ResNet-18 layer1 geometry at BASELINE config 2 (256 x 256 maps; B = 4 keeps the fp64 oracle to seconds)."""
    for variant in ("tr", "v1", "gen"):
        a = _check(cuda, 4, 256, 256, seed=5, variant=variant)
        b = _check(cuda, 4, 256, 256, seed=5, variant=variant)
        assert torch.equal(a, b), variant


def test_wrw_conv_module_matches_stock_autocast(cuda):
    import torch.nn as nn
    from torchseg_amd.convwrw import WrwConv2d, install_conv_wrw
    torch.manual_seed(0)
    ref = nn.Conv2d(64, 64, 3, 1, 1, bias=False).to(cuda)
    ref.weight.data = ref.weight.data.contiguous(memory_format=torch.channels_last)
    mod = nn.Sequential(nn.Conv2d(64, 64, 3, 1, 1, bias=False)).to(cuda)
    mod[0].load_state_dict(ref.state_dict())
    mod[0].weight.data = mod[0].weight.data.contiguous(memory_format=torch.channels_last)
    assert install_conv_wrw(mod) == 1 and isinstance(mod[0], WrwConv2d)
    x0 = torch.randn(2, 64, 40, 72, device=cuda).contiguous(memory_format=torch.channels_last).requires_grad_()
    x1 = x0.detach().clone().requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y0, y1 = ref(x0), mod(x1)
    assert y1.dtype == torch.bfloat16
    torch.testing.assert_close(y1.float(), y0.float(), rtol=2e-2, atol=2e-2)
    dy = torch.randn_like(y0)
    y0.backward(dy)
    y1.backward(dy)
    g0, g1 = ref.weight.grad, mod[0].weight.grad
    assert g1.dtype == torch.float32 and g1.shape == g0.shape
    assert ((g1 - g0).norm() / g0.norm()).item() <= 1e-2          # MIOpen rounds its result to bf16, ours stays fp32
    assert ((x1.grad - x0.grad).norm() / x0.grad.norm()).item() <= 1e-2


# ---- pair-tiled kernel: any C_in, C_out multiple of 64 -----------------------------------------------------------------
def _check_gen(cuda, B, H, W, Cin, Cout, seed=0, stride=1):
    from torchseg_amd import kernels as K
    kp = K.provider()
    g = torch.Generator().manual_seed(seed + Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    dy = torch.randn(B, Cout, (H - 1) // stride + 1, (W - 1) // stride + 1, generator=g)
    # channel- and position-dependent scaling: a swapped (oc tile, ci tile) pair or tap cannot pass
    x = x * (1.0 + 0.01 * torch.arange(Cin).view(1, Cin, 1, 1)) + 0.1 * torch.arange(W).view(1, 1, 1, W) / W
    dy = dy * (1.0 + 0.02 * torch.arange(Cout).view(1, Cout, 1, 1))
    xb = x.to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    dyb = dy.to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    dw = kp.conv3x3_wrw(xb, dyb, variant="gen", stride=stride)
    assert dw.shape == (Cout, Cin, 3, 3) and dw.dtype == torch.float32
    assert dw.is_contiguous(memory_format=torch.channels_last)
    want = conv_ref.conv2d_wgrad_ref(conv_ref.bf16_round(x), conv_ref.bf16_round(dy), ksize=3, stride=stride, pad=1)
    rel = ((dw.double().cpu() - want).norm() / want.norm()).item()
    assert rel <= 1e-4, rel
    worst = ((dw.double().cpu() - want).abs().amax(dim=(2, 3)) / want.abs().amax()).max().item()
    assert worst <= 1e-4, worst                       # every (oc, ci) pair individually
    return dw


@pytest.mark.parametrize("shape", [(1, 8, 32, 128, 64), (2, 6, 40, 64, 128), (1, 5, 7, 128, 192), (2, 12, 36, 256, 128),
                                   (3, 9, 33, 64, 64), (1, 4, 32, 512, 512)])
def test_conv3x3_wrw_gen_vs_oracle(cuda, shape):
    _check_gen(cuda, *shape)


@pytest.mark.parametrize("shape", [(1, 16, 64, 64, 64), (2, 13, 70, 64, 128), (1, 9, 33, 128, 64), (2, 8, 64, 256, 128),
                                   (1, 2, 2, 64, 64), (1, 31, 129, 64, 64)])
def test_conv3x3_wrw_gen_stride2_vs_oracle(cuda, shape):
    """bisenet network.py:114-137 (spatial path) and resnet.py layer2-4 first blocks: 3x3 / stride 2 / padding 1, even and
    odd input sizes (OH = (Hin - 1) // 2 + 1)."""
    _check_gen(cuda, *shape, stride=2)


def test_conv3x3_wrw_gen_stride2_spatial_path_size_vs_miopen(cuda):
    from torchseg_amd import kernels as K
    kp = K.provider()
    for (B, Cin, Cout, Hin) in [(16, 64, 64, 512), (16, 128, 256, 128)]:
        g = torch.Generator(device=cuda).manual_seed(Cin)
        x = torch.randn(B, Cin, Hin, Hin, generator=g, device=cuda).bfloat16().contiguous(memory_format=torch.channels_last)
        dy = torch.randn(B, Cout, Hin // 2, Hin // 2, generator=g, device=cuda).bfloat16().contiguous(memory_format=torch.channels_last)
        w = torch.zeros(Cout, Cin, 3, 3, device=cuda).bfloat16().contiguous(memory_format=torch.channels_last)
        a = kp.conv3x3_wrw(x, dy, stride=2)
        assert torch.equal(a, kp.conv3x3_wrw(x, dy, stride=2))
        ref = torch.ops.aten.convolution_backward(dy, x, w, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        assert ((a - ref.float()).norm() / ref.float().norm()).item() <= 1e-2


@pytest.mark.parametrize("B,C,S", [(16, 128, 128), (16, 256, 64), (16, 512, 32)])
def test_conv3x3_wrw_gen_resnet18_sizes_vs_miopen_and_deterministic(cuda, B, C, S):
    """layer2 / layer3 / layer4 geometry at BASELINE config 2, whole batch: against MIOpen's own weight gradient on the same
    operands (its result is rounded to bf16: 1e-2), run-to-run bit-identical, plus the oracle on a 2-image slice."""
    from torchseg_amd import kernels as K
    kp = K.provider()
    g = torch.Generator(device=cuda).manual_seed(C)
    x = torch.randn(B, C, S, S, generator=g, device=cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, C, S, S, generator=g, device=cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.zeros(C, C, 3, 3, device=cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    a = kp.conv3x3_wrw(x, dy)
    b = kp.conv3x3_wrw(x, dy)
    assert torch.equal(a, b)
    ref = torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    assert ((a - ref.float()).norm() / ref.float().norm()).item() <= 1e-2
    sl = kp.conv3x3_wrw(x[:2].contiguous(memory_format=torch.channels_last), dy[:2].contiguous(memory_format=torch.channels_last))
    want = conv_ref.conv2d_wgrad_ref(x[:2].double().cpu(), dy[:2].double().cpu(), ksize=3, stride=1, pad=1)
    assert ((sl.double().cpu() - want).norm() / want.norm()).item() <= 1e-4


def test_install_conv_wrw_covers_every_3x3_of_bisenet_but_the_stems(cuda):
    import torch.nn as nn
    from torchseg_amd.convwrw import WrwConv2d, install_conv_wrw
    from torchseg_amd.workloads.bisenet import BiSeNet
    net = BiSeNet(19, True, None, None, nn.BatchNorm2d)
    n = install_conv_wrw(net)
    want = sum(1 for m in net.modules() if isinstance(m, nn.Conv2d) and m.kernel_size == (3, 3)
               and m.in_channels % 64 == 0 and m.out_channels % 64 == 0)
    assert n == want == sum(isinstance(m, WrwConv2d) for m in net.modules()) and n == 25


@pytest.mark.parametrize("O,I", [(64, 64), (128, 64), (96, 160), (512, 512)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_weight_rot180_transpose_is_exact(cuda, O, I, dtype):
    from torchseg_amd import kernels as K
    w = torch.randn(O, I, 3, 3, device=cuda).to(dtype).contiguous(memory_format=torch.channels_last)
    got = K.provider().conv3x3_weight_rot180_t(w)
    want = w.flip(2, 3).transpose(0, 1).to(torch.bfloat16)
    assert got.shape == (I, O, 3, 3) and got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, want)


def test_dgrad_through_forward_convolution_matches_oracle(cuda):
    """dx of a stride-1 / padding-1 3x3 convolution == conv(dy, rot180(w)^T) (resnet.py:24-29 backward): against the
    fp64 data gradient of the conv oracle on the same bf16-rounded operands, and against autograd through the module."""
    import torch.nn as nn
    import torch.nn.functional as F
    from torchseg_amd import convwrw, kernels as K
    g = torch.Generator().manual_seed(3)
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.1
    dy = torch.randn(2, 64, 20, 36, generator=g)
    wr, dyr = conv_ref.bf16_round(w), conv_ref.bf16_round(dy)
    xz = torch.zeros(2, 64, 20, 36, dtype=torch.float64, requires_grad=True)
    F.conv2d(xz, wr, None, 1, 1).backward(dyr)                         # torch CPU fp64 autograd of the oracle's definition
    wb = w.to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    dyb = dy.to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    dx = F.conv2d(dyb, K.provider().conv3x3_weight_rot180_t(wb), None, 1, 1)
    rel = ((dx.double().cpu() - xz.grad).norm() / xz.grad.norm()).item()
    assert rel <= 5e-3, rel                                            # bf16 output rounding
    assert convwrw._DGRAD_FWD


def test_buffer_fetch_is_bit_equal_to_pointer_fetch(cuda, tmp_path):
    """conv3_wrw_gen_k fetches its operands by raw buffer loads (default) or by predicated pointer loads
    (TSG_CONV_WRW_BUF=0; read once per process, hence two subprocesses).  Only the addressing differs, so the fp32
    results must be identical bit for bit — bench layers, ragged sizes, stride 2, BN-on-load (tools/ab_wrw_buf.py)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for buf in ("0", "1"):
        env = dict(os.environ, TSG_CONV_WRW_BUF=buf, TSG_AB_DIR=str(tmp_path))
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "ab_wrw_buf.py")], env=env, cwd=root,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        out[buf] = r.stdout
    assert "RESULT OK" in out["1"], out["1"]
    assert out["1"].count("bit-equal") == out["0"].count("saved") >= 7, out["1"]
