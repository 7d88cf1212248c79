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


@pytest.mark.parametrize("variant", ["tr", "v1"])
@pytest.mark.parametrize("shape", [(1, 8, 32), (2, 6, 40), (1, 3, 5), (3, 17, 70), (2, 64, 64)])
def test_conv3x3_wrw_vs_oracle(cuda, shape, variant):
    _check(cuda, *shape, variant=variant)


def test_conv3x3_wrw_layer1_size_and_determinism(cuda):
    """ResNet-18 layer1 geometry at BASELINE config 2 (256 x 256 maps; B = 4 keeps the fp64 oracle to seconds)."""
    for variant in ("tr", "v1"):
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
