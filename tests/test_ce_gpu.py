"""GPU parity of the plain cross-entropy heads on the OHEM kernels in plain-CE mode, against
what the reference calls: nn.CrossEntropyLoss(reduction='mean', ignore_index=...) on the CPU (dfn train.py:48-49 /
network.py:140-143 on logits; pspnet & psanet network.py:50-56 on F.log_softmax output, ignore -1).
Bars: fp32 loss within 1e-4 (north_star), gradient within 1e-4 of the largest gradient entry; bf16 logits are
compared against the CPU criterion fed the same bf16-rounded values."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (B, C, H, W, ignore): dfn head at the BASELINE configs[3] crop; pspnet/psanet head at the configs[2]/[4] crop (480)
FULL = [(2, 19, 1024, 1024, 255), (2, 150, 480, 480, -1)]
SMALL = [(3, 19, 37, 53, 255), (2, 150, 24, 24, -1), (1, 5, 8, 8, -100), (2, 2, 16, 40, 255)]


def _case(B, C, H, W, ignore, seed=0):
    g = torch.Generator().manual_seed(seed + C + H)
    x = torch.randn(B, C, H, W, generator=g) * 3.0
    y = torch.randint(0, C, (B, H, W), generator=g)
    y[:, : max(1, H // 16)] = ignore
    y[0, -1, ::3] = ignore
    return x, y


def _cpu(x, y, ignore, weight=None, via_log_softmax=False):
    xr = x.clone().requires_grad_(True)
    crit = nn.CrossEntropyLoss(weight=weight, reduction='mean', ignore_index=ignore)
    loss = crit(F.log_softmax(xr, dim=1) if via_log_softmax else xr, y)
    loss.backward()
    return loss.item(), xr.grad


@pytest.mark.parametrize("shape", FULL + SMALL)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_cross_entropy_matches_reference_criterion(cuda, shape, dtype):
    from torchseg_amd.losses import cross_entropy_2d
    B, C, H, W, ignore = shape
    x, y = _case(*shape)
    x = x.to(dtype).float()                                  # both sides see the same (possibly bf16-rounded) values
    ref_loss, ref_grad = _cpu(x, y, ignore)
    xd = x.to(cuda).to(dtype).requires_grad_(True)
    loss, sel = cross_entropy_2d(xd, y.to(cuda), ignore_index=ignore, return_selection=True)
    (loss * 2.0).backward()
    sel = sel.cpu()
    assert int(sel[3]) == 2 and int(sel[5]) == 0             # plain-CE branch, no out-of-range labels
    assert int(sel[1]) == int(sel[2]) == int((y != ignore).sum())
    assert abs(loss.item() - ref_loss) <= 1e-4 * max(1.0, abs(ref_loss)), (loss.item(), ref_loss)
    got = xd.grad.float().cpu() / 2.0
    tol = 1e-4 if dtype == torch.float32 else 2 ** -8
    assert (got - ref_grad).abs().max().item() <= tol * ref_grad.abs().max().item()
    assert torch.equal(got[:, :, : max(1, H // 16)], torch.zeros_like(got[:, :, : max(1, H // 16)]))   # ignored rows


def test_class_weights_all_ignored_and_uint8_labels(cuda):
    from torchseg_amd.losses import CrossEntropyLoss2d, cross_entropy_2d
    x, y = _case(2, 19, 32, 48, 255)
    w = torch.rand(19) + 0.5
    ref_loss, ref_grad = _cpu(x, y, 255, weight=w)
    crit = CrossEntropyLoss2d(weight=w, ignore_index=255).to(cuda)
    xd = x.to(cuda).requires_grad_(True)
    loss = crit(xd, y.to(cuda).to(torch.uint8))
    loss.backward()
    assert abs(loss.item() - ref_loss) <= 1e-4 * max(1.0, abs(ref_loss))
    assert (xd.grad.cpu() - ref_grad).abs().max().item() <= 1e-4 * ref_grad.abs().max().item()
    none = torch.full_like(y, 255)
    assert torch.isnan(cross_entropy_2d(x.to(cuda), none.to(cuda), ignore_index=255))          # like torch: 0 / 0
    assert torch.isnan(F.cross_entropy(x, none, ignore_index=255))


def test_out_of_range_labels_are_counted_and_dropped(cuda):
    from torchseg_amd.losses import check_labels, cross_entropy_2d
    x, y = _case(2, 19, 16, 16, 255)
    ref_loss, _ = _cpu(x, y, 255)
    bad = y.clone()
    bad[1, 5, 5] = 19
    bad[1, 6, 6] = -7
    ref_loss_bad, _ = _cpu(x, torch.where((bad < 0) | (bad >= 19), torch.full_like(bad, 255), bad), 255)
    loss, sel = cross_entropy_2d(x.to(cuda), bad.to(cuda), ignore_index=255, return_selection=True)
    assert int(sel[5].item()) == 2
    assert abs(loss.item() - ref_loss_bad) <= 1e-4 * max(1.0, abs(ref_loss_bad)) and ref_loss != ref_loss_bad
    with pytest.raises(Exception, match="neither ignore_label nor a class"):
        check_labels(sel)


class _RefStyleHead(nn.Module):
    """The statements of pspnet network.py:46-57 / dfn network.py:140-143 around a tiny body, as an UNCHANGED
    network.py writes them: F.interpolate -> F.log_softmax -> self.criterion (an nn.CrossEntropyLoss)."""

    def __init__(self, C, criterion, log_softmax):
        super().__init__()
        self.conv = nn.Conv2d(8, C, 1)
        self.criterion = criterion
        self.log_softmax = log_softmax

    def forward(self, data, label=None):
        fm = F.interpolate(self.conv(data), scale_factor=8, mode='bilinear', align_corners=True)
        if self.log_softmax:
            fm = F.log_softmax(fm, dim=1)
        if label is not None:
            return self.criterion(fm, label)
        return fm


@pytest.mark.parametrize("log_softmax,ignore,C", [(True, -1, 150), (False, 255, 19)])
def test_unchanged_call_pattern_runs_on_the_hip_kernels(cuda, log_softmax, ignore, C):
    """nn.CrossEntropyLoss built by an unchanged train.py, called by an unchanged network.py, under our DDP wrapper:
    tsg_ohem_fwd/bwd must run (no aten log_softmax / nll_loss) — for <= 32 classes their fused-upsample forms, so
    that not even the full-resolution logits exist — and the numbers must equal the CPU's."""
    from torchseg_amd import kernels as K
    from torchseg_amd.ddp import DistributedDataParallel
    torch.manual_seed(4)
    ref = _RefStyleHead(C, nn.CrossEntropyLoss(reduction='mean', ignore_index=ignore), log_softmax)
    net = _RefStyleHead(C, nn.CrossEntropyLoss(reduction='mean', ignore_index=ignore), log_softmax)
    net.load_state_dict(ref.state_dict())
    net = DistributedDataParallel(net.to(cuda), compute_dtype=torch.float32)
    assert net.fuse_loss
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 8, 12, 12, generator=g)
    y = torch.randint(0, C, (2, 96, 96), generator=g)
    y[:, :5] = ignore
    ref_loss = ref(x, y)
    ref_loss.backward()
    kp = K.provider()
    calls = []
    orig_f, orig_b, orig_uf, orig_ub = kp.ohem_fwd, kp.ohem_bwd, kp.ohem_up_fwd, kp.ohem_up_bwd
    kp.ohem_fwd = lambda *a, **k: (calls.append("fwd"), orig_f(*a, **k))[1]
    kp.ohem_bwd = lambda *a, **k: (calls.append("bwd"), orig_b(*a, **k))[1]
    kp.ohem_up_fwd = lambda *a, **k: (calls.append("up_fwd"), orig_uf(*a, **k))[1]
    kp.ohem_up_bwd = lambda *a, **k: (calls.append("up_bwd"), orig_ub(*a, **k))[1]
    try:
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
            loss = net(x.to(cuda), y.to(cuda))
            loss.backward()
    finally:
        del kp.ohem_fwd, kp.ohem_bwd, kp.ohem_up_fwd, kp.ohem_up_bwd
    assert calls == (["up_fwd", "up_bwd"] if C <= 32 else ["fwd", "bwd"])
    ops = {e.key for e in prof.key_averages()}
    if C <= 32:
        assert not any("upsample_bilinear2d" in o for o in ops), ops
    assert not ({"aten::_log_softmax", "aten::nll_loss2d_forward", "aten::nll_loss_nd"} & ops), ops
    assert abs(loss.item() - ref_loss.item()) <= 1e-4 * max(1.0, abs(ref_loss.item()))
    for (n, p), (_, q) in zip(net.module.named_parameters(), ref.named_parameters()):
        assert (p.grad.cpu() - q.grad).abs().max().item() <= 1e-4 * max(q.grad.abs().max().item(), 1e-6), n
    net.eval()
    with torch.no_grad():                                   # eval path returns a real tensor (eval.py / evaluator.py)
        out = net(x.to(cuda))
    assert isinstance(out, torch.Tensor)
    torch.testing.assert_close(out.cpu(), ref.eval()(x).detach(), rtol=1e-4, atol=1e-4)
