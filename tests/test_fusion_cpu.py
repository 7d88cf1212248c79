"""Host logic of torchseg_amd.fusion.FuseMode on CPU: which call patterns of the unchanged network.py files are
recognised, that everything else sees ordinary tensors, and that autograd through the fused forms equals eager.
The kernels are the tests/-only stand-in provider; the HIP kernels behind the same calls are covered by -m gpu."""
import os
import sys

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def standin(monkeypatch):
    from _cpu_provider import OracleProvider
    from torchseg_amd import fusion, kernels as K
    prov = OracleProvider()
    prov.calls = []
    old = K._set_provider_for_tests(prov)
    monkeypatch.setattr(fusion, "_is_map", lambda t: isinstance(t, torch.Tensor) and t.dim() == 4
                        and t.dtype in (torch.float32, torch.bfloat16))
    monkeypatch.setattr(fusion, "_TARGET_ON_DEVICE", False)
    yield prov
    K._set_provider_for_tests(old)


def _data(C=7, ignore=-1):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, C, 6, 5, generator=g)
    y = torch.randint(0, C, (2, 6, 5), generator=g)
    y[:, :2] = ignore
    return x, y


@pytest.mark.parametrize("via_log_softmax", [False, True])
@pytest.mark.parametrize("ignore", [-1, 255])
def test_plain_ce_heads_are_routed_to_the_kernels(standin, via_log_softmax, ignore):
    """dfn train.py:48-49 (CE on logits) and pspnet network.py:50-56 (CE on log_softmax) both end in tsg_ohem_*."""
    from torchseg_amd.fusion import FuseMode
    x, y = _data(ignore=ignore)
    crit = nn.CrossEntropyLoss(reduction='mean', ignore_index=ignore)
    xr = x.clone().requires_grad_(True)
    ref = crit(F.log_softmax(xr * 1.5, dim=1) if via_log_softmax else xr * 1.5, y)
    ref.backward()
    xf = x.clone().requires_grad_(True)
    with FuseMode():
        z = xf * 1.5
        out = crit(F.log_softmax(z, dim=1) if via_log_softmax else z, y)
    out.backward()
    assert standin.calls == ["ohem_fwd", "ohem_bwd"]
    torch.testing.assert_close(out, ref, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(xf.grad, xr.grad, rtol=1e-5, atol=1e-7)


def test_log_softmax_used_by_anything_else_is_the_real_tensor(standin):
    from torchseg_amd.fusion import DeferredLogSoftmax, FuseMode, materialize
    x, _ = _data()
    xf = x.clone().requires_grad_(True)
    with FuseMode():
        z = F.log_softmax(xf * 1.0, dim=1)
        assert isinstance(z, DeferredLogSoftmax)
        e = torch.exp(z)                               # evaluator.py:263 does exp(log_softmax)
        s = z.sum()
    assert isinstance(e, torch.Tensor) and isinstance(s, torch.Tensor)
    torch.testing.assert_close(e, torch.softmax(x, 1))
    assert isinstance(materialize(z), torch.Tensor) and standin.calls == []
    with FuseMode(), torch.no_grad():
        assert isinstance(F.log_softmax(x, dim=1), torch.Tensor)       # eval path: not deferred at all


def test_unsupported_ce_arguments_fall_through_to_torch(standin):
    from torchseg_amd.fusion import FuseMode
    x, y = _data(ignore=-100)
    with FuseMode():
        a = F.cross_entropy(x, y, reduction='sum')
        b = F.cross_entropy(x, y, label_smoothing=0.1)
    assert standin.calls == []
    torch.testing.assert_close(a, F.cross_entropy(x, y, reduction='sum'))
    torch.testing.assert_close(b, F.cross_entropy(x, y, label_smoothing=0.1))


def test_iadd_then_interpolate_is_one_kernel(standin):
    """bisenet network.py:91-95: fm = arm(fm); fm += last_fm; last_fm = F.interpolate(fm, ...)."""
    from torchseg_amd.fusion import FuseMode
    g = torch.Generator().manual_seed(1)
    a0, b0 = torch.randn(2, 8, 4, 4, generator=g), torch.randn(2, 8, 4, 4, generator=g)

    def run(fused):
        a, b = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        fm = a * 2.0
        last = b * 1.0
        fm += last
        out = F.interpolate(fm, size=(8, 8), mode='bilinear', align_corners=True)
        (out * out).sum().backward()
        return out, a.grad, b.grad

    ref = run(False)
    with FuseMode():
        got = run(True)
    assert standin.calls == ["upsample_presum_fwd", "upsample_bwd"]
    for r, o in zip(ref, got):
        torch.testing.assert_close(o, r, rtol=1e-5, atol=1e-6)


def test_iadd_followed_by_anything_else_is_an_ordinary_in_place_add(standin):
    from torchseg_amd.fusion import DeferredSum, FuseMode
    a = torch.ones(1, 8, 2, 2, requires_grad=True)
    with FuseMode():
        fm = a * 2.0
        fm += torch.ones(1, 8, 2, 2)
        assert isinstance(fm, DeferredSum)
        out = F.relu(fm)                               # not interpolate -> the add happens, in place
    assert standin.calls == []
    assert torch.equal(out, torch.full((1, 8, 2, 2), 3.0))
    with FuseMode():                                   # an alias can observe the update: the add stays eager
        fm = a * 2.0
        alias = fm
        keep = [fm]
        fm += torch.ones(1, 8, 2, 2)
        assert isinstance(fm, torch.Tensor) and fm is alias and keep[0] is fm
        up = F.interpolate(fm, size=(4, 4), mode='bilinear', align_corners=True)
    assert torch.equal(alias, torch.full((1, 8, 2, 2), 3.0)) and tuple(up.shape) == (1, 8, 4, 4)
    with FuseMode():                                   # leaves / no-grad tensors are never deferred
        p = torch.zeros(1, 8, 2, 2)
        p += 1
        assert isinstance(p, torch.Tensor)


def test_iadd_interpolate_positional_arguments_and_late_modification(standin):
    """`F.interpolate(fm, (h, w), ...)` with positional size; an in-place change of an addend between the
    `+=` and its use must not change the result silently."""
    from torchseg_amd.fusion import FuseMode
    g = torch.Generator().manual_seed(2)
    a0, b0 = torch.randn(1, 8, 4, 4, generator=g), torch.randn(1, 8, 4, 4, generator=g)
    ref = F.interpolate(a0 * 2.0 + b0, (8, 8), None, 'bilinear', True)
    with FuseMode():
        fm = a0.clone().requires_grad_(True) * 2.0
        fm += b0
        out = F.interpolate(fm, (8, 8), None, 'bilinear', True)
    assert standin.calls == ["upsample_presum_fwd"]
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-6)
    from torchseg_amd import fusion
    before = dict(fusion.stats)
    with FuseMode():                                   # legal eager code must compute, not raise
        fm = a0.clone().requires_grad_(True) * 2.0
        last = b0.clone()
        fm += last
        last.mul_(0.0)                                 # the eager program had already consumed `last`: the add happens first
        out = F.interpolate(fm, size=(8, 8), mode='bilinear', align_corners=True)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-6)
    assert torch.equal(last, torch.zeros_like(last))
    assert fusion.stats["materialized_before_mutation"] == before["materialized_before_mutation"] + 1
    with FuseMode():                                   # same for the left operand and for out= / indexed writes
        fm = a0.clone().requires_grad_(True) * 2.0
        last = b0.clone()
        fm += last
        last[0, 0] = 7.0
        out = F.interpolate(fm, size=(8, 8), mode='bilinear', align_corners=True)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-6)
    with FuseMode():                                   # a change behind the mode's back is still caught, loudly
        fm = a0.clone().requires_grad_(True) * 2.0
        last = b0.clone()
        fm += last
        with torch._C.DisableTorchFunction():
            last.mul_(0.0)
        with pytest.raises(RuntimeError, match="modified in place"):
            F.interpolate(fm, size=(8, 8), mode='bilinear', align_corners=True)


def test_the_fused_iadd_path_is_taken_on_this_interpreter(standin):
    """the deferral depends on a reference-count calibration and a bytecode check and turns itself off
    silently when either fails.  On the interpreter this suite runs on it must be ON, and the counters must say so."""
    from torchseg_amd import fusion
    from torchseg_amd.fusion import DeferredSum, FuseMode
    assert fusion._IADD_BASE_REFS > 0
    before = dict(fusion.stats)
    with FuseMode():
        fm = torch.ones(1, 8, 4, 4, requires_grad=True) * 2.0
        fm += torch.ones(1, 8, 4, 4)
        assert isinstance(fm, DeferredSum)
        F.interpolate(fm, size=(8, 8), mode='bilinear', align_corners=True)
    assert fusion.stats["iadd_deferred"] == before["iadd_deferred"] + 1
    assert fusion.stats["presum_fused"] == before["presum_fused"] + 1
    assert standin.calls == ["upsample_presum_fwd"]


def test_bad_labels_are_reported_not_indexed(standin, monkeypatch):
    from torchseg_amd import losses
    x, y = _data(C=7, ignore=255)
    y[0, 3, 3] = 9                                     # neither ignore nor a class
    loss, sel = losses.cross_entropy_2d(x, y, ignore_index=255, return_selection=True)
    assert int(sel[5]) == 1 and torch.isfinite(loss)
    with pytest.raises(Exception, match="neither ignore_label nor a class"):
        losses.check_labels(sel)


@pytest.mark.parametrize("via_log_softmax", [False, True])
def test_head_upsampling_stays_pending_for_the_criterion(standin, via_log_softmax):
    """pspnet network.py:46-56 / bisenet network.py:164-166 + train.py's criterion: F.interpolate(logits, x8) ->
    [F.log_softmax ->] CrossEntropyLoss runs as ONE fused call (tsg_ohem_up_*), the full-resolution logits are never
    produced; gradients equal eager's."""
    from torchseg_amd.fusion import DeferredLogSoftmax, FuseMode
    from torchseg_amd.upsample import DeferredUpsample
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 7, 4, 5, generator=g)
    y = torch.randint(0, 7, (2, 32, 40), generator=g)
    y[:, :3] = 255
    crit = nn.CrossEntropyLoss(reduction='mean', ignore_index=255)

    def head(t):
        fm = F.interpolate(t * 1.5, scale_factor=8, mode='bilinear', align_corners=True)
        return F.log_softmax(fm, dim=1) if via_log_softmax else fm

    xr = x.clone().requires_grad_(True)
    ref = crit(head(xr), y)
    ref.backward()
    xf = x.clone().requires_grad_(True)
    with FuseMode(head=True):
        fm = head(xf)
        assert isinstance(fm, DeferredLogSoftmax if via_log_softmax else DeferredUpsample)
        out = crit(fm, y)
    out.backward()
    assert standin.calls == ["ohem_up_fwd", "ohem_up_bwd"]
    torch.testing.assert_close(out, ref, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(xf.grad, xr.grad, rtol=1e-5, atol=1e-7)


def test_pending_upsampling_is_a_real_tensor_for_everything_else(standin):
    from torchseg_amd.fusion import FuseMode, materialize
    from torchseg_amd.upsample import DeferredUpsample
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 5, 3, 3, generator=g)
    ref = F.interpolate(x, scale_factor=4, mode='bilinear', align_corners=True)
    xf = x.clone().requires_grad_(True)
    with FuseMode(head=True):
        up = F.interpolate(xf, scale_factor=4, mode='bilinear', align_corners=True)
        assert isinstance(up, DeferredUpsample)
        assert isinstance(F.interpolate(xf, scale_factor=2, mode='bilinear', align_corners=True), torch.Tensor)   # < 4
        assert isinstance(F.interpolate(xf, scale_factor=4, mode='nearest'), torch.Tensor)
        wide = torch.randn(1, 64, 3, 3, requires_grad=True)
        assert isinstance(F.interpolate(wide, scale_factor=4, mode='bilinear', align_corners=True), torch.Tensor)  # features
        s = torch.sigmoid(up)                          # e.g. a focal-loss head or an evaluator
        m = up.argmax(1)
    assert isinstance(s, torch.Tensor) and isinstance(m, torch.Tensor)
    torch.testing.assert_close(s, torch.sigmoid(ref))
    out = materialize(up)
    assert isinstance(out, torch.Tensor)
    torch.testing.assert_close(out, ref)
    with FuseMode(head=True), torch.no_grad():         # eval path: nothing is deferred
        assert isinstance(F.interpolate(x, scale_factor=4, mode='bilinear', align_corners=True), torch.Tensor)


def test_add_underscore_call_is_never_deferred(standin):
    """`fm.add_(x)` dispatches to the same torch function as `fm += x` but drops the returned object: deferring it
    would lose the update."""
    from torchseg_amd.fusion import FuseMode
    a = torch.ones(1, 8, 2, 2, requires_grad=True)
    with FuseMode():
        fm = a * 2.0
        fm.add_(torch.ones(1, 8, 2, 2))
        assert isinstance(fm, torch.Tensor) and torch.equal(fm, torch.full((1, 8, 2, 2), 3.0))


# ---- consecutive ConvBnRelu modules (bisenet network.py:131-137) ----------------------------------------------------

def _spatial_path(seed=0):
    from torchseg_amd.workloads import ensure_furnace_on_path
    ensure_furnace_on_path()
    from seg_opr.seg_oprs import ConvBnRelu
    from torchseg_amd.syncbn import SyncBatchNorm
    torch.manual_seed(seed)

    class SpatialPath(nn.Module):                      # the statements of network.py:131-137, literally
        def __init__(self):
            super().__init__()
            kw = dict(has_bn=True, norm_layer=SyncBatchNorm, has_relu=True, has_bias=False)
            self.conv_7x7 = ConvBnRelu(3, 64, 7, 2, 3, **kw)
            self.conv_3x3_1 = ConvBnRelu(64, 64, 3, 2, 1, **kw)
            self.conv_3x3_2 = ConvBnRelu(64, 64, 3, 2, 1, **kw)
            self.conv_1x1 = ConvBnRelu(64, 128, 1, 1, 0, **kw)

        def forward(self, x):
            x = self.conv_7x7(x)
            x = self.conv_3x3_1(x)
            x = self.conv_3x3_2(x)
            output = self.conv_1x1(x)
            return output
    return SpatialPath()


def test_consecutive_conv_bn_relu_modules_hand_their_batchnorm_on(standin, monkeypatch):
    from torchseg_amd import fusion
    monkeypatch.setattr(fusion, "_on_device", lambda t: True)
    ref, net = _spatial_path(), _spatial_path()
    net.load_state_dict(ref.state_dict())
    x = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(1))
    want = ref(x)
    want.square().mean().backward()
    before = dict(fusion.stats)
    with fusion.FuseMode(loss=False, add_up=False, chain=True):
        got = net(x)
        assert isinstance(got, torch.Tensor)           # 128 output channels: nothing downstream could take its BatchNorm
    got.square().mean().backward()
    d = {k: fusion.stats[k] - before[k] for k in fusion.stats}
    assert (d["cbr_deferred"], d["cbr_fed"], d["cbr_materialized"]) == (3, 3, 0), d
    torch.testing.assert_close(got, want, rtol=0, atol=0)
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        torch.testing.assert_close(p.grad, q.grad, rtol=0, atol=0, msg=n)
    for (n, b), (_, c) in zip(net.named_buffers(), ref.named_buffers()):
        torch.testing.assert_close(b, c, rtol=0, atol=0, msg=n)   # every BatchNorm updated its running statistics once
    assert fusion.CHAIN_ACTIVE is False


def test_pending_conv_bn_relu_is_a_real_tensor_for_everything_else(standin, monkeypatch):
    from torchseg_amd import fusion
    monkeypatch.setattr(fusion, "_on_device", lambda t: True)
    ref, net = _spatial_path(), _spatial_path()
    net.load_state_dict(ref.state_dict())
    x = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(1))
    pool = nn.AvgPool2d(2)
    a = ref.conv_7x7(x)
    want = pool(a) .sum() + a.mean() + float(a.size(1))
    with fusion.FuseMode(loss=False, add_up=False, chain=True):
        p = net.conv_7x7(x)
        assert isinstance(p, fusion.PendingCbr)
        got = pool(p).sum() + p.mean() + float(p.size(1))        # a module, a torch function, an attribute
        dead = net.conv_3x3_1(p)                                 # consumed by the next ConvBnRelu AFTER it was materialised
        assert isinstance(dead, fusion.PendingCbr)               # ... and never used: its BatchNorm still has to run
    torch.testing.assert_close(got, want, rtol=0, atol=0)
    ref.conv_3x3_1(a)
    torch.testing.assert_close(net.conv_3x3_1.bn.running_mean, ref.conv_3x3_1.bn.running_mean, rtol=0, atol=0)
    assert int(net.conv_3x3_1.bn.num_batches_tracked) == 1


def test_a_conv_bn_relu_nested_in_one_of_our_modules_does_not_defer(standin, monkeypatch):
    """chains are a pattern of network.py level.  A 64-output ConvBnRelu called INSIDE another forward that runs
    with torch-function dispatch disabled (fusion.outside_mode: AttentionRefinement, FeatureFusion, ResNet, the criteria)
    returns a real tensor there — a PendingCbr would reach code that cannot see it for what it is."""
    from torchseg_amd import fusion
    monkeypatch.setattr(fusion, "_on_device", lambda t: True)
    net = _spatial_path()
    seen = {}

    class Outer(nn.Module):
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        @fusion.outside_mode
        def forward(self, x):
            y = self.inner(x)
            seen["type"] = type(y)
            seen["chain"] = fusion.CHAIN_ACTIVE
            return y * 2.0

    outer = Outer(net.conv_7x7)
    x = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(1))
    before = dict(fusion.stats)
    with fusion.FuseMode(loss=False, add_up=False, chain=True) as mode:
        got = outer(x)
        assert fusion.CHAIN_ACTIVE is mode                       # restored for network.py level
        top = net.conv_7x7(x)
        assert isinstance(top, fusion.PendingCbr)                # ... where the same module still defers
    assert seen["type"] is torch.Tensor and seen["chain"] is False
    assert fusion.stats["cbr_deferred"] - before["cbr_deferred"] == 1
    torch.testing.assert_close(got, net.conv_7x7(x) * 2.0, rtol=0, atol=0)


def test_a_conv_bn_relu_output_consumed_twice_is_reported(standin, monkeypatch):
    from torchseg_amd import fusion
    monkeypatch.setattr(fusion, "_on_device", lambda t: True)
    net = _spatial_path()
    x = torch.randn(2, 3, 32, 32)
    with pytest.raises(RuntimeError, match="TSG_FUSE_CHAIN=0"):
        with fusion.FuseMode(loss=False, add_up=False, chain=True):
            p = net.conv_7x7(x)
            net.conv_3x3_1(p)
            p.sum()                                              # the first BatchNorm already ran inside the fused pair


def test_inplace_keyword_settles_a_pending_sum(standin):
    """F.relu(x, inplace=True) dispatches as 'relu'; the pending `a += b` must happen before it."""
    from torchseg_amd.fusion import FuseMode
    a0 = torch.randn(1, 4, 3, 3)
    b0 = torch.randn(1, 4, 3, 3)

    def run(mode):
        a = (a0 * 1.0).requires_grad_(True) * 1.0
        b = (b0 * 1.0).requires_grad_(True) * 1.0
        with mode:
            a += b
            F.relu(b, inplace=True)
            return a * 1.0
    import contextlib
    torch.testing.assert_close(run(FuseMode(loss=False)), run(contextlib.nullcontext()), rtol=0, atol=0)


def test_outside_mode_materialises_pending_keyword_arguments():
    """The mode's global forward pre-hook sees positional arguments only: a pending value handed to one of our modules by
    keyword is materialised by the `outside_mode` wrapper before the forward leaves torch-function dispatch."""
    import torch
    from torchseg_amd import fusion

    seen = {}

    @fusion.outside_mode
    def forward(x, residual=None):
        seen["type"] = type(residual)
        return x + residual

    a, b = torch.ones(1, 2, 2, 2), torch.full((1, 2, 2, 2), 2.0)
    pending = fusion.DeferredSum(a, b)
    with fusion.FuseMode():
        out = forward(torch.zeros(1, 2, 2, 2), residual=pending)
    assert seen["type"] is torch.Tensor and torch.equal(out, torch.full((1, 2, 2, 2), 3.0))
