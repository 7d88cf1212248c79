"""GPU parity of the fused upsample+OHEM head: criterion applied to a
deferred bilinear upsample == criterion(F.interpolate(z)) of the reference, in loss, kept
set and gradient w.r.t. the low-resolution logits."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ohem_ref

pytestmark = pytest.mark.gpu

CASES = [  # B, C, IH, IW, OH, OW, regime, min_kept fraction, thresh
    (2, 19, 16, 16, 128, 128, "random", 1 / 16, 0.7),
    (2, 19, 16, 16, 128, 128, "confident", 1 / 2, 0.7),
    (2, 19, 8, 8, 128, 128, "confident", 1 / 4, 0.7),
    (2, 19, 12, 10, 96, 80, "confident", 1 / 3, 0.7),
    (2, 5, 9, 7, 36, 28, "confident", 1 / 2, 0.6),
    (1, 19, 64, 64, 512, 512, "confident", 1 / 16, 0.7),
    (2, 150, 8, 8, 64, 64, "random", 1 / 8, 0.7),
]


def _make(B, C, IH, IW, OH, OW, regime, seed):
    g = torch.Generator().manual_seed(seed)
    lab_lo = torch.randint(0, C, (B, IH, IW), generator=g)
    t = F.interpolate(lab_lo[:, None].float(), size=(OH, OW), mode="nearest")[:, 0].long()
    flip = torch.rand(t.shape, generator=g) < 0.1
    t[flip] = torch.randint(0, C, (int(flip.sum()),), generator=g)
    t[:, : max(1, OH // 16)] = 255
    z = torch.randn(B, C, IH, IW, generator=g)
    if regime == "confident":
        z = z + 8.0 * F.one_hot(lab_lo, C).permute(0, 3, 1, 2).float()
    return z.contiguous(), t


@pytest.mark.parametrize("case", CASES)
def test_fused_head_vs_oracle_fp32(cuda, case):
    from torchseg_amd.losses import ohem_cross_entropy
    from torchseg_amd.upsample import DeferredUpsample
    B, C, IH, IW, OH, OW, regime, frac, thresh = case
    z, t = _make(B, C, IH, IW, OH, OW, regime, seed=IH * 7 + OW)
    min_kept = int(B * OH * OW * frac)
    zr = z.clone().requires_grad_(True)
    logits = F.interpolate(zr, size=(OH, OW), mode="bilinear", align_corners=True)
    ref_loss, info = ohem_ref.ohem_cross_entropy(logits, t, 255, thresh, min_kept, None, return_info=True)
    ref_loss.backward()
    zd = z.to(cuda).requires_grad_(True)
    loss, sel = ohem_cross_entropy(DeferredUpsample(zd, (OH, OW)), t.to(cuda), 255, thresh, min_kept, None,
                                   return_selection=True)
    loss.backward()
    sel = sel.cpu()
    assert int(sel[3]) == info["branch"]
    assert int(sel[2]) == info["num_valid"]
    assert abs(loss.item() - ref_loss.item()) <= 1e-4 * max(1.0, abs(ref_loss.item()))
    # membership may flip only for pixels within ~2 ulp of the threshold
    if info["mask_prob"] is not None:
        near = int((np.abs(info["mask_prob"].numpy() - info["threshold"]) <= 4e-7 * info["threshold"]).sum())
    else:
        near = 0
    assert abs(int(sel[1]) - info["n_kept"]) <= near
    gscale = zr.grad.abs().max().item()
    err = (zd.grad.cpu() - zr.grad).abs().max().item()
    assert err <= 2e-4 * gscale + (1e-3 * gscale if near else 0.0), (err, gscale, near)


def test_fused_equals_unfused_hip_path(cuda):
    """Same inputs through the two HIP paths: deferred (fused) vs materialised upsample."""
    from torchseg_amd.losses import ohem_cross_entropy
    from torchseg_amd.upsample import DeferredUpsample, upsample_bilinear_ac
    z, t = _make(4, 19, 32, 32, 256, 256, "confident", seed=9)
    k = 4 * 256 * 256 // 16
    for dtype, tol in ((torch.float32, 1e-5), (torch.bfloat16, 2e-2)):
        za = z.to(cuda).to(dtype).requires_grad_(True)
        zb = z.to(cuda).to(dtype).requires_grad_(True)
        la = ohem_cross_entropy(DeferredUpsample(za, (256, 256)), t.to(cuda), 255, 0.7, k)
        lb = ohem_cross_entropy(upsample_bilinear_ac(zb, size=(256, 256)), t.to(cuda), 255, 0.7, k)
        la.backward(); lb.backward()
        assert abs(la.item() - lb.item()) <= max(tol, 1e-5) * max(1.0, abs(lb.item())) * (50 if dtype == torch.bfloat16 else 1)
        scale = zb.grad.float().abs().max().item()
        assert (za.grad.float() - zb.grad.float()).abs().max().item() <= (tol * 5) * scale


def test_deferred_interpolate_is_transparent(cuda):
    """After install, F.interpolate(x>=4) yields a DeferredUpsample; a non-criterion consumer
    (log_softmax, arithmetic, indexing) sees exactly the materialised tensor, with autograd."""
    from torchseg_amd import kernels as K
    from torchseg_amd.losses import ProbOhemCrossEntropy2d
    from torchseg_amd.upsample import (DeferredUpsample, install_aten_overrides, install_deferred_interpolate,
                                       uninstall_deferred_interpolate)
    install_aten_overrides()
    install_deferred_interpolate()
    try:
        _transparent_body(cuda, DeferredUpsample, K, ProbOhemCrossEntropy2d)
    finally:
        uninstall_deferred_interpolate()
    assert not isinstance(F.interpolate(torch.randn(1, 3, 4, 4, device=cuda), scale_factor=8, mode="bilinear",
                                        align_corners=True), DeferredUpsample)


def _transparent_body(cuda, DeferredUpsample, K, ProbOhemCrossEntropy2d):
    x = torch.randn(2, 19, 8, 8, device=cuda, requires_grad=True)
    y = F.interpolate(x, scale_factor=8, mode="bilinear", align_corners=True)
    assert isinstance(y, DeferredUpsample) and tuple(y.shape) == (2, 19, 64, 64)
    small = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    assert isinstance(small, torch.Tensor)
    ref = F.interpolate(x.detach().cpu(), scale_factor=8, mode="bilinear", align_corners=True)
    out = F.log_softmax(y, dim=1)
    torch.testing.assert_close(out.detach().cpu(), F.log_softmax(ref, dim=1), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close((y * 2 + 1)[0].detach().cpu(), (ref * 2 + 1)[0], rtol=1e-5, atol=1e-5)
    out.sum().backward()
    assert x.grad is not None
    # criterion consumer: fused kernel runs, no full-resolution upsample is launched
    kp = K.provider()
    calls = {"up": 0, "fused": 0}
    o1, o2 = kp.upsample_fwd, kp.ohem_up_fwd
    kp.upsample_fwd = lambda *a: (calls.__setitem__("up", calls["up"] + 1), o1(*a))[1]
    kp.ohem_up_fwd = lambda *a: (calls.__setitem__("fused", calls["fused"] + 1), o2(*a))[1]
    try:
        t = torch.randint(0, 19, (2, 64, 64), device=cuda)
        crit = ProbOhemCrossEntropy2d(255, thresh=0.7, min_kept=2 * 64 * 64 // 16)
        loss = crit(F.interpolate(x, scale_factor=8, mode="bilinear", align_corners=True), t)
        loss.backward()
    finally:
        kp.upsample_fwd, kp.ohem_up_fwd = o1, o2
    assert calls == {"up": 0, "fused": 1}
