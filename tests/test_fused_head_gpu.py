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


# ---------------------------------------------------------------------------------------------------------------------
# The instantiation bench.py times: ohem_up_fwd_k<bf16, uint8 labels, 20> / ohem_up_bwd_k<bf16, 1, 20,
# 1024> at 16 x 19 x 128^2 -> 1024^2 (the two x8 heads of bisenet network.py:104-106,160-168) and 16 x 19 x 64^2 -> 1024^2
# (the x16 head), min_kept = 16 * 1024^2 / 16 (train.py:48-49), thresh 0.7 — against oracle.ohem_ref (loss_opr.py:68-98)
# applied to F.interpolate of the SAME bf16-rounded logits, both threshold branches.
def _bench_head_case(IH, regime, seed):
    B, C, S = 16, 19, 1024
    g = torch.Generator().manual_seed(seed)
    if regime == "random":                                  # what a randomly initialised network emits: p_t ~ 1 / 19
        z = torch.randn(B, C, IH, IH, generator=g)
        t = torch.randint(0, C, (B, S, S), generator=g)
    else:                                                   # trained-like: labels constant over quadrants, 2 % re-drawn
        quad = torch.randint(0, C, (B, 2, 2), generator=g)
        lab_lo = quad.repeat_interleave(IH // 2, 1).repeat_interleave(IH // 2, 2)
        t = lab_lo.repeat_interleave(S // IH, 1).repeat_interleave(S // IH, 2)
        flip = torch.rand(t.shape, generator=g) < 0.02
        t = torch.where(flip, torch.randint(0, C, t.shape, generator=g), t)
        z = torch.randn(B, C, IH, IH, generator=g) + 8.0 * F.one_hot(lab_lo, C).permute(0, 3, 1, 2).float()
    t[:, :8] = 255
    return z.to(torch.bfloat16).contiguous(), t


@pytest.mark.parametrize("IH,regime", [(128, "random"), (128, "trained"), (64, "random"), (64, "trained")])
def test_benched_head_bf16_uint8_labels_vs_oracle(cuda, IH, regime):
    import os
    from torchseg_amd import kernels as K
    from torchseg_amd.losses import ohem_cross_entropy
    from torchseg_amd.upsample import DeferredUpsample
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    kp = K.provider()
    B, C, S = 16, 19, 1024
    k = B * S * S // 16
    zb, t = _bench_head_case(IH, regime, seed=IH + len(regime))
    # ---- oracle: the reference's statement on the fp32 bilinear interpolation of the same bf16 values
    zr = zb.float().requires_grad_(True)
    logits = F.interpolate(zr, size=(S, S), mode="bilinear", align_corners=True)
    ref_loss, info = ohem_ref.ohem_cross_entropy(logits, t, 255, 0.7, k, None, return_info=True)
    ref_loss.backward()
    del logits
    assert info["branch"] == (1 if regime == "trained" else 0)
    # ---- device, kernel level: the forward's own probabilities
    zd = zb.to(cuda)
    t8 = t.to(torch.uint8).to(cuda)
    assert kp.ohem_up_supported(zd, S, S, 0.7)
    loss_k, nll, lse, sel = kp.ohem_up_fwd(zd, t8, S, S, 255, 0.7, k, None)
    sel = sel.cpu()
    assert int(sel[3]) == info["branch"] and int(sel[2]) == info["num_valid"]
    p_dev = kp.ohem_target_prob(nll, t8, C, 255).cpu()
    thr_dev = sel[0:1].view(torch.float32).item()
    valid = t.view(-1) != 255
    if info["branch"] == 1:     # bit-exact top-k given the device's probabilities: torch.sort(p)[k - 1], loss_opr.py:86-89
        assert int(sel[0]) == torch.sort(p_dev)[0][k - 1].view(torch.int32).item()
    else:
        assert thr_dev == np.float32(0.7)
    kept_dev = valid & (p_dev <= thr_dev)
    assert int(sel[1]) == int(kept_dev.sum())
    # ---- against the oracle: the two sides' probabilities differ by the rounding of two fp32 interpolation orders and two
    # expf implementations; membership may differ only inside the band that MEASURED difference implies
    mp = info["mask_prob"]
    thr = info["threshold"]
    perr = ((p_dev - mp).abs()[valid] / mp[valid].clamp_min(1e-30)).max().item()
    assert perr <= 2e-5, perr                               # fp32 interpolation + softmax on |z| <= ~12
    band = 2.0 * perr * thr + 1e-7
    assert abs(thr_dev - thr) <= band, (thr_dev, thr, perr)
    near = (mp - thr).abs() <= band
    kept_ref = info["kept"].view(-1)
    assert torch.equal(kept_dev[~near], kept_ref[~near])
    n_near = int((near & valid).sum())
    assert n_near <= 4096 and abs(int(sel[1]) - info["n_kept"]) <= n_near, (n_near, int(sel[1]), info["n_kept"])
    assert abs(loss_k.item() - ref_loss.item()) <= 1e-4 * max(1.0, abs(ref_loss.item()))
    # ---- the autograd path the step takes: criterion(DeferredUpsample(z)), loss and dz (bf16)
    za = zd.clone().requires_grad_(True)
    loss, sel2 = ohem_cross_entropy(DeferredUpsample(za, (S, S)), t8, 255, 0.7, k, None, return_selection=True)
    loss.backward()
    assert torch.equal(sel2.cpu()[:4], sel[:4]) and loss.item() == loss_k.item()
    gref = zr.grad
    err = (za.grad.float().cpu() - gref).abs()
    scale = gref.abs().max().item()
    bad = err > gref.abs() * 2.0 ** -7 + 1e-3 * scale        # one bf16 rounding of dz + fp32 accumulation order
    # a pixel whose membership differs reaches 4 source pixels x 19 classes of dz
    assert int(bad.sum()) <= 4 * 19 * n_near, (int(bad.sum()), n_near, err.max().item(), scale)
    print("head %dx19x%d^2->1024^2 bf16/u8 %s: branch %d, threshold %.9g (oracle %.9g), kept %d (oracle %d), p rel err %.2e, "
          "%d pixels in the band, loss %.6f (oracle %.6f), max |ddz| %.3e of %.3e"
          % (B, IH, regime, info["branch"], thr_dev, thr, int(sel[1]), info["n_kept"], perr, n_near, loss.item(),
             ref_loss.item(), err.max().item(), scale))
