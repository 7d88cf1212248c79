"""CPU: pin the oracle restatements against (a) golden vectors produced by the
reference's own Python (tests/golden/make_golden.py) and (b) torch CPU ops."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import focal_ref, ohem_ref, upsample_ref

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _cases(npz):
    return sorted({k.split("/")[0] for k in npz.files})


def test_ohem_oracle_matches_reference_golden():
    z = np.load(os.path.join(GOLD, "ohem_golden.npz"))
    assert len(_cases(z)) >= 8
    for name in _cases(z):
        pred = torch.from_numpy(z[name + "/pred"]).requires_grad_(True)
        target = torch.from_numpy(z[name + "/target"].astype(np.int64))
        thresh, min_kept, use_w = z[name + "/cfg"]
        w = torch.tensor(ohem_ref.CITYSCAPES_WEIGHT) if use_w else None
        loss, info = ohem_ref.ohem_cross_entropy(pred, target, 255, float(thresh), int(min_kept), w, return_info=True)
        ref_loss = float(z[name + "/loss"])
        if np.isnan(ref_loss):
            assert torch.isnan(loss), name
            continue
        assert loss.item() == pytest.approx(ref_loss, rel=0, abs=0), name   # same ops => bit-equal
        loss.backward()
        np.testing.assert_array_equal(pred.grad.numpy(), z[name + "/grad"], err_msg=name)
        kept_ref = np.abs(z[name + "/grad"]).sum(1) > 0
        np.testing.assert_array_equal(info["kept"].numpy(), kept_ref, err_msg=name)


def test_focal_oracle_matches_reference_golden():
    z = np.load(os.path.join(GOLD, "focal_golden.npz"))
    for name in _cases(z):
        pred = torch.from_numpy(z[name + "/pred"]).requires_grad_(True)
        target = torch.from_numpy(z[name + "/target"].astype(np.int64))
        gamma, alpha = z[name + "/cfg"]
        loss = focal_ref.sigmoid_focal_loss(pred, target, 255, float(gamma), float(alpha))
        assert loss.item() == pytest.approx(float(z[name + "/loss"]), rel=1e-7)
        loss.backward()
        np.testing.assert_allclose(pred.grad.numpy(), z[name + "/grad"], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("ih,iw,oh,ow", [(1, 1, 32, 32), (32, 32, 64, 64), (8, 8, 64, 64), (7, 5, 13, 9),
                                         (16, 12, 128, 96), (6, 6, 6, 6), (9, 9, 4, 3), (3, 4, 1, 1)])
def test_upsample_oracle_matches_torch_cpu(ih, iw, oh, ow):
    g = torch.Generator().manual_seed(ih * 100 + ow)
    x = torch.randn(2, 3, ih, iw, generator=g, dtype=torch.float32, requires_grad=True)
    y = F.interpolate(x, size=(oh, ow), mode="bilinear", align_corners=True)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float32)
    y.backward(dy)
    np.testing.assert_allclose(upsample_ref.upsample_bilinear_ac(x.detach().numpy(), oh, ow), y.detach().numpy(),
                               rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(upsample_ref.upsample_bilinear_ac_backward(dy.numpy(), ih, iw), x.grad.numpy(),
                               rtol=1e-4, atol=1e-4)


def test_nearest_oracle_matches_torch_cpu():
    x = torch.arange(2 * 20 * 30, dtype=torch.float32).reshape(1, 2, 20, 30)
    for oh, ow in [(10, 15), (40, 45), (7, 11), (20, 30)]:
        y = F.interpolate(x, size=(oh, ow), mode="nearest")
        np.testing.assert_array_equal(upsample_ref.upsample_nearest(x.numpy(), oh, ow), y.numpy())


def test_conv_oracle_matches_torch_cpu_convolution():
    """oracle/conv_ref.py (tap-by-tap restatement) == torch's CPU conv2d and its autograd, fp64."""
    import torch
    import torch.nn.functional as F
    from oracle import conv_ref
    g = torch.Generator().manual_seed(4)
    for (B, H, W) in [(2, 20, 24), (1, 33, 18)]:
        x = torch.randn(B, 3, H, W, dtype=torch.float64, generator=g)
        w = torch.randn(64, 3, 7, 7, dtype=torch.float64, generator=g).requires_grad_()
        y = F.conv2d(x, w, None, 2, 3)
        dy = torch.randn(y.shape, dtype=torch.float64, generator=g)
        y.backward(dy)
        assert torch.allclose(conv_ref.conv2d_ref(x, w.detach()), y.detach(), rtol=0, atol=1e-11)
        assert torch.allclose(conv_ref.conv2d_wgrad_ref(x, dy), w.grad, rtol=0, atol=1e-10)


def _metric_cases():
    z = np.load(os.path.join(GOLD, "metric_golden.npz"))
    for name in sorted({k.split("/")[0] for k in z.files}):
        yield name, {k.split("/")[1]: z[k] for k in z.files if k.startswith(name + "/")}


def test_metric_oracle_matches_reference_golden():
    """oracle/metric_ref.py == the reference's metric.py (hist bit-exact, scores to 1e-12, NaN where it is NaN)."""
    from oracle import metric_ref
    n = 0
    for name, c in _metric_cases():
        n_cl = int(c["n_cl"])
        hist, labeled, correct = metric_ref.hist_info(n_cl, c["pred"], c["gt"])
        assert np.array_equal(hist, c["hist"]), name
        assert [labeled, correct] == c["counts"].tolist(), name
        iu, miu, miu_nb, acc = metric_ref.compute_score(hist, correct, labeled)
        np.testing.assert_allclose(iu, c["iu"], rtol=1e-12, atol=0, equal_nan=True, err_msg=name)
        np.testing.assert_allclose([miu, miu_nb, acc], c["scores"], rtol=1e-12, atol=0, equal_nan=True, err_msg=name)
        n += 1
    assert n == 4


def test_metric_host_mirror_scores_and_cpu_refusal():
    """Our seg_opr.metric.compute_score (host arithmetic) == reference golden; hist_info refuses to run without a GPU."""
    import sys
    import pytest
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(GOLD), "..", "torchseg_amd", "furnace"))
    from seg_opr import metric
    for name, c in _metric_cases():
        iu, miu, miu_nb, acc = metric.compute_score(c["hist"], c["counts"][1], c["counts"][0])
        np.testing.assert_allclose(iu, c["iu"], rtol=1e-12, equal_nan=True, err_msg=name)
        np.testing.assert_allclose([miu, miu_nb, acc], c["scores"], rtol=1e-12, equal_nan=True, err_msg=name)
    if not torch.cuda.is_available():
        with pytest.raises(Exception):
            metric.hist_info(19, np.zeros((4, 4), np.int64), np.zeros((4, 4), np.int64))


def test_augment_oracle_geometry_matches_torch_resampling():
    """oracle/augment_ref.py restates cv2.resize from OpenCV's documented geometry (cv2 is not installed).  The same
    geometry is what torch implements: INTER_LINEAR == F.interpolate(bilinear, align_corners=False, no antialias) on
    float data, INTER_NEAREST == torch 'nearest' (floor(dst * in / out)).  Pin the oracle's float taps to torch."""
    import numpy as np
    import torch
    import torch.nn.functional as F
    from oracle import augment_ref as R
    rng = np.random.RandomState(3)
    for (h, w, sh, sw) in [(20, 30, 35, 52), (20, 30, 10, 15), (17, 23, 17, 40), (8, 8, 3, 5)]:
        img = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
        t = torch.from_numpy(img).permute(2, 0, 1)[None].double()
        want = F.interpolate(t, size=(sh, sw), mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()
        raw = R.resize_linear_u8(img, sh, sw, rounded=False)
        assert np.abs(raw - want).max() <= 2e-4                                        # the taps and weights (float32 weight)
        got = R.resize_linear_u8(img, sh, sw).astype(np.float64)
        assert np.array_equal(got, np.clip(np.floor(raw + 0.5), 0, 255))               # the uint8 rounding rule
        gt = rng.randint(0, 19, size=(h, w)).astype(np.uint8)
        wantn = F.interpolate(torch.from_numpy(gt)[None, None].float(), size=(sh, sw), mode="nearest")[0, 0].numpy()
        # identical except where dst * in / out is an exact integer: torch forms the step in float32, OpenCV (and the
        # oracle) in double, and the two can land on different sides of that integer
        tie = ((np.arange(sh)[:, None] * h) % sh == 0) | ((np.arange(sw)[None, :] * w) % sw == 0)
        assert np.array_equal(R.resize_nearest(gt, sh, sw)[~tie], wantn.astype(np.uint8)[~tie])
    # crop + pad bookkeeping of random_crop_pad_to_shape / pad_image_to_shape (img_utils.py:24-75)
    a = np.arange(5 * 7).reshape(5, 7)
    p = R.pad_to_shape(a, (8, 10), 255)
    assert p.shape == (8, 10) and p[1, 1] == a[0, 0] and p[0, 0] == 255 and p[-2, -2] == 255 and p[5, 7] == a[4, 6]


def test_edge_oracle_equals_the_cv2_stand_in_and_the_kernel_sector_rule():
    """oracle/edge_ref.py (DFN border labels, dfn dataloader.py:24-29) against the cv2 stand-in the unchanged dataloader
    runs on here (Canny + dilate bit-equal), and the integer / double sector rule of csrc/augment.hip's edge_sobel_k
    against the oracle's arctan2 quantisation on every gradient pair a label image can produce nearby."""
    import importlib.util
    import os
    import numpy as np
    from oracle import edge_ref
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("cv2_standin", os.path.join(root, "torchseg_amd", "shims_optional", "cv2", "__init__.py"))
    cv2 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cv2)
    rng = np.random.RandomState(0)
    for (h, w) in [(40, 56), (33, 71), (8, 8)]:
        gt = np.repeat(np.repeat(rng.randint(0, 19, size=(h // 8 + 1, w // 8 + 1)).astype(np.uint8), 8, 0), 8, 1)[:h, :w].copy()
        a, b = cv2.Canny(gt, 5, 5, apertureSize=7), edge_ref.canny(gt, 5, 5, aperture=7)
        assert np.array_equal(a, b)
        assert np.array_equal(cv2.dilate(a, np.ones((7, 7), np.uint8)), edge_ref.dilate(b, 7))
    # the kernel's sector rule: q = 0 if |gy| < tan(22.5) |gx| (or both zero), 2 if |gy| > tan(67.5) |gx|, else 1 / 3 by sign
    gx, gy = np.meshgrid(np.arange(-600, 601, 7), np.arange(-600, 601, 5))
    gx, gy = gx.astype(np.float64).ravel(), gy.astype(np.float64).ravel()
    ang = (np.rad2deg(np.arctan2(gy, gx)) + 180.0) % 180.0
    want = ((ang + 22.5) // 45).astype(int) % 4
    a, b = np.abs(gx), np.abs(gy)
    t1, t2 = 0.41421356237309503, 2.414213562373095
    got = np.where((b < t1 * a) | ((a == 0) & (b == 0)), 0, np.where(b > t2 * a, 2, np.where((gx < 0) == (gy < 0), 1, 3)))
    assert np.array_equal(got, want)
