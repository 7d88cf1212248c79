"""DFN border-label oracle (numpy).  Test infrastructure only.

Restates the label branch of TrainPre.__call__ in model/dfn/cityscapes.dfn.R101_v1c/dataloader.py:16-47 on top of
oracle/augment_ref.py:
    :24-26  no255_gt = gt with 255 -> 0
    :27     cgt = cv2.Canny(no255_gt, 5, 5, apertureSize=7)
    :28     cgt = cv2.dilate(cgt, cv2.getStructuringElement(cv2.MORPH_RECT, (7, 7)))        (:19-21)
    :29     cgt[cgt == 255] = 1
    :38     p_cgt = random_crop_pad_to_shape(cgt, crop_pos, crop_size, 255)
cv2 is not installed in the build image: Canny and dilate are restated from OpenCV's documented algorithm (separable
Sobel of the given aperture with reflect-101 borders, L1 gradient magnitude, non-maximum suppression along the gradient
direction quantised to four sectors, hysteresis between the two thresholds; rectangular dilation with zero borders).
Parity with OpenCV's own integer implementation is UNPINNED; the restatement is pinned to the cv2 stand-in
of this repository (torchseg_amd/shims_optional/cv2), on which the reference's UNCHANGED dataloader runs
(tests/test_oracles_cpu.py)."""
import numpy as np

from . import augment_ref as A


def sobel(a, aperture):
    """cv2.Sobel kernels: binomial smoothing x its difference, BORDER_REFLECT_101.  -> (gx, gy) float64 (exact integers)."""
    smooth = np.array([1.0])
    for _ in range(aperture - 1):
        smooth = np.convolve(smooth, [1.0, 1.0])
    diff = np.array([1.0])
    for _ in range(aperture - 2):
        diff = np.convolve(diff, [1.0, 1.0])
    diff = -np.convolve(diff, [1.0, -1.0])[::-1]
    r = aperture // 2
    p = np.pad(a.astype(np.float64), r, mode="reflect")
    H, W = a.shape

    def sep(ky, kx):
        t = sum(ky[i] * p[i:i + H, :] for i in range(aperture))
        return sum(kx[j] * t[:, j:j + W] for j in range(aperture))
    return sep(smooth, diff), sep(diff, smooth)


def dilate(a, size):
    lo, hi = size // 2, size - 1 - size // 2
    p = np.pad(a, ((lo, hi), (lo, hi)), mode="constant", constant_values=0)
    out = np.zeros_like(a)
    for dy in range(size):
        for dx in range(size):
            out = np.maximum(out, p[dy:dy + a.shape[0], dx:dx + a.shape[1]])
    return out


def canny(a, t1, t2, aperture=3):
    gx, gy = sobel(a, aperture)
    mag = np.abs(gx) + np.abs(gy)
    lo, hi = min(t1, t2), max(t1, t2)
    ang = (np.rad2deg(np.arctan2(gy, gx)) + 180.0) % 180.0
    q = ((ang + 22.5) // 45).astype(int) % 4
    pm = np.pad(mag, 1, mode="constant")
    H, W = mag.shape
    keep = np.zeros((H, W), bool)
    for k, (dy, dx) in {0: (0, 1), 1: (1, 1), 2: (1, 0), 3: (1, -1)}.items():
        n1 = pm[1 + dy:1 + dy + H, 1 + dx:1 + dx + W]
        n2 = pm[1 - dy:1 - dy + H, 1 - dx:1 - dx + W]
        keep |= (q == k) & (mag > n1) & (mag >= n2)
    strong, weak = keep & (mag > hi), keep & (mag > lo)
    out = strong.copy()
    while True:
        grown = dilate(out.astype(np.uint8), 3).astype(bool) & weak
        if (grown == out).all():
            break
        out = grown
    return out.astype(np.uint8) * 255


def dfn_edge_label(gt, params, crop_size, edge_radius=7, pad_label=255):
    """gt uint8 [H, W], params as augment_ref.draw_params -> int64 [ch, cw] with values {0, 1, 255}."""
    if params["flip"]:
        gt = gt[:, ::-1]                                          # random_mirror, dataloader.py:17
    gt = A.resize_nearest(gt, params["sh"], params["sw"])        # random_scale, :18-19
    no255 = np.array(gt)
    no255[gt == 255] = 0                                          # :24-26
    cgt = canny(no255, 5, 5, aperture=7)                          # :27
    cgt = dilate(cgt, edge_radius)                                # :28
    cgt[cgt == 255] = 1                                           # :29
    y0, x0 = params["crop_y"], params["crop_x"]
    c = cgt[y0:y0 + crop_size[0], x0:x0 + crop_size[1]]
    return np.ascontiguousarray(A.pad_to_shape(c, crop_size, pad_label)).astype(np.int64)   # :38
