"""DFN border-label oracle (numpy).  Test infrastructure only.

Restates the label branch of TrainPre.__call__ in model/dfn/cityscapes.dfn.R101_v1c/dataloader.py:16-47 on top of
oracle/augment_ref.py:
    :24-26  no255_gt = gt with 255 -> 0
    :27     cgt = cv2.Canny(no255_gt, 5, 5, apertureSize=7)
    :28     cgt = cv2.dilate(cgt, cv2.getStructuringElement(cv2.MORPH_RECT, (7, 7)))        (:19-21)
    :29     cgt[cgt == 255] = 1
    :38     p_cgt = random_crop_pad_to_shape(cgt, crop_pos, crop_size, 255)
cv2 is not installed in the build image: Canny and dilate are restated from OpenCV's algorithm (modules/imgproc/src/
canny.cpp of OpenCV 3.x / 4.x, the versions a 2018-19 checkout of the reference ran on; round 5 follows the advisor's
reading of that source where round 4 followed the documentation's prose):
  * Sobel of the given aperture into 16-bit integers with BORDER_REPLICATE; for aperture 7 the gradients are scaled by
    1 / 16 (rounded half to even: a 7 x 7 Sobel of an 8-bit image does not fit 16 bits) and BOTH thresholds are divided by
    16 and floored — `Canny(gt, 5, 5, apertureSize=7)` therefore thresholds at 0;
  * L1 magnitude |dx| + |dy|; direction sectors from the fixed-point test |dy| << 15 against |dx| tan(22.5) and
    |dx| tan(67.5) with TG22 = round(tan(22.5 deg) 2^15);
  * non-maximum suppression: horizontal gradient `m > left && m >= right`, vertical `m > up && m >= down`, diagonal
    STRICT on both sides, the diagonal chosen by the sign of dx * dy; magnitudes outside the image are 0;
  * hysteresis: pixels above `high` and the pixels above `low` 8-connected to them.
Parity with OpenCV itself stays UNPINNED (no OpenCV here to produce a single vector: SURVEY 8 row f3); the restatement is
pinned to the cv2 stand-in of this repository (torchseg_amd/shims_optional/cv2), on which the reference's UNCHANGED
dataloader runs (tests/test_oracles_cpu.py)."""
import numpy as np

from . import augment_ref as A


def sobel(a, aperture):
    """cv2.Sobel(a, CV_16S, ..., ksize=aperture, scale, borderType=BORDER_REPLICATE) as cv2.Canny calls it: binomial
    smoothing x its difference; scale 1 / 16 with round-half-to-even for aperture 7.  -> (dx, dy) int64."""
    smooth = np.array([1], dtype=np.int64)
    for _ in range(aperture - 1):
        smooth = np.convolve(smooth, [1, 1])
    diff = np.array([1], dtype=np.int64)
    for _ in range(aperture - 2):
        diff = np.convolve(diff, [1, 1])
    diff = -np.convolve(diff, [1, -1])[::-1]
    r = aperture // 2
    p = np.pad(a.astype(np.int64), r, mode="edge")
    H, W = a.shape

    def sep(ky, kx):
        t = sum(int(ky[i]) * p[i:i + H, :] for i in range(aperture))
        return sum(int(kx[j]) * t[:, j:j + W] for j in range(aperture))
    gx, gy = sep(smooth, diff), sep(diff, smooth)
    if aperture == 7:
        def rne16(g):                                    # cvRound(g / 16.0): round half to even
            q, rem = g >> 4, g & 15
            return q + ((rem > 8) | ((rem == 8) & ((q & 1) == 1)))
        gx, gy = rne16(gx), rne16(gy)
    return np.clip(gx, -32768, 32767), np.clip(gy, -32768, 32767)


def dilate(a, size):
    lo, hi = size // 2, size - 1 - size // 2
    p = np.pad(a, ((lo, hi), (lo, hi)), mode="constant", constant_values=0)
    out = np.zeros_like(a)
    for dy in range(size):
        for dx in range(size):
            out = np.maximum(out, p[dy:dy + a.shape[0], dx:dx + a.shape[1]])
    return out


TG22 = int(0.4142135623730950488016887242097 * (1 << 15) + 0.5)


def canny(a, t1, t2, aperture=3):
    if aperture == 7:
        t1, t2 = t1 / 16.0, t2 / 16.0
    lo, hi = int(np.floor(min(t1, t2))), int(np.floor(max(t1, t2)))
    gx, gy = sobel(a, aperture)
    mag = np.abs(gx) + np.abs(gy)
    x, y = np.abs(gx), np.abs(gy) << 15
    tg22x = x * TG22
    tg67x = tg22x + (x << 16)
    horiz = y < tg22x
    vert = (~horiz) & (y > tg67x)
    diag = ~(horiz | vert)
    s_neg = (gx ^ gy) < 0                                       # dx, dy of opposite sign: the other diagonal
    pm = np.pad(mag, 1, mode="constant")
    H, W = mag.shape

    def at(dy, dx):
        return pm[1 + dy:1 + dy + H, 1 + dx:1 + dx + W]
    keep = horiz & (mag > at(0, -1)) & (mag >= at(0, 1))
    keep |= vert & (mag > at(-1, 0)) & (mag >= at(1, 0))
    keep |= diag & ~s_neg & (mag > at(-1, -1)) & (mag > at(1, 1))
    keep |= diag & s_neg & (mag > at(-1, 1)) & (mag > at(1, -1))
    keep &= mag > lo
    strong, weak = keep & (mag > hi), keep
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
