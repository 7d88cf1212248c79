"""Stand-in for the few OpenCV entry points the reference's host-side scripts touch, for machines WITHOUT OpenCV
(this build image): `import cv2` at the top of an unchanged dataloader.py / eval.py then resolves.  Put this directory
on PYTHONPATH only when the real package is missing (tests/_dropin.py does exactly that) — it must never shadow cv2.

Implemented with numpy / Pillow from OpenCV's documented semantics (the same restatements as
furnace/utils/img_utils.py): imread / imwrite, resize (INTER_LINEAR, INTER_NEAREST), flip, copyMakeBorder,
getStructuringElement, dilate, Canny (Sobel + non-maximum suppression + hysteresis; NOT bit-identical to OpenCV's: the DFN
edge labels it produces are equivalent in kind, unpinned in detail).  GUI calls are no-ops."""
import numpy as np

__version__ = "4.0.0-torchseg_amd-standin"
IMREAD_GRAYSCALE, IMREAD_COLOR = 0, 1
INTER_NEAREST, INTER_LINEAR = 0, 1
BORDER_CONSTANT = 0
MORPH_RECT = 0
RETR_EXTERNAL, CHAIN_APPROX_SIMPLE = 0, 2


def imread(path, flags=IMREAD_COLOR):
    from PIL import Image
    try:
        with Image.open(path) as im:
            if flags == IMREAD_GRAYSCALE:
                return np.array(im.convert("L"))
            return np.ascontiguousarray(np.array(im.convert("RGB"))[:, :, ::-1])
    except (FileNotFoundError, OSError):
        return None


def imwrite(path, img):
    from PIL import Image
    a = np.asarray(img)
    Image.fromarray(a[:, :, ::-1] if a.ndim == 3 else a).save(path)
    return True


def imshow(*a, **k):
    return None


def waitKey(*a, **k):
    return -1


def _taps(n_in, n_out, inv_scale):
    step = 1.0 / inv_scale
    src = (np.arange(n_out, dtype=np.float64) + 0.5) * step - 0.5
    i0 = np.floor(src).astype(np.int64)
    w = (src - i0).astype(np.float32).astype(np.float64)
    i0, w = np.where(i0 < 0, 0, i0), np.where(i0 < 0, 0.0, w)
    edge = i0 >= n_in - 1
    i0, w = np.where(edge, n_in - 1, i0), np.where(edge, 0.0, w)
    return i0, np.minimum(i0 + 1, n_in - 1), w


def resize(src, dsize, dst=None, fx=0, fy=0, interpolation=INTER_LINEAR):
    src = np.asarray(src)
    h, w = src.shape[:2]
    if dsize is None or tuple(dsize) == (0, 0):
        ow, oh = int(round(w * fx)), int(round(h * fy))
        isx, isy = float(fx), float(fy)
    else:
        ow, oh = int(dsize[0]), int(dsize[1])
        isx, isy = ow / w, oh / h
    if interpolation == INTER_NEAREST:
        iy = np.minimum(np.floor(np.arange(oh) * (1.0 / isy)).astype(np.int64), h - 1)
        ix = np.minimum(np.floor(np.arange(ow) * (1.0 / isx)).astype(np.int64), w - 1)
        return np.ascontiguousarray(src[iy][:, ix])
    y0, y1, wy = _taps(h, oh, isy)
    x0, x1, wx = _taps(w, ow, isx)
    f = src.astype(np.float64)
    if f.ndim == 3:
        wy, wx = wy[:, None, None], wx[None, :, None]
    else:
        wy, wx = wy[:, None], wx[None, :]
    v = (1 - wy) * ((1 - wx) * f[y0][:, x0] + wx * f[y0][:, x1]) + wy * ((1 - wx) * f[y1][:, x0] + wx * f[y1][:, x1])
    if np.issubdtype(src.dtype, np.integer):
        info = np.iinfo(src.dtype)
        v = np.clip(np.floor(v + 0.5), info.min, info.max)
    return v.astype(src.dtype)


def flip(src, flipCode):
    a = np.asarray(src)
    if flipCode > 0:
        return np.ascontiguousarray(a[:, ::-1])
    if flipCode == 0:
        return np.ascontiguousarray(a[::-1])
    return np.ascontiguousarray(a[::-1, ::-1])


def copyMakeBorder(src, top, bottom, left, right, borderType, value=0):
    a = np.asarray(src)
    pads = [(int(top), int(bottom)), (int(left), int(right))] + [(0, 0)] * (a.ndim - 2)
    return np.pad(a, pads, mode="constant", constant_values=value)


def getStructuringElement(shape, ksize):
    return np.ones((int(ksize[1]), int(ksize[0])), np.uint8)


def dilate(src, kernel):
    a = np.asarray(src)
    kh, kw = kernel.shape
    ph, pw = kh // 2, kw // 2
    p = np.pad(a, ((ph, kh - 1 - ph), (pw, kw - 1 - pw)), mode="constant", constant_values=0)
    out = np.zeros_like(a)
    for dy in range(kh):
        for dx in range(kw):
            if kernel[dy, dx]:
                out = np.maximum(out, p[dy:dy + a.shape[0], dx:dx + a.shape[1]])
    return out


def _sobel16(a, aperture):
    """The two cv2.Sobel calls of cv2.Canny (imgproc/src/canny.cpp): 16-bit results, BORDER_REPLICATE; for aperture 7 scaled
    by 1 / 16 and rounded half to even (the unscaled 7 x 7 response of an 8-bit image overflows 16 bits)."""
    smooth = np.array([1], dtype=np.int64)
    for _ in range(aperture - 1):
        smooth = np.convolve(smooth, [1, 1])
    diff = np.array([1], dtype=np.int64)
    for _ in range(aperture - 2):
        diff = np.convolve(diff, [1, 1])
    diff = np.convolve(diff, [1, -1])[::-1] * -1 if aperture > 1 else diff
    r = aperture // 2
    p = np.pad(a.astype(np.int64), r, mode="edge")
    H, W = a.shape[0], a.shape[1]

    def sep(ky, kx):
        t = sum(int(ky[i]) * p[i:i + H, :] for i in range(aperture))
        return sum(int(kx[j]) * t[:, j:j + W] for j in range(aperture))
    gx, gy = sep(smooth, diff), sep(diff, smooth)
    if aperture == 7:
        def half_even(g):
            q, rem = g >> 4, g & 15
            return q + ((rem > 8) | ((rem == 8) & ((q & 1) == 1)))
        gx, gy = half_even(gx), half_even(gy)
    return np.clip(gx, -32768, 32767), np.clip(gy, -32768, 32767)


_TG22 = int(0.4142135623730950488016887242097 * (1 << 15) + 0.5)


def Canny(image, threshold1, threshold2, apertureSize=3, L2gradient=False):
    """OpenCV's integer Canny (L1 gradient): see _sobel16; thresholds of aperture 7 divided by 16, both floored; sectors by
    the fixed-point tangent test; non-maximum suppression strict against the left / upper neighbour and >= against the
    right / lower one, strict on both sides along the diagonals; zero magnitude outside the image; 8-connected hysteresis."""
    if L2gradient:
        raise NotImplementedError("the stand-in restates the L1-gradient path (what the reference calls)")
    a = np.asarray(image)
    if apertureSize == 7:
        threshold1, threshold2 = threshold1 / 16.0, threshold2 / 16.0
    lo = int(np.floor(min(threshold1, threshold2)))
    hi = int(np.floor(max(threshold1, threshold2)))
    gx, gy = _sobel16(a, apertureSize)
    mag = np.abs(gx) + np.abs(gy)
    x, y = np.abs(gx), np.abs(gy) << 15
    tg22x = x * _TG22
    tg67x = tg22x + (x << 16)
    horiz = y < tg22x
    vert = ~horiz & (y > tg67x)
    diag = ~(horiz | vert)
    opposite = (gx ^ gy) < 0
    pm = np.pad(mag, 1, mode="constant")
    H, W = mag.shape

    def nb(dy, dx):
        return pm[1 + dy:1 + dy + H, 1 + dx:1 + dx + W]
    keep = horiz & (mag > nb(0, -1)) & (mag >= nb(0, 1))
    keep |= vert & (mag > nb(-1, 0)) & (mag >= nb(1, 0))
    keep |= diag & ~opposite & (mag > nb(-1, -1)) & (mag > nb(1, 1))
    keep |= diag & opposite & (mag > nb(-1, 1)) & (mag > nb(1, -1))
    weak = keep & (mag > lo)
    out = weak & (mag > hi)
    while True:                                   # hysteresis: grow strong edges through connected weak ones
        grown = dilate(out.astype(np.uint8), np.ones((3, 3), np.uint8)).astype(bool) & weak
        if (grown == out).all():
            break
        out = grown
    return out.astype(np.uint8) * 255
