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


def _sobel(a, aperture):
    """Separable Sobel of the given aperture (binomial smoothing x its difference), like cv2.Sobel's kernels."""
    smooth = np.array([1.0])
    for _ in range(aperture - 1):
        smooth = np.convolve(smooth, [1.0, 1.0])
    diff = np.array([1.0])
    for _ in range(aperture - 2):
        diff = np.convolve(diff, [1.0, 1.0])
    diff = np.convolve(diff, [1.0, -1.0])[::-1] * -1.0 if aperture > 1 else diff
    r = aperture // 2
    p = np.pad(a.astype(np.float64), r, mode="reflect")

    def sep(ky, kx):
        t = sum(ky[i] * p[i:i + a.shape[0] + 2 * r - (aperture - 1) + 0, :] for i in range(aperture))
        return sum(kx[j] * t[:, j:j + a.shape[1]] for j in range(aperture))
    return sep(smooth, diff), sep(diff, smooth)


def Canny(image, threshold1, threshold2, apertureSize=3, L2gradient=False):
    a = np.asarray(image)
    gx, gy = _sobel(a, apertureSize)
    mag = np.hypot(gx, gy) if L2gradient else np.abs(gx) + np.abs(gy)
    lo, hi = min(threshold1, threshold2), max(threshold1, threshold2)
    ang = (np.rad2deg(np.arctan2(gy, gx)) + 180.0) % 180.0
    q = ((ang + 22.5) // 45).astype(int) % 4
    pm = np.pad(mag, 1, mode="constant")
    H, W = mag.shape
    offs = {0: (0, 1), 1: (1, 1), 2: (1, 0), 3: (1, -1)}
    keep = np.zeros_like(mag, dtype=bool)
    for k, (dy, dx) in offs.items():
        n1 = pm[1 + dy:1 + dy + H, 1 + dx:1 + dx + W]
        n2 = pm[1 - dy:1 - dy + H, 1 - dx:1 - dx + W]
        keep |= (q == k) & (mag > n1) & (mag >= n2)
    strong = keep & (mag > hi)
    weak = keep & (mag > lo)
    out = strong.copy()
    while True:                                   # hysteresis: grow strong edges through connected weak ones
        grown = dilate(out.astype(np.uint8), np.ones((3, 3), np.uint8)).astype(bool) & weak
        if (grown == out).all():
            break
        out = grown
    return out.astype(np.uint8) * 255
