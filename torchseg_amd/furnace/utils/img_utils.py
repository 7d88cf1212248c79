"""utils.img_utils — the host-side image helpers an unchanged `dataloader.py` / `eval.py` imports
(reference furnace/utils/img_utils.py: same names, arguments, return values and `random` call order).

These run on the CPU in the reference too (they are the DataLoader workers' code); the MI355X form of the whole chain is
`torchseg_amd.data.GpuTrainPre` / `tsg_augment_crop`.  OpenCV is used when it is importable; otherwise the few cv2 calls
are evaluated by the numpy restatements below (same geometry: half-pixel-centre bilinear with border taps clamped,
floor nearest, constant border), so the module imports and works on a machine without cv2."""
import collections.abc
import numbers
import random

import numpy as np

try:                                             # pragma: no cover - depends on the installation
    import cv2
except ImportError:                              # this image has no OpenCV
    cv2 = None

INTER_NEAREST, INTER_LINEAR, BORDER_CONSTANT = 0, 1, 0      # cv2's values


# ---- cv2 stand-ins ---------------------------------------------------------------------------------------------
def _taps(n_in, n_out):
    step = 1.0 / (float(n_out) / float(n_in))
    src = (np.arange(n_out, dtype=np.float64) + 0.5) * step - 0.5
    i0 = np.floor(src).astype(np.int64)
    w = (src - i0).astype(np.float32).astype(np.float64)
    i0, w = np.where(i0 < 0, 0, i0), np.where(i0 < 0, 0.0, w)
    edge = i0 >= n_in - 1
    i0, w = np.where(edge, n_in - 1, i0), np.where(edge, 0.0, w)
    return i0, np.minimum(i0 + 1, n_in - 1), w


def _resize(img, dsize, interpolation):
    """cv2.resize(img, dsize=(w, h), interpolation=...) for INTER_LINEAR / INTER_NEAREST."""
    if cv2 is not None:
        return cv2.resize(img, dsize, interpolation=interpolation)
    ow, oh = int(dsize[0]), int(dsize[1])
    if interpolation == INTER_NEAREST:
        iy = np.minimum(np.floor(np.arange(oh) * (1.0 / (oh / img.shape[0]))).astype(np.int64), img.shape[0] - 1)
        ix = np.minimum(np.floor(np.arange(ow) * (1.0 / (ow / img.shape[1]))).astype(np.int64), img.shape[1] - 1)
        return np.ascontiguousarray(img[iy][:, ix])
    y0, y1, wy = _taps(img.shape[0], oh)
    x0, x1, wx = _taps(img.shape[1], ow)
    f = img.astype(np.float64)
    if f.ndim == 3:
        wy, wx = wy[:, None, None], wx[None, :, None]
    else:
        wy, wx = wy[:, None], wx[None, :]
    v = (1 - wy) * ((1 - wx) * f[y0][:, x0] + wx * f[y0][:, x1]) + wy * ((1 - wx) * f[y1][:, x0] + wx * f[y1][:, x1])
    if np.issubdtype(img.dtype, np.integer):
        info = np.iinfo(img.dtype)
        v = np.clip(np.floor(v + 0.5), info.min, info.max)
    return v.astype(img.dtype)


def _need_cv2(what):
    if cv2 is None:
        raise ImportError("utils.img_utils.%s needs OpenCV (cv2), which is not installed" % what)


# ---- the reference's functions -----------------------------------------------------------------------------------
def get_2dshape(shape, *, zero=True):
    if isinstance(shape, collections.abc.Iterable):
        h, w = map(int, shape)
    else:
        h = w = int(shape)
    assert min(h, w) >= (0 if zero else 1), 'invalid shape: {}'.format((h, w))
    return h, w


def pad_image_to_shape(img, shape, border_mode, value):
    """Centre `img` in a canvas of at least `shape`; returns (padded, margin = [top, bottom, left, right])."""
    h, w = get_2dshape(shape)
    ph, pw = max(h - img.shape[0], 0), max(w - img.shape[1], 0)
    margin = np.array([ph // 2, ph // 2 + ph % 2, pw // 2, pw // 2 + pw % 2], np.uint32)
    if cv2 is not None:
        out = cv2.copyMakeBorder(img, int(margin[0]), int(margin[1]), int(margin[2]), int(margin[3]), border_mode, value=value)
    else:
        pads = [(int(margin[0]), int(margin[1])), (int(margin[2]), int(margin[3]))] + [(0, 0)] * (img.ndim - 2)
        out = np.pad(img, pads, mode="constant", constant_values=value)
    return out, margin


def random_crop_pad_to_shape(img, crop_pos, crop_size, pad_label_value):
    h, w = img.shape[:2]
    y0, x0 = crop_pos
    assert 0 <= y0 < h and 0 <= x0 < w
    ch, cw = get_2dshape(crop_size)
    return pad_image_to_shape(img[y0:y0 + ch, x0:x0 + cw, ...], (ch, cw), BORDER_CONSTANT, pad_label_value)


def generate_random_crop_pos(ori_size, crop_size):
    h, w = get_2dshape(ori_size)
    ch, cw = get_2dshape(crop_size)
    pos_h = random.randint(0, h - ch + 1) if h > ch else 0       # the reference's inclusive "+ 1"
    pos_w = random.randint(0, w - cw + 1) if w > cw else 0
    return pos_h, pos_w


def pad_image_size_to_multiples_of(img, multiple, pad_value):
    up = lambda s: -(-s // multiple) * multiple                  # noqa: E731
    return pad_image_to_shape(img, (up(img.shape[0]), up(img.shape[1])), BORDER_CONSTANT, pad_value)


def resize_ensure_shortest_edge(img, edge_length, interpolation_mode=INTER_LINEAR):
    assert isinstance(edge_length, int) and edge_length > 0, edge_length
    h, w = img.shape[:2]
    if h < w:
        th, tw = edge_length, max(1, int(float(edge_length) / h * w))
    else:
        th, tw = max(1, int(float(edge_length) / w * h)), edge_length
    return _resize(img, (tw, th), interpolation_mode)


def random_scale(img, gt, scales):
    scale = random.choice(scales)
    sh, sw = int(img.shape[0] * scale), int(img.shape[1] * scale)
    return _resize(img, (sw, sh), INTER_LINEAR), _resize(gt, (sw, sh), INTER_NEAREST), scale


def random_scale_with_length(img, gt, length):
    size = random.choice(length)
    return _resize(img, (size, size), INTER_LINEAR), _resize(gt, (size, size), INTER_NEAREST), size


def random_mirror(img, gt):
    if random.random() >= 0.5:
        img, gt = np.ascontiguousarray(img[:, ::-1]), np.ascontiguousarray(gt[:, ::-1])
    return img, gt,


def random_rotation(img, gt):
    _need_cv2("random_rotation")
    angle = random.random() * 20 - 10
    h, w = img.shape[:2]
    m = cv2.getRotationMatrix2D((w / 2, h / 2), angle, 1)
    return cv2.warpAffine(img, m, (w, h), flags=cv2.INTER_LINEAR), cv2.warpAffine(gt, m, (w, h), flags=cv2.INTER_NEAREST)


def random_gaussian_blur(img):
    k = random.choice([1, 3, 5, 7])
    if k > 1:
        _need_cv2("random_gaussian_blur")
        img = cv2.GaussianBlur(img, (k, k), 0)
    return img


def center_crop(img, shape):
    h, w = shape[0], shape[1]
    y, x = (img.shape[0] - h) // 2, (img.shape[1] - w) // 2
    return img[y:y + h, x:x + w]


def random_crop(img, gt, size):
    ch, cw = (int(size), int(size)) if isinstance(size, numbers.Number) else (size[0], size[1])
    h, w = img.shape[:2]
    if h > ch:
        y = random.randint(0, h - ch + 1)
        img, gt = img[y:y + ch, :, :], gt[y:y + ch, :]
    if w > cw:
        x = random.randint(0, w - cw + 1)
        img, gt = img[:, x:x + cw, :], gt[:, x:x + cw]
    return img, gt


def normalize(img, mean, std):
    """uint8 -> float in the range the ImageNet-pretrained backbones expect."""
    return (img.astype(np.float32) / 255.0 - mean) / std


def findContours(*args, **kwargs):
    _need_cv2("findContours")
    res = cv2.findContours(*args, **kwargs)
    return res[-2], res[-1]                      # OpenCV 3 returns (image, contours, hierarchy), OpenCV 4 the last two
