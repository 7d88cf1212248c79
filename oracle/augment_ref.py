"""Training pre-processing oracle (numpy, float64).  Test infrastructure only.

Restates TrainPre.__call__ of the reference's dataloaders (model/bisenet/cityscapes.bisenet.R18/dataloader.py:16-35)
step by step from furnace/utils/img_utils.py: random_mirror (:138-143), random_scale (:110-117), normalize (:174-180),
generate_random_crop_pos (:43-57), random_crop_pad_to_shape (:24-40) with pad_image_to_shape (:60-75), and the
float() / long() conversions of datasets/BaseDataset.py:47-48.  The cv2 calls are restated from OpenCV's documented
geometry (cv2 is not installed here, so the 11-bit fixed-point rounding of cv2's uint8 INTER_LINEAR is NOT reproduced:
parity with cv2 itself is unpinned; results are rounded to the nearest uint8 like any uint8 resize):
  cv2.flip(img, 1)                     img[:, ::-1]
  cv2.resize(INTER_LINEAR)             src = (dst + 0.5) * in / out - 0.5, taps clamped at the border with weight 0
  cv2.resize(INTER_NEAREST)            src = min(floor(dst * in / out), in - 1)
  cv2.copyMakeBorder(BORDER_CONSTANT)  np.pad with a constant
"""
import random

import numpy as np


def _cv_scale(n_in, n_out):
    """OpenCV derives the source step as 1 / inv_scale with inv_scale = dsize / ssize (resize.cpp), both in double."""
    return 1.0 / (float(n_out) / float(n_in))


def _lin_taps(n_in, n_out, inv_scale=None):
    scale = _cv_scale(n_in, n_out) if inv_scale is None else 1.0 / float(inv_scale)
    src = (np.arange(n_out, dtype=np.float64) + 0.5) * scale - 0.5
    i0 = np.floor(src).astype(np.int64)
    w = (src - i0).astype(np.float32).astype(np.float64)      # cv2 keeps the weight in float
    lo = i0 < 0
    i0[lo] = 0; w[lo] = 0.0
    hi = i0 >= n_in - 1
    i0[hi] = n_in - 1; w[hi] = 0.0
    i1 = np.minimum(i0 + 1, n_in - 1)
    return i0, i1, w


def resize_linear_u8(img, sh, sw, rounded=True, inv_scale=None):
    """cv2.resize(img, (sw, sh), interpolation=cv2.INTER_LINEAR) for an HWC uint8 image (float geometry, see header);
    rounded=False returns the interpolated values before the uint8 rounding."""
    y0, y1, wy = _lin_taps(img.shape[0], sh, None if inv_scale is None else inv_scale[0])
    x0, x1, wx = _lin_taps(img.shape[1], sw, None if inv_scale is None else inv_scale[1])
    f = img.astype(np.float64)
    wy = wy[:, None, None]; wx = wx[None, :, None]
    top = (1 - wx) * f[y0][:, x0] + wx * f[y0][:, x1]
    bot = (1 - wx) * f[y1][:, x0] + wx * f[y1][:, x1]
    v = (1 - wy) * top + wy * bot
    if not rounded:
        return v
    return np.clip(np.floor(v + 0.5), 0, 255).astype(np.uint8)


def resize_nearest(gt, sh, sw):
    """cv2.resize(gt, (sw, sh), interpolation=cv2.INTER_NEAREST)."""
    iy = np.minimum(np.floor(np.arange(sh) * _cv_scale(gt.shape[0], sh)).astype(np.int64), gt.shape[0] - 1)
    ix = np.minimum(np.floor(np.arange(sw) * _cv_scale(gt.shape[1], sw)).astype(np.int64), gt.shape[1] - 1)
    return gt[iy][:, ix]


def draw_params(img_shape, scales, crop_size, rng=random):
    """The reference's random call sequence: random_mirror -> random.random(); random_scale -> random.choice(scales);
    generate_random_crop_pos -> random.randint(0, h - crop_h + 1), random.randint(0, w - crop_w + 1) (img_utils.py:52-55;
    the +1 is the reference's, so a position one past the last full crop can be drawn and is then padded)."""
    flip = rng.random() >= 0.5                                 # img_utils.py:139
    scale = rng.choice(scales) if scales is not None else 1.0  # :111
    sh, sw = int(img_shape[0] * scale), int(img_shape[1] * scale)
    ch, cw = crop_size
    pos_h = rng.randint(0, sh - ch + 1) if sh > ch else 0
    pos_w = rng.randint(0, sw - cw + 1) if sw > cw else 0
    return dict(flip=bool(flip), scale=scale, sh=sh, sw=sw, crop_y=pos_h, crop_x=pos_w)


def pad_to_shape(a, shape, value):
    """pad_image_to_shape (img_utils.py:60-75)."""
    ph = max(shape[0] - a.shape[0], 0)
    pw = max(shape[1] - a.shape[1], 0)
    pads = [(ph // 2, ph // 2 + ph % 2), (pw // 2, pw // 2 + pw % 2)] + [(0, 0)] * (a.ndim - 2)
    return np.pad(a, pads, mode="constant", constant_values=value)


def train_pre(img, gt, params, mean, std, crop_size, pad_label=255):
    """-> (image float32 [3, ch, cw], label int64 [ch, cw]) for one sample, given the drawn parameters."""
    if params["flip"]:
        img, gt = img[:, ::-1], gt[:, ::-1]
    img = resize_linear_u8(img, params["sh"], params["sw"])
    gt = resize_nearest(gt, params["sh"], params["sw"])
    x = img.astype(np.float32) / 255.0                           # img_utils.py:176-178
    x = (x - np.asarray(mean)) / np.asarray(std)
    y0, x0 = params["crop_y"], params["crop_x"]
    assert 0 <= y0 < x.shape[0] and 0 <= x0 < x.shape[1]         # :27-28
    xc = x[y0:y0 + crop_size[0], x0:x0 + crop_size[1]]
    gc = gt[y0:y0 + crop_size[0], x0:x0 + crop_size[1]]
    xc = pad_to_shape(xc, crop_size, 0)
    gc = pad_to_shape(gc, crop_size, pad_label)
    return np.ascontiguousarray(xc.transpose(2, 0, 1)).astype(np.float32), np.ascontiguousarray(gc).astype(np.int64)
