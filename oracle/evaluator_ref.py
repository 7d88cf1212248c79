"""Sliding-window / multi-scale / flip evaluation oracle (numpy + torch CPU).  Test infrastructure only.

Restates Evaluator.sliding_eval / scale_process / val_func_process / process_image of
furnace/engine/evaluator.py:186-298 statement by statement; cv2.resize / cv2.copyMakeBorder are the float restatements
of oracle/augment_ref.py (cv2 is not installed: parity with cv2's uint8 fixed-point rounding is unpinned)."""
import numpy as np
import torch

from . import augment_ref as A


def _cv_round(v):
    return int(round(v))                      # cvRound: round half to even, like Python's round


def resize_image_by_factor(img, s):
    """cv2.resize(img, None, fx=s, fy=s, interpolation=cv2.INTER_LINEAR) on a uint8 HWC image (evaluator.py:192-193)."""
    sh, sw = _cv_round(img.shape[0] * s), _cv_round(img.shape[1] * s)
    return A.resize_linear_u8(img, sh, sw, inv_scale=(s, s))


def resize_scores(score_hwc, oh, ow):
    """cv2.resize(score, (ow, oh), interpolation=cv2.INTER_LINEAR) on float32 HWC scores (evaluator.py:250-252)."""
    y0, y1, wy = A._lin_taps(score_hwc.shape[0], oh)
    x0, x1, wx = A._lin_taps(score_hwc.shape[1], ow)
    f = score_hwc.astype(np.float64)
    wy = wy[:, None, None]; wx = wx[None, :, None]
    top = (1 - wx) * f[y0][:, x0] + wx * f[y0][:, x1]
    bot = (1 - wx) * f[y1][:, x0] + wx * f[y1][:, x1]
    return ((1 - wy) * top + wy * bot).astype(np.float32)


def pad_image_to_shape(img, shape, value):
    """img_utils.pad_image_to_shape (:60-75) -> (padded, margin[4])."""
    if isinstance(shape, int):
        shape = (shape, shape)
    ph = max(shape[0] - img.shape[0], 0)
    pw = max(shape[1] - img.shape[1], 0)
    margin = np.array([ph // 2, ph // 2 + ph % 2, pw // 2, pw // 2 + pw % 2])
    pads = [(margin[0], margin[1]), (margin[2], margin[3])] + [(0, 0)] * (img.ndim - 2)
    return np.pad(img, pads, mode="constant", constant_values=value), margin


def process_image(img, mean, std, crop_size=None):
    """evaluator.py:275-298: normalise, (pad the NORMALISED image with 0), HWC -> CHW."""
    p = img.astype(np.float32) / 255.0
    p = (p - np.asarray(mean)) / np.asarray(std)
    if crop_size is not None:
        p, margin = pad_image_to_shape(p, crop_size, 0)
        return p.transpose(2, 0, 1), margin
    return p.transpose(2, 0, 1)


def val_func_process(net, input_chw, is_flip):
    """evaluator.py:255-273: log-probabilities of the net (+ those of the mirrored input, mirrored back), then exp."""
    x = torch.tensor(np.ascontiguousarray(input_chw[None], dtype=np.float32))
    net.eval()
    with torch.no_grad():
        score = net(x)[0]
        if is_flip:
            score = score + net(x.flip(-1))[0].flip(-1)
        score = torch.exp(score)
    return score.numpy()


def scale_process(net, img, ori_shape, crop_size, stride_rate, mean, std, is_flip):
    """evaluator.py:203-253 -> float32 [ori_rows, ori_cols, C]."""
    new_rows, new_cols, _ = img.shape
    long_size = new_cols if new_cols > new_rows else new_rows
    if long_size <= crop_size:
        input_data, margin = process_image(img, mean, std, crop_size)
        score = val_func_process(net, input_data, is_flip)
        score = score[:, margin[0]:(score.shape[1] - margin[1]), margin[2]:(score.shape[2] - margin[3])]
    else:
        stride = int(np.ceil(crop_size * stride_rate))
        img_pad, margin = pad_image_to_shape(img, crop_size, 0)                      # the RAW image is padded with 0
        pad_rows, pad_cols = img_pad.shape[0], img_pad.shape[1]
        r_grid = int(np.ceil((pad_rows - crop_size) / stride)) + 1
        c_grid = int(np.ceil((pad_cols - crop_size) / stride)) + 1
        data_scale = None
        for gy in range(r_grid):
            for gx in range(c_grid):
                s_x, s_y = gx * stride, gy * stride
                e_x, e_y = min(s_x + crop_size, pad_cols), min(s_y + crop_size, pad_rows)
                s_x, s_y = e_x - crop_size, e_y - crop_size
                img_sub = img_pad[s_y:e_y, s_x:e_x, :]
                input_data, tmargin = process_image(img_sub, mean, std, crop_size)
                t = val_func_process(net, input_data, is_flip)
                t = t[:, tmargin[0]:(t.shape[1] - tmargin[1]), tmargin[2]:(t.shape[2] - tmargin[3])]
                if data_scale is None:
                    data_scale = np.zeros((t.shape[0], pad_rows, pad_cols), dtype=np.float32)
                data_scale[:, s_y:e_y, s_x:e_x] += t
        score = data_scale[:, margin[0]:(data_scale.shape[1] - margin[1]), margin[2]:(data_scale.shape[2] - margin[3])]
    return resize_scores(np.ascontiguousarray(score.transpose(1, 2, 0)), ori_shape[0], ori_shape[1])


def sliding_eval(net, img, class_num, multi_scales, crop_size, stride_rate, mean, std, is_flip, return_scores=False):
    """evaluator.py:186-201 -> class map [ori_rows, ori_cols] (argmax of the scores summed over the scales)."""
    ori_rows, ori_cols, _ = img.shape
    processed = np.zeros((ori_rows, ori_cols, class_num))
    for s in multi_scales:
        img_scale = resize_image_by_factor(img, s)
        processed += scale_process(net, img_scale, (ori_rows, ori_cols), crop_size, stride_rate, mean, std, is_flip)
    pred = processed.argmax(2)
    return (pred, processed) if return_scores else pred
