"""Training input pipeline on the GPU.

`GpuTrainPre` is the batch form of the reference's `TrainPre.__call__`
(model/bisenet/cityscapes.bisenet.R18/dataloader.py:16-35): random mirror, random scale, normalise, random crop,
pad, CHW — one `tsg_augment_crop` launch per <= 16 samples, sampling straight from the uint8 source images that sit
in HBM.  The random draws are made on the host with the reference's own call order on Python's `random`
(furnace/utils/img_utils.py:139, :111, :52-55), so a seeded run picks the same flips / scales / crops as the CPU
pipeline; only the resampling arithmetic moved (cv2 on 24 CPU workers in the reference, config.py:85).

`SyntheticSegLoader` stands in for `get_train_loader` where no dataset exists (bench, tests): it yields the dicts
`BaseDataset.__getitem__` + the default collate produce (datasets/BaseDataset.py:60-63: data, label, fn, n) and has
the `.next()` the reference's train.py:119 calls on its iterator.
"""
import random

import numpy as np
import torch

from . import kernels as K


class GpuTrainPre(object):
    def __init__(self, img_mean, img_std, crop_size, scale_array=None, ignore_label=255, label_dtype=torch.int64,
                 rng=None):
        self.mean = np.asarray(img_mean, dtype=np.float32)
        self.std = np.asarray(img_std, dtype=np.float32)
        self.crop_size = (int(crop_size[0]), int(crop_size[1])) if hasattr(crop_size, "__len__") else (int(crop_size),) * 2
        self.scale_array = list(scale_array) if scale_array is not None else None
        self.ignore_label = int(ignore_label)
        self.label_dtype = label_dtype
        self.rng = rng if rng is not None else random

    def draw(self, img_shape):
        """One sample's random parameters, in the reference's call order."""
        r = self.rng
        flip = r.random() >= 0.5                                             # random_mirror, img_utils.py:139
        scale = r.choice(self.scale_array) if self.scale_array is not None else 1.0     # random_scale, :111
        sh, sw = int(img_shape[0] * scale), int(img_shape[1] * scale)        # :112-113
        ch, cw = self.crop_size
        pos_h = r.randint(0, sh - ch + 1) if sh > ch else 0                  # generate_random_crop_pos, :52-55
        pos_w = r.randint(0, sw - cw + 1) if sw > cw else 0
        # a position one past the last full crop can be drawn (the reference's "+ 1"); it yields a padded crop
        pos_h, pos_w = min(pos_h, sh - 1), min(pos_w, sw - 1)
        return dict(flip=bool(flip), scale=scale, sh=sh, sw=sw, crop_y=pos_h, crop_x=pos_w)

    def __call__(self, imgs, gts, params=None):
        """imgs[i] uint8 [H,W,3] (RGB as BaseDataset.py:45 hands it over), gts[i] uint8 [H,W], on the GPU."""
        if params is None:
            params = [self.draw(im.shape[:2]) for im in imgs]
        geom = np.array([[im.shape[0], im.shape[1], p["sh"], p["sw"], int(p["flip"]), p["crop_y"], p["crop_x"]]
                         for im, p in zip(imgs, params)], dtype=np.int32)
        data, label = K.provider().augment_crop(imgs, gts, geom, self.crop_size, self.mean, self.std,
                                                pad_label=self.ignore_label, label_dtype=self.label_dtype)
        return data, label


class GpuTrainPreDFN(GpuTrainPre):
    """DFN's TrainPre (model/dfn/cityscapes.dfn.R101_v1c/dataloader.py:16-47): the BiSeNet pipeline plus the border label —
    Canny(aperture 7, thresholds 5 / 5) of the mirrored / scaled label image with 255 -> 0, 7 x 7 dilation, cropped and
    padded with 255 like the label (`extra_dict = {'aux_label': p_cgt}`).  Returns (data, label, aux_label); everything
    stays on the GPU (round 3 left the edge label on the host)."""

    edge_radius = 7                                                          # dataloader.py:20

    def __call__(self, imgs, gts, params=None):
        if params is None:
            params = [self.draw(im.shape[:2]) for im in imgs]
        data, label = super().__call__(imgs, gts, params)
        geom = np.array([[im.shape[0], im.shape[1], p["sh"], p["sw"], int(p["flip"]), p["crop_y"], p["crop_x"]]
                         for im, p in zip(imgs, params)], dtype=np.int32)
        aux = K.provider().edge_labels(gts, geom, self.crop_size, ignore_label=255, threshold=5, aperture=7,
                                       dilate_size=self.edge_radius, pad_label=255, label_dtype=self.label_dtype)
        return data, label, aux


class SyntheticSegLoader(object):
    """Endless loader of synthetic Cityscapes-shaped samples kept in HBM as uint8, augmented on the GPU per batch."""

    def __init__(self, batch_size, device, pre, image_hw=(1024, 2048), num_classes=19, pool=4, length=None, seed=0):
        self.batch_size, self.pre, self.length = int(batch_size), pre, length
        g = torch.Generator(device=device).manual_seed(seed)
        H, W = image_hw
        self.images = [torch.randint(0, 256, (H, W, 3), generator=g, device=device, dtype=torch.uint8) for _ in range(pool)]
        self.labels = []
        for _ in range(pool):
            t = torch.randint(0, num_classes, (H // 16, W // 16), generator=g, device=device, dtype=torch.uint8)
            t = t.repeat_interleave(16, 0).repeat_interleave(16, 1).contiguous()        # blocky "segments"
            t[:8] = pre.ignore_label
            self.labels.append(t)
        self._i = 0

    def __len__(self):
        return self.length if self.length is not None else 1 << 30

    def __iter__(self):
        return self

    def __next__(self):
        if self.length is not None and self._i >= self.length:
            raise StopIteration
        idx = [(self._i * self.batch_size + j) % len(self.images) for j in range(self.batch_size)]
        self._i += 1
        data, label = self.pre([self.images[k] for k in idx], [self.labels[k] for k in idx])
        return dict(data=data, label=label, fn=["synthetic_%06d" % k for k in idx], n=len(self.images))

    next = __next__          # train.py:119 calls dataloader.next()
