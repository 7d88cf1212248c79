"""datasets.BaseDataset — file-list segmentation dataset with the sample dict of the reference
(furnace/datasets/BaseDataset.py): `dict(data=, label=, fn=, n=)` (+ the preprocess' extra entries), images handed to
`preprocess(img, gt)` as RGB uint8 HWC and uint8 HW, train samples converted to float / long tensors.

Reading uses OpenCV when it is installed and Pillow otherwise (this image ships Pillow only); both give BGR -> RGB uint8
arrays for 8-bit PNG / JPEG files."""
import os

import numpy as np
import torch
import torch.utils.data as data

try:                                             # pragma: no cover
    import cv2
    IMREAD_COLOR, IMREAD_GRAYSCALE = cv2.IMREAD_COLOR, cv2.IMREAD_GRAYSCALE
except ImportError:
    cv2 = None
    IMREAD_COLOR, IMREAD_GRAYSCALE = 1, 0


class BaseDataset(data.Dataset):
    def __init__(self, setting, split_name, preprocess=None, file_length=None):
        super(BaseDataset, self).__init__()
        self._split_name = split_name
        self._img_path, self._gt_path = setting['img_root'], setting['gt_root']
        self._train_source, self._eval_source = setting['train_source'], setting['eval_source']
        self._file_names = self._get_file_names(split_name)
        self._file_length = file_length
        self.preprocess = preprocess

    def __len__(self):
        return self._file_length if self._file_length is not None else len(self._file_names)

    def __getitem__(self, index):
        names = (self._construct_new_file_names(self._file_length) if self._file_length is not None
                 else self._file_names)[index]
        img, gt = self._fetch_data(os.path.join(self._img_path, names[0]), os.path.join(self._gt_path, names[1]))
        img = img[:, :, ::-1]                                    # BGR (as read) -> RGB
        extra = None
        if self.preprocess is not None:
            img, gt, extra = self.preprocess(img, gt)
        if self._split_name == 'train':
            img = torch.from_numpy(np.ascontiguousarray(img)).float()
            gt = torch.from_numpy(np.ascontiguousarray(gt)).long()
            if extra is not None:
                for k, v in extra.items():
                    t = torch.from_numpy(np.ascontiguousarray(v))
                    extra[k] = t.long() if 'label' in k else (t.float() if 'img' in k else t)
        out = dict(data=img, label=gt, fn=str(names[1].split("/")[-1].split(".")[0]), n=len(self._file_names))
        if extra is not None:
            out.update(**extra)
        return out

    def _fetch_data(self, img_path, gt_path, dtype=None):
        return self._open_image(img_path), self._open_image(gt_path, IMREAD_GRAYSCALE, dtype=dtype)

    def _get_file_names(self, split_name):
        assert split_name in ['train', 'val']
        source = self._eval_source if split_name == "val" else self._train_source
        with open(source) as f:
            return [list(self._process_item_names(line)) for line in f.readlines()]

    def _construct_new_file_names(self, length):
        """`length` names: whole passes over the list plus a random remainder (a fresh permutation each call)."""
        assert isinstance(length, int)
        n = len(self._file_names)
        names = self._file_names * (length // n)
        order = torch.randperm(n).tolist()
        return names + [self._file_names[i] for i in order[:length % n]]

    @staticmethod
    def _process_item_names(item):
        img_name, gt_name = item.strip().split('\t')[:2]
        return img_name, gt_name

    def get_length(self):
        return self.__len__()

    @staticmethod
    def _open_image(filepath, mode=IMREAD_COLOR, dtype=None):
        """-> HWC BGR (colour) or HW (grey) array, like cv2.imread."""
        if cv2 is not None:
            return np.array(cv2.imread(filepath, mode), dtype=dtype)
        from PIL import Image
        with Image.open(filepath) as im:
            if mode == IMREAD_GRAYSCALE:
                a = np.array(im.convert("L") if im.mode not in ("L", "P", "I;16", "I") else im)
            else:
                a = np.array(im.convert("RGB"))[:, :, ::-1]
        return np.array(a, dtype=dtype)

    @classmethod
    def get_class_colors(*args):
        raise NotImplementedError

    @classmethod
    def get_class_names(*args):
        raise NotImplementedError
