"""ADE20K, 150 classes (reference furnace/datasets/ade/ade.py): one file name per line in the source list (the label is
the same stem with .png), labels read as float32 so that the dataloader's `gt - 1` maps 0 to the ignore value -1.
Colours come from the dataset's own color150.mat; class names from its objectInfo150.csv when that file is at hand
(the reference hard-codes the same list), otherwise they are numbered."""
import csv
import os.path as osp

import numpy as np

from datasets.BaseDataset import BaseDataset, IMREAD_GRAYSCALE


class ADE(BaseDataset):
    def _fetch_data(self, img_path, gt_path, dtype=np.float32):
        return self._open_image(img_path), self._open_image(gt_path, IMREAD_GRAYSCALE, dtype=dtype)

    @staticmethod
    def _process_item_names(item):
        item = item.strip()
        return item, item.split('.')[0] + ".png"

    @classmethod
    def get_class_colors(*args):
        import scipy.io as sio
        colors = np.array(sio.loadmat(osp.join('.', 'color150.mat'))['colors'][:, ::-1]).astype(int).tolist()
        return [[0, 0, 0]] + colors

    @classmethod
    def get_class_names(*args):
        for path in ('objectInfo150.csv', osp.join('data', 'objectInfo150.csv')):
            if osp.exists(path):
                with open(path) as f:
                    return [row['Name'] for row in csv.DictReader(f)]
        return ['class %d' % (i + 1) for i in range(150)]
