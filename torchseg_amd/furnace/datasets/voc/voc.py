"""PASCAL VOC 2012, 21 classes (reference furnace/datasets/voc/voc.py): the standard VOC colour map (bit-interleaved
class index) and names."""
from datasets.BaseDataset import BaseDataset

_NAMES = ['background', 'aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow',
          'diningtable', 'dog', 'horse', 'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train', 'tv/monitor']


def _voc_colormap(n):
    """The PASCAL VOC palette (bit k of the class index feeds bit 7 - k // 3 of channel k % 3), in the B, G, R order
    the reference lists its colours in (it draws with OpenCV)."""
    out = []
    for idx in range(n):
        rgb, c = [0, 0, 0], idx
        for shift in range(7, -1, -1):
            for ch in range(3):
                rgb[ch] |= ((c >> ch) & 1) << shift
            c >>= 3
        out.append(rgb[::-1])
    return out


class VOC(BaseDataset):
    @classmethod
    def get_class_colors(*args):
        return _voc_colormap(len(_NAMES))

    @classmethod
    def get_class_names(*args):
        return list(_NAMES)
