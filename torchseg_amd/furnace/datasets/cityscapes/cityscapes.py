"""Cityscapes, 19 training classes (reference furnace/datasets/cityscapes/cityscapes.py): the official train-id palette
and names, and the train-id -> label-id mapping used when writing submissions."""
import numpy as np

from datasets.BaseDataset import BaseDataset

# (train id -> official label id, colour, name), the public Cityscapes label table
_TABLE = [(7, (128, 64, 128), 'road'), (8, (244, 35, 232), 'sidewalk'), (11, (70, 70, 70), 'building'),
          (12, (102, 102, 156), 'wall'), (13, (190, 153, 153), 'fence'), (17, (153, 153, 153), 'pole'),
          (19, (250, 170, 30), 'traffic light'), (20, (220, 220, 0), 'traffic sign'), (21, (107, 142, 35), 'vegetation'),
          (22, (152, 251, 152), 'terrain'), (23, (70, 130, 180), 'sky'), (24, (220, 20, 60), 'person'),
          (25, (255, 0, 0), 'rider'), (26, (0, 0, 142), 'car'), (27, (0, 0, 70), 'truck'), (28, (0, 60, 100), 'bus'),
          (31, (0, 80, 100), 'train'), (32, (0, 0, 230), 'motorcycle'), (33, (119, 11, 32), 'bicycle')]


class Cityscapes(BaseDataset):
    trans_labels = [row[0] for row in _TABLE]

    @classmethod
    def get_class_colors(*args):
        return [list(row[1]) for row in _TABLE]

    @classmethod
    def get_class_names(*args):
        return [row[2] for row in _TABLE]

    @classmethod
    def transform_label(cls, pred, name):
        """Train ids -> official label ids, and the submission file name (drop the last '_' field of `name`)."""
        label = np.zeros(pred.shape)
        for train_id in np.unique(pred):
            label[pred == train_id] = cls.trans_labels[train_id]
        new_name = '_'.join(name.split('.')[0].split('_')[:-1]) + '.png'
        print('Trans', name, 'to', new_name, '    ', np.unique(np.array(pred, np.uint8)), ' ---------> ',
              np.unique(np.array(label, np.uint8)))
        return label, new_name
