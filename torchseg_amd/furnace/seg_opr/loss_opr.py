"""seg_opr.loss_opr surface (furnace/seg_opr/loss_opr.py): the two criteria the
reference's train.py files import (bisenet train.py:22,50-52; dfn train.py:22,52),
backed by the HIP kernels in torchseg_amd.losses."""
from torchseg_amd.losses import ProbOhemCrossEntropy2d, SigmoidFocalLoss, ohem_cross_entropy

__all__ = ['ProbOhemCrossEntropy2d', 'SigmoidFocalLoss', 'ohem_cross_entropy']
