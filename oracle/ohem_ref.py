"""OHEM cross-entropy oracle (torch CPU).  Test infrastructure only.

Line-by-line restatement of ProbOhemCrossEntropy2d.forward
(furnace/seg_opr/loss_opr.py:68-98) with the one edit modern torch needs
(`1 - valid_mask` on a bool tensor -> `~valid_mask`, lines 81 and 95).  Pinned
against the reference class itself by tests/golden/make_golden.py ->
tests/golden/ohem_golden.npz.
"""
import torch
import torch.nn.functional as F

CITYSCAPES_WEIGHT = [1.4297, 1.4805, 1.4363, 3.365, 2.6635, 1.4311, 2.1943, 1.4817,
                     1.4513, 2.1984, 1.5295, 1.6892, 3.2224, 1.4727, 7.5978, 9.4117,
                     15.2588, 5.6818, 2.2067]  # loss_opr.py:57-61


def ohem_select(pred, target, ignore_label=255, thresh=0.7, min_kept=0):
    """loss_opr.py:69-96.  Returns (new_target [B,H,W], info dict)."""
    b, c, h, w = pred.size()
    target = target.reshape(-1).clone()
    valid_mask = target.ne(ignore_label)                      # :71
    target = target * valid_mask.long()                       # :72
    num_valid = valid_mask.sum()                              # :73
    prob = F.softmax(pred.float(), dim=1)                     # :75
    prob = (prob.transpose(0, 1)).reshape(c, -1)              # :76
    info = dict(num_valid=int(num_valid), branch=2, threshold=float("inf"), mask_prob=None)
    if min_kept > num_valid:                                  # :78 (only logs)
        pass
    elif num_valid > 0:                                       # :80
        prob = prob.masked_fill_(~valid_mask, 1)              # :81
        mask_prob = prob[target, torch.arange(len(target), dtype=torch.long)]   # :82-83
        threshold = thresh                                    # :84
        info["mask_prob"] = mask_prob
        if min_kept > 0:                                      # :85
            _, index = torch.sort(mask_prob)                  # :86
            threshold_index = index[min(len(index), min_kept) - 1]              # :87
            info["branch"] = 0
            if mask_prob[threshold_index] > thresh:           # :88
                threshold = mask_prob[threshold_index]        # :89
                info["branch"] = 1
            kept_mask = mask_prob.le(threshold)               # :90
            target = target * kept_mask.long()                # :91
            valid_mask = valid_mask * kept_mask               # :92
            info["threshold"] = float(threshold)
    target = target.masked_fill_(~valid_mask, ignore_label)   # :95
    info["n_kept"] = int(valid_mask.sum())
    info["kept"] = valid_mask.view(b, h, w)
    return target.view(b, h, w), info                         # :96


def ohem_cross_entropy(pred, target, ignore_label=255, thresh=0.7, min_kept=0, weight=None,
                       return_info=False):
    """loss_opr.py:68-98: selection, then nn.CrossEntropyLoss(ignore_index, weight) (:62-66,98)."""
    new_target, info = ohem_select(pred.detach(), target, ignore_label, thresh, min_kept)
    loss = F.cross_entropy(pred.float(), new_target, weight=weight, ignore_index=ignore_label,
                           reduction="mean")
    return (loss, info) if return_info else loss


class ProbOhemCrossEntropy2d(torch.nn.Module):
    """Module form with the reference constructor (loss_opr.py:49-66); used by the CPU
    baseline / parity runs of whole networks."""

    def __init__(self, ignore_label, reduction='mean', thresh=0.6, min_kept=256, down_ratio=1,
                 use_weight=False):
        super().__init__()
        self.ignore_label, self.thresh, self.min_kept = ignore_label, float(thresh), int(min_kept)
        self.weight = torch.tensor(CITYSCAPES_WEIGHT) if use_weight else None

    def forward(self, pred, target):
        return ohem_cross_entropy(pred, target, self.ignore_label, self.thresh, self.min_kept, self.weight)
