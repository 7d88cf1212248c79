"""Sigmoid focal loss oracle (torch CPU).  Test infrastructure only.

Restates SigmoidFocalLoss.forward (furnace/seg_opr/loss_opr.py:23-45) verbatim,
including the sigmoid-where-logit-was-meant quirk (:32-39).  Pinned by
tests/golden/focal_golden.npz (generated from the reference class)."""
import torch


def sigmoid_focal_loss(pred, target, ignore_label=255, gamma=2.0, alpha=0.25):
    b, h, w = target.size()                                   # :24
    pred = pred.float().reshape(b, -1, 1)                     # :25
    pred_sigmoid = pred.sigmoid()                             # :26
    target = target.reshape(b, -1).float()                    # :27
    mask = (target.ne(ignore_label)).float()                  # :28
    target = mask * target                                    # :29
    onehot = target.view(b, -1, 1)                            # :30
    max_val = (-pred_sigmoid).clamp(min=0)                    # :33
    pos_part = (1 - pred_sigmoid) ** gamma * (pred_sigmoid - pred_sigmoid * onehot)      # :35-36
    neg_part = pred_sigmoid ** gamma * (max_val + ((-max_val).exp() + (-pred_sigmoid - max_val).exp()).log())  # :37-38
    loss = -(alpha * pos_part + (1 - alpha) * neg_part).sum(dim=-1) * mask               # :40-41
    return loss.mean()                                        # :42-43


class SigmoidFocalLoss(torch.nn.Module):
    def __init__(self, ignore_label, gamma=2.0, alpha=0.25, reduction='mean'):
        super().__init__()
        self.ignore_label, self.gamma, self.alpha = ignore_label, gamma, alpha

    def forward(self, pred, target):
        return sigmoid_focal_loss(pred, target, self.ignore_label, self.gamma, self.alpha)
