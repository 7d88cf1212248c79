"""Loss operators of furnace/seg_opr/loss_opr.py on the HIP kernels.

`ProbOhemCrossEntropy2d` (loss_opr.py:48-98) and `SigmoidFocalLoss`
(loss_opr.py:14-45) keep the reference's constructor and forward signatures.
Both run entirely on the device: no `.item()`, no host-side branch on
`num_valid`, so a training step never synchronises inside the criterion (the
reference's `if self.min_kept > num_valid` / `elif num_valid > 0` are host
syncs, loss_opr.py:78-80).
"""
import logging

import torch
import torch.nn as nn

from . import kernels as K
from .fusion import outside_mode as _outside_mode

_CITYSCAPES_WEIGHT = [1.4297, 1.4805, 1.4363, 3.365, 2.6635, 1.4311, 2.1943, 1.4817,
                      1.4513, 2.1984, 1.5295, 1.6892, 3.2224, 1.4727, 7.5978, 9.4117,
                      15.2588, 5.6818, 2.2067]  # loss_opr.py:57-61

_log = logging.getLogger(__name__)


class _OhemCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, ignore_label, thresh, min_kept, weight):
        kp = K.provider()
        pred = pred.contiguous()          # NCHW planar logits
        target = target.contiguous()
        loss, nll, lse, sel = kp.ohem_fwd(pred, target, ignore_label, thresh, min_kept, weight)
        ctx.save_for_backward(pred, target, nll, lse, sel, weight)
        ctx.ignore_label = ignore_label
        ctx.mark_non_differentiable(sel)
        return loss.reshape(()), sel

    @staticmethod
    def backward(ctx, gloss, _gsel):
        kp = K.provider()
        pred, target, nll, lse, sel, weight = ctx.saved_tensors
        g = gloss.reshape(1).to(torch.float32).contiguous()
        dpred = kp.ohem_bwd(pred, target, ctx.ignore_label, weight, nll, lse, sel, g)
        return dpred, None, None, None, None, None


class _OhemUpCEFn(torch.autograd.Function):
    """criterion(F.interpolate(z)) with the interpolation evaluated inside the kernels."""

    @staticmethod
    def forward(ctx, z, target, OH, OW, ignore_label, thresh, min_kept, weight):
        kp = K.provider()
        z = z.contiguous()
        target = target.contiguous()
        loss, nll, lse, sel = kp.ohem_up_fwd(z, target, OH, OW, ignore_label, thresh, min_kept, weight)
        ctx.save_for_backward(z, target, nll, lse, sel, weight)
        ctx.cfg = (OH, OW, ignore_label)
        ctx.mark_non_differentiable(sel)
        return loss.reshape(()), sel

    @staticmethod
    def backward(ctx, gloss, _gsel):
        kp = K.provider()
        z, target, nll, lse, sel, weight = ctx.saved_tensors
        OH, OW, ignore_label = ctx.cfg
        g = gloss.reshape(1).to(torch.float32).contiguous()
        dz = kp.ohem_up_bwd(z, target, OH, OW, ignore_label, weight, nll, lse, sel, g)
        return dz, None, None, None, None, None, None, None


def ohem_cross_entropy(pred, target, ignore_label=255, thresh=0.7, min_kept=0, weight=None,
                       return_selection=False):
    """Functional form.  pred [B,C,H,W] (f32/bf16), target [B,H,W] (int64/uint8).

    selection (int32[8], on device) = {thr bits, n_kept, num_valid, branch, denom bits, ...}.
    """
    from .upsample import DeferredUpsample
    if isinstance(pred, DeferredUpsample):
        OH, OW = pred.out_hw
        z = pred.z
        if (target.dim() == 3 and tuple(target.shape) == (z.shape[0], OH, OW)
                and K.provider().ohem_up_supported(z, OH, OW, thresh)):
            loss, sel = _OhemUpCEFn.apply(z, target, OH, OW, int(ignore_label), float(thresh), int(min_kept), weight)
            return (loss, sel) if return_selection else loss
        pred = pred.materialize()
    if pred.dim() != 4 or target.dim() != 3:
        raise ValueError("expected pred [B,C,H,W] and target [B,H,W]")
    if pred.shape[0] != target.shape[0] or pred.shape[2:] != target.shape[1:]:
        raise ValueError(f"pred {tuple(pred.shape)} and target {tuple(target.shape)} do not match")
    loss, sel = _OhemCEFn.apply(pred, target, int(ignore_label), float(thresh), int(min_kept), weight)
    return (loss, sel) if return_selection else loss


def check_labels(sel, what="labels"):
    """Raise if the last criterion call saw labels that are neither the ignore value nor a class index (sel[5],
    see tsg_ohem_fwd).  The reference device-asserts in that situation; this check synchronises, so the criteria call
    it only under TSG_CHECK_LABELS=1 (or call it yourself once per epoch on `criterion.last_selection`)."""
    n_bad = int(sel[5].item())
    if n_bad:
        raise K.L.TsgError(f"{what}: {n_bad} label values are neither ignore_label nor a class index in [0, C)")


def _maybe_check_labels(sel):
    import os
    if os.environ.get("TSG_CHECK_LABELS", "0") == "1":
        check_labels(sel)


def cross_entropy_2d(pred, target, ignore_index=-100, weight=None, return_selection=False):
    """`F.cross_entropy(pred, target, weight, ignore_index=, reduction='mean')` for pred [B,C,H,W] / target [B,H,W]
    on the OHEM kernels in plain-CE mode (min_kept = 0 => branch 2 of tsg_ohem_fwd: every non-ignored pixel is
    kept).  One read of the logits forward, one read + one write backward, fp32 accumulation;
    mean over the non-ignored pixels (weighted mean with `weight`), NaN when there are none, like torch."""
    if weight is not None:
        weight = weight.to(device=pred.device, dtype=torch.float32).contiguous()
    out = ohem_cross_entropy(pred, target, ignore_label=ignore_index, thresh=1.0, min_kept=0, weight=weight,
                             return_selection=True)
    _maybe_check_labels(out[1])
    return out if return_selection else out[0]


class CrossEntropyLoss2d(nn.Module):
    """Drop-in for the `nn.CrossEntropyLoss(reduction='mean', ignore_index=...)` the reference builds for its plain
    heads (dfn train.py:48-49; pspnet / psanet train.py:48-49) on HIP tensors.  The unchanged train.py keeps
    constructing nn.CrossEntropyLoss: torchseg_amd.fusion.FuseMode routes that call here as well."""

    def __init__(self, weight=None, size_average=None, ignore_index=-100, reduce=None, reduction='mean',
                 label_smoothing=0.0):
        super().__init__()
        if size_average is not None or reduce is not None or reduction != 'mean' or label_smoothing != 0.0:
            raise NotImplementedError("only reduction='mean' without label smoothing (the reference's only use)")
        self.ignore_index = int(ignore_index)
        if weight is not None:
            self.register_buffer("weight", torch.as_tensor(weight, dtype=torch.float32))
        else:
            self.weight = None
        self.last_selection = None

    @_outside_mode
    def forward(self, pred, target):
        from .fusion import DeferredLogSoftmax
        if isinstance(pred, DeferredLogSoftmax):
            pred = pred.x                        # CE(log_softmax(x)) == CE(x)
        loss, sel = cross_entropy_2d(pred, target, self.ignore_index, self.weight, return_selection=True)
        self.last_selection = sel
        return loss


class ProbOhemCrossEntropy2d(nn.Module):
    """Drop-in for seg_opr.loss_opr.ProbOhemCrossEntropy2d (loss_opr.py:48-98).

    Constructed as train.py:50-52 does: (ignore_label=255, thresh=0.7,
    min_kept=..., use_weight=False).  `reduction` other than 'mean' is not used
    anywhere in the reference and is rejected.
    """

    def __init__(self, ignore_label, reduction='mean', thresh=0.6, min_kept=256,
                 down_ratio=1, use_weight=False):
        super().__init__()
        if reduction != 'mean':
            raise NotImplementedError("only reduction='mean' (the reference's only use) is implemented")
        self.ignore_label = ignore_label
        self.thresh = float(thresh)
        self.min_kept = int(min_kept)
        self.down_ratio = down_ratio
        if use_weight:
            self.register_buffer("weight", torch.tensor(_CITYSCAPES_WEIGHT, dtype=torch.float32))
        else:
            self.weight = None
        self.last_selection = None

    @_outside_mode
    def forward(self, pred, target):
        w = self.weight
        if w is not None and w.device != pred.device:
            w = w.to(pred.device)
        loss, sel = ohem_cross_entropy(pred, target, self.ignore_label, self.thresh, self.min_kept,
                                       w, return_selection=True)
        self.last_selection = sel
        _maybe_check_labels(sel)
        return loss


class _FocalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, ignore_label, gamma, alpha):
        kp = K.provider()
        pred = pred.contiguous()
        target = target.contiguous()
        loss = kp.focal_fwd(pred, target, ignore_label, gamma, alpha)
        ctx.save_for_backward(pred, target)
        ctx.cfg = (ignore_label, gamma, alpha)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gloss):
        kp = K.provider()
        pred, target = ctx.saved_tensors
        ignore_label, gamma, alpha = ctx.cfg
        g = gloss.reshape(1).to(torch.float32).contiguous()
        return kp.focal_bwd(pred, target, ignore_label, gamma, alpha, g), None, None, None, None


class SigmoidFocalLoss(nn.Module):
    """Drop-in for seg_opr.loss_opr.SigmoidFocalLoss (loss_opr.py:14-45), as DFN
    builds it: SigmoidFocalLoss(ignore_label=255, gamma=2.0, alpha=0.25)
    (model/dfn/cityscapes.dfn.R101_v1c/train.py:52)."""

    def __init__(self, ignore_label, gamma=2.0, alpha=0.25, reduction='mean'):
        super().__init__()
        if reduction != 'mean':
            raise NotImplementedError("only reduction='mean' (the reference's only use) is implemented")
        self.ignore_label = ignore_label
        self.gamma = gamma
        self.alpha = alpha
        self.reduction = reduction

    def forward(self, pred, target):
        b, h, w = target.size()
        if pred.numel() != b * h * w:
            raise ValueError("SigmoidFocalLoss expects one logit per pixel: pred [B,1,H,W]")
        return _FocalFn.apply(pred, target, int(self.ignore_label), float(self.gamma), float(self.alpha))
