"""seg_opr.metric — the evaluator's metrics (reference furnace/seg_opr/metric.py) with the
confusion matrix computed on the MI355X (`tsg_confusion_map` / `tsg_confusion_logits`).

Same names, argument order and return values as the reference module:

    hist, labeled, correct = hist_info(n_cl, pred, gt)            # metric.py:9-19
    iu, mean_IU, mean_IU_no_back, mean_pixel_acc = compute_score(hist, correct, labeled)   # metric.py:22-30

`pred` / `gt` may be numpy arrays (what the reference's CPU evaluator passes; they are moved to
the GPU) or torch tensors already on the device.  `hist_info_from_logits` is the form a GPU
evaluator uses: the class arg-max is fused into the histogram kernel, so the score map is read
once.  `ConfusionAccumulator` keeps one device buffer for a whole validation set (eval.py:54-63
sums the per-image results on the host).  The small per-class arithmetic of compute_score stays
numpy, as in the reference.
"""
import numpy as np
import torch

from torchseg_amd import kernels as K

np.seterr(divide='ignore', invalid='ignore')


def _device():
    if not torch.cuda.is_available():
        raise K.L.TsgError("seg_opr.metric needs the MI355X (no CPU fallback in the product path)")
    return torch.device("cuda", torch.cuda.current_device())


def _as_label_tensor(a, dev):
    if isinstance(a, np.ndarray):
        if a.dtype != np.uint8:
            a = a.astype(np.int64)
        a = torch.from_numpy(np.ascontiguousarray(a))
    if a.dtype not in (torch.int64, torch.uint8):
        a = a.to(torch.int64)
    return a.to(dev).contiguous()


def _unpack(out, n_cl):
    host = out.cpu().numpy()
    if host[n_cl * n_cl + 2]:
        raise ValueError("hist_info: %d labelled pixels have a prediction outside [0, %d)" % (host[n_cl * n_cl + 2], n_cl))
    return host[:n_cl * n_cl].reshape(n_cl, n_cl).copy(), host[n_cl * n_cl], host[n_cl * n_cl + 1]


def hist_info(n_cl, pred, gt):
    assert (tuple(pred.shape) == tuple(gt.shape))
    dev = gt.device if isinstance(gt, torch.Tensor) and gt.is_cuda else _device()
    out = K.provider().confusion_map(_as_label_tensor(pred, dev), _as_label_tensor(gt, dev), n_cl)
    return _unpack(out, n_cl)


def hist_info_from_logits(n_cl, logits, gt):
    """logits: [B, C, H, W] (or [C, H, W]) float32 / bfloat16 on the GPU; gt: [B, H, W] ([H, W])."""
    if logits.dim() == 3:
        logits, gt = logits[None], gt[None]
    out = K.provider().confusion_logits(logits.contiguous(), _as_label_tensor(gt, logits.device), n_cl)
    return _unpack(out, n_cl)


class ConfusionAccumulator(object):
    """hist / labeled / correct of a whole validation set in one device buffer."""

    def __init__(self, n_cl, device=None):
        self.n_cl = n_cl
        self.out = torch.zeros(n_cl * n_cl + 3, dtype=torch.int64, device=device or _device())

    def add_logits(self, logits, gt):
        if logits.dim() == 3:
            logits, gt = logits[None], gt[None]
        K.provider().confusion_logits(logits.contiguous(), _as_label_tensor(gt, self.out.device), self.n_cl, self.out)

    def add_pred(self, pred, gt):
        dev = self.out.device
        K.provider().confusion_map(_as_label_tensor(pred, dev), _as_label_tensor(gt, dev), self.n_cl, self.out)

    def result(self):
        return _unpack(self.out, self.n_cl)


def compute_score(hist, correct, labeled):
    hist = np.asarray(hist)
    tp = np.diag(hist)
    iu = tp / (hist.sum(1) + hist.sum(0) - tp)
    mean_IU = np.nanmean(iu)
    mean_IU_no_back = np.nanmean(iu[1:])
    mean_pixel_acc = correct / labeled
    return iu, mean_IU, mean_IU_no_back, mean_pixel_acc


# ---- ADE-style per-image metrics (reference metric.py:33-91), small host-side numpy like the reference ----------------
def meanIoU(area_intersection, area_union):
    """Per-class IoU over a set of images ([n_images, n_classes] areas), its mean, and the mean without class 0."""
    iou = 1.0 * np.sum(area_intersection, axis=1) / np.sum(area_union, axis=1)
    return iou, np.nanmean(iou), np.nanmean(iou[1:])


def intersectionAndUnion(imPred, imLab, numClass):
    """Per-class intersection / union areas of one image; label -1 (after the +1 shift: 0) is unlabeled and removes the
    pixel from prediction and ground truth alike."""
    pred = np.asarray(imPred).copy() + 1
    lab = np.asarray(imLab).copy() + 1
    pred = pred * (lab > 0)
    hist = lambda a: np.histogram(a, bins=numClass, range=(1, numClass))[0]       # noqa: E731
    inter = hist(pred * (pred == lab))
    return inter, hist(pred) + hist(lab) - inter


def mean_pixel_accuracy(pixel_correct, pixel_labeled):
    return 1.0 * np.sum(pixel_correct) / (np.spacing(1) + np.sum(pixel_labeled))


def pixelAccuracy(imPred, imLab):
    """(accuracy, correct, labeled) over the labeled (>= 0) pixels of one image."""
    labeled = np.sum(imLab >= 0)
    correct = np.sum((imPred == imLab) * (imLab >= 0))
    return 1.0 * correct / labeled, correct, labeled


def accuracy(preds, label):
    valid = (label >= 0)
    hits = (valid * (preds == label)).sum()
    total = valid.sum()
    return float(hits) / (total + 1e-10), total
