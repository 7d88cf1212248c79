"""TEST INFRASTRUCTURE (oracle): CPU restatement of the evaluator's confusion matrix and scores,
furnace/seg_opr/metric.py:9-19 (`hist_info`) and :22-30 (`compute_score`), in plain numpy loops /
array ops written independently of the reference's bincount form.  Pinned against the reference
module itself through tests/golden/metric_golden.npz (tests/golden/make_golden.py imports
/root/reference/furnace/seg_opr/metric.py to produce it)."""
import numpy as np


def hist_info(n_cl, pred, gt):
    pred = np.asarray(pred).reshape(-1).astype(np.int64)
    gt = np.asarray(gt).reshape(-1).astype(np.int64)
    assert pred.shape == gt.shape
    hist = np.zeros((n_cl, n_cl), dtype=np.int64)
    labeled = correct = 0
    valid = (gt >= 0) & (gt < n_cl)                       # metric.py:11
    for g, p in zip(gt[valid], pred[valid]):
        hist[g, p] += 1                                   # metric.py:15-17: bincount(n_cl * gt + pred)
        labeled += 1                                      # metric.py:12
        correct += int(g == p)                            # metric.py:13
    return hist, labeled, correct


def argmax_first(logits):
    """Class arg-max over axis 1 of [B, C, ...]: first maximum, NaN wins (numpy's rule, which the
    reference evaluator gets from `.argmax`)."""
    return np.argmax(np.asarray(logits, dtype=np.float64), axis=1)


def compute_score(hist, correct, labeled):
    hist = np.asarray(hist, dtype=np.float64)
    n = hist.shape[0]
    iu = np.full(n, np.nan)
    for c in range(n):
        union = hist[c, :].sum() + hist[:, c].sum() - hist[c, c]     # metric.py:23
        if union > 0:
            iu[c] = hist[c, c] / union
    with np.errstate(all="ignore"):
        mean_iu = np.nanmean(iu)                                      # metric.py:24
        mean_iu_no_back = np.nanmean(iu[1:])                          # metric.py:25
    acc = correct / labeled if labeled else np.nan                    # metric.py:28
    return iu, mean_iu, mean_iu_no_back, acc
