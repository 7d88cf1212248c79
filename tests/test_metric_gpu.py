"""GPU parity (bit-exact: integer work) of tsg_confusion_map / tsg_confusion_logits, through the seg_opr.metric
host mirror, with oracle/metric_ref.py, the reference golden vectors, and — at BASELINE size — with an independent
torch.bincount formulation plus the size-independent identities sum(hist) == labeled, trace(hist) == correct."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import metric_ref

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torchseg_amd", "furnace"))
GOLD = os.path.join(os.path.dirname(__file__), "golden", "metric_golden.npz")


def _cases():
    z = np.load(GOLD)
    for name in sorted({k.split("/")[0] for k in z.files}):
        yield name, {k.split("/")[1]: z[k] for k in z.files if k.startswith(name + "/")}


def test_hist_info_matches_reference_golden(cuda):
    from seg_opr import metric
    for name, c in _cases():
        n_cl = int(c["n_cl"])
        hist, labeled, correct = metric.hist_info(n_cl, c["pred"].astype(np.int64), c["gt"].astype(np.int64))
        assert np.array_equal(hist, c["hist"]), name
        assert [int(labeled), int(correct)] == c["counts"].tolist(), name
        if c["gt"].max() <= 255 and c["gt"].min() >= 0:            # uint8 label maps (Cityscapes png)
            h2, l2, c2 = metric.hist_info(n_cl, c["pred"].astype(np.uint8), c["gt"].astype(np.uint8))
            assert np.array_equal(h2, c["hist"]) and [int(l2), int(c2)] == c["counts"].tolist(), name


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 19, 24, 40), (1, 150, 17, 13), (3, 5, 1, 7), (1, 1, 4, 4)])
def test_hist_info_from_logits_vs_oracle(cuda, dtype, shape):
    from seg_opr import metric
    B, C, H, W = shape
    g = torch.Generator().manual_seed(B * 1000 + C)
    z = torch.randn(B, C, H, W, generator=g).to(dtype)
    if C > 2:
        z[:, 2, 0, :2] = z[:, 1, 0, :2]                             # ties -> first maximum
        z[0, C - 1, -1, -1] = float("nan")                          # NaN wins
        z[0, 0, -1, 0] = float("inf")
    gt = torch.randint(0, C, (B, H, W), generator=g)
    gt[torch.rand(B, H, W, generator=g) < 0.2] = 255 if C < 200 else -1
    pred = metric_ref.argmax_first(z.float().numpy())
    want = metric_ref.hist_info(C, pred, gt.numpy())
    got = metric.hist_info_from_logits(C, z.to(cuda), gt.to(cuda))
    assert np.array_equal(got[0], want[0]) and int(got[1]) == want[1] and int(got[2]) == want[2]
    acc = metric.ConfusionAccumulator(C)
    acc.add_logits(z.to(cuda), gt.to(cuda))
    acc.add_pred(torch.from_numpy(pred).to(cuda), gt.to(torch.uint8).to(cuda) if C < 200 else gt.to(cuda))
    h2, l2, c2 = acc.result()
    assert np.array_equal(h2, 2 * want[0]) and int(l2) == 2 * want[1] and int(c2) == 2 * want[2]


def test_out_of_range_prediction_is_reported(cuda):
    from seg_opr import metric
    with pytest.raises(ValueError):
        metric.hist_info(4, np.array([[0, 7]]), np.array([[0, 1]]))


def test_confusion_full_size_identities_and_bincount(cuda):
    """16 x 19 x 1024^2 bf16 logits (BASELINE config 2's evaluation shape)."""
    from seg_opr import metric
    B, C, S = 16, 19, 1024
    g = torch.Generator(device=cuda).manual_seed(0)
    z = torch.randn(B, C, S, S, device=cuda, generator=g).bfloat16()
    gt = torch.randint(0, C, (B, S, S), device=cuda, generator=g)
    gt[:, :32] = 255
    hist, labeled, correct = metric.hist_info_from_logits(C, z, gt)
    assert hist.sum() == labeled == B * (S - 32) * S
    assert np.trace(hist) == correct
    pred = z.float().argmax(1)
    k = gt < C
    ref = torch.bincount(C * gt[k] + pred[k], minlength=C * C).reshape(C, C).cpu().numpy()
    assert np.array_equal(hist, ref)
