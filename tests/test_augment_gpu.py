"""GPU parity of the fused training pre-processing kernel against oracle/augment_ref.py, the step-by-step
numpy restatement of TrainPre.__call__ / img_utils.py.  Labels bit-exact (nearest resize + crop + pad), images to 1e-5
in normalised units; every branch: flip, up- and down-scaling, crop inside, crop hanging over the border (the
reference's randint(+1)), image smaller than the crop (centred padding), odd padding, uint8 and int64 labels."""
import random

import numpy as np
import pytest
import torch

from oracle import augment_ref as R

pytestmark = pytest.mark.gpu
MEAN, STD = np.array([0.485, 0.456, 0.406]), np.array([0.229, 0.224, 0.225])      # config.py:69-70


def _sample(h, w, seed):
    rng = np.random.RandomState(seed)
    img = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
    gt = rng.randint(0, 19, size=(h, w)).astype(np.uint8)
    gt[rng.rand(h, w) < 0.1] = 255
    return img, gt


CASES = [  # (H, W), crop, dict(flip, scale, crop_y, crop_x)
    ((64, 96), (32, 48), dict(flip=False, scale=1.0, crop_y=5, crop_x=7)),
    ((64, 96), (32, 48), dict(flip=True, scale=1.5, crop_y=40, crop_x=60)),
    ((64, 96), (32, 48), dict(flip=True, scale=0.75, crop_y=16, crop_x=24)),       # crop reaches the border exactly
    ((64, 96), (32, 48), dict(flip=False, scale=0.5, crop_y=0, crop_x=0)),         # 32 x 48 after scaling: no padding
    ((64, 96), (48, 80), dict(flip=False, scale=0.5, crop_y=0, crop_x=0)),         # smaller than the crop: centred pad
    ((50, 70), (33, 41), dict(flip=True, scale=1.25, crop_y=30, crop_x=47)),       # hangs over both borders, odd pads
    ((37, 53), (16, 16), dict(flip=False, scale=2.0, crop_y=58, crop_x=90)),
    ((1024, 2048), (1024, 1024), dict(flip=True, scale=1.75, crop_y=300, crop_x=1500)),   # the headline crop
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("label_dtype", [torch.int64, torch.uint8])
def test_augment_matches_oracle(cuda, case, label_dtype):
    from torchseg_amd.data import GpuTrainPre
    (H, W), crop, p = case
    img, gt = _sample(H, W, H + W)
    p = dict(p, sh=int(H * p["scale"]), sw=int(W * p["scale"]))
    want_img, want_gt = R.train_pre(img, gt, p, MEAN, STD, crop)
    pre = GpuTrainPre(MEAN, STD, crop, label_dtype=label_dtype)
    data, label = pre([torch.from_numpy(img).to(cuda)], [torch.from_numpy(gt).to(cuda)], params=[p])
    assert data.shape == (1, 3) + tuple(crop) and label.dtype == label_dtype
    assert np.array_equal(label[0].cpu().numpy().astype(np.int64), want_gt)
    got = data[0].cpu().numpy()
    # a uint8 rounding tie (x.5 in the interpolated value) may resolve one level apart between float32 and float64
    # arithmetic: allow <= 1e-4 of the pixels to differ by exactly one grey level, everything else to 1e-5
    diff = np.abs(got - want_img)
    loose = diff > 1e-5
    assert loose.mean() <= 1e-4, loose.mean()
    assert diff.max() <= (1.0 / 255.0) / STD.min() + 1e-5


def test_batch_draws_follow_the_reference_call_order(cuda):
    """Seeded `random`: the batch pipeline consumes random.random / choice / randint exactly like TrainPre on each
    sample in turn, and the whole batch (20 samples = two launches) equals the oracle sample by sample."""
    from torchseg_amd.data import GpuTrainPre
    scales = [0.5, 0.75, 1.0, 1.5, 1.75, 2.0]                                      # config.train_scale_array
    crop = (40, 56)
    samples = [_sample(48 + 3 * i, 64 + 5 * i, i) for i in range(20)]
    r1, r2 = random.Random(7), random.Random(7)
    pre = GpuTrainPre(MEAN, STD, crop, scale_array=scales, rng=r1)
    data, label = pre([torch.from_numpy(s[0]).to(cuda) for s in samples], [torch.from_numpy(s[1]).to(cuda) for s in samples])
    for i, (img, gt) in enumerate(samples):
        p = R.draw_params(img.shape[:2], scales, crop, rng=r2)
        p["crop_y"], p["crop_x"] = min(p["crop_y"], p["sh"] - 1), min(p["crop_x"], p["sw"] - 1)
        want_img, want_gt = R.train_pre(img, gt, p, MEAN, STD, crop)
        assert np.array_equal(label[i].cpu().numpy(), want_gt), i
        assert (np.abs(data[i].cpu().numpy() - want_img) > 1e-5).mean() <= 1e-4, i
    assert r1.random() == r2.random()                     # both consumed the same number of draws


def test_synthetic_loader_yields_the_reference_dict(cuda):
    from torchseg_amd.data import GpuTrainPre, SyntheticSegLoader
    pre = GpuTrainPre(MEAN, STD, (64, 64), scale_array=[0.5, 1.0, 2.0], rng=random.Random(0))
    loader = SyntheticSegLoader(3, cuda, pre, image_hw=(96, 128), pool=2, length=4)
    it = iter(loader)
    mb = it.next()                                        # train.py:119
    assert set(mb) == {"data", "label", "fn", "n"}        # BaseDataset.py:60-63
    assert mb["data"].shape == (3, 3, 64, 64) and mb["data"].dtype == torch.float32 and mb["data"].is_cuda
    assert mb["label"].shape == (3, 64, 64) and mb["label"].dtype == torch.int64
    assert len(list(it)) == 3


# DFN's border labels (dfn dataloader.py:24-29): bit-exact against oracle/edge_ref.py
def _segments(h, w, seed):
    """Label images with structure: blocky segments with some ignored pixels (a Canny of pure noise is all edges)."""
    rng = np.random.RandomState(seed)
    bh, bw = max(h // 6, 1), max(w // 7, 1)
    gt = rng.randint(0, 19, size=((h + bh - 1) // bh, (w + bw - 1) // bw)).astype(np.uint8)
    gt = np.repeat(np.repeat(gt, bh, 0), bw, 1)[:h, :w].copy()
    gt[rng.rand(h, w) < 0.02] = 255
    yy, xx = np.mgrid[0:h, 0:w]
    gt[(yy - h // 2) ** 2 + (xx - w // 3) ** 2 < (min(h, w) // 5) ** 2] = 7          # a disc: every gradient direction
    return gt


@pytest.mark.parametrize("case", CASES[:7] + [((256, 512), (192, 192), dict(flip=True, scale=1.75, crop_y=100, crop_x=300))])
@pytest.mark.parametrize("label_dtype", [torch.int64, torch.uint8])
def test_dfn_edge_labels_match_oracle(cuda, case, label_dtype):
    from oracle import edge_ref
    from torchseg_amd.data import GpuTrainPreDFN
    (H, W), crop, p = case
    img, _ = _sample(H, W, H + W)
    gt = _segments(H, W, H * 3 + W)
    p = dict(p, sh=int(H * p["scale"]), sw=int(W * p["scale"]))
    want = edge_ref.dfn_edge_label(gt, p, crop)
    pre = GpuTrainPreDFN(MEAN, STD, crop, label_dtype=label_dtype)
    data, label, aux = pre([torch.from_numpy(img).to(cuda)], [torch.from_numpy(gt).to(cuda)], params=[p])
    assert aux.dtype == label_dtype and tuple(aux.shape) == (1,) + tuple(crop)
    got = aux[0].cpu().numpy().astype(np.int64)
    assert set(np.unique(got)) <= {0, 1, 255}
    assert np.array_equal(got, want), (int((got != want).sum()), got.size)
    assert 0 < (want == 1).mean() < 1                      # the case has edges and non-edges
    _, want_gt = R.train_pre(img, gt, p, MEAN, STD, crop)
    assert np.array_equal(label[0].cpu().numpy().astype(np.int64), want_gt)
