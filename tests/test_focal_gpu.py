"""GPU parity: sigmoid focal loss vs reference-generated golden vectors and the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import focal_ref

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_focal_golden(cuda):
    from torchseg_amd.losses import SigmoidFocalLoss
    z = np.load(os.path.join(GOLD, "focal_golden.npz"))
    for name in sorted({k.split("/")[0] for k in z.files}):
        gamma, alpha = z[name + "/cfg"]
        crit = SigmoidFocalLoss(255, gamma=float(gamma), alpha=float(alpha))
        for ltype in (torch.int64, torch.uint8):
            pred = torch.from_numpy(z[name + "/pred"]).to(cuda).requires_grad_(True)
            tgt = torch.from_numpy(z[name + "/target"].astype(np.int64)).to(cuda).to(ltype)
            loss = crit(pred, tgt)
            loss.backward()
            assert abs(loss.item() - float(z[name + "/loss"])) <= 1e-6
            np.testing.assert_allclose(pred.grad.cpu().numpy(), z[name + "/grad"], rtol=1e-4, atol=1e-8)


def test_focal_dfn_size_bf16(cuda):
    from torchseg_amd.losses import SigmoidFocalLoss
    g = torch.Generator().manual_seed(3)
    pred = (torch.randn(2, 1, 1024, 1024, generator=g) * 3).bfloat16()
    tgt = torch.randint(0, 2, (2, 1024, 1024), generator=g)
    tgt[torch.rand(tgt.shape, generator=g) < 0.1] = 255
    pr = pred.float().requires_grad_(True)
    ref = focal_ref.sigmoid_focal_loss(pr, tgt, 255, 2.0, 0.1)
    ref.backward()
    pd = pred.to(cuda).requires_grad_(True)
    loss = SigmoidFocalLoss(255, 2.0, 0.1)(pd, tgt.to(cuda))
    loss.backward()
    assert abs(loss.item() - ref.item()) <= 1e-5
    np.testing.assert_allclose(pd.grad.float().cpu().numpy(), pr.grad.numpy(), rtol=1e-2, atol=1e-9)
