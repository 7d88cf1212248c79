"""One training step of each BASELINE.json model family other than BiSeNet (configs 3-5:
PSPNet-R50, DFN-R101, PSANet-R50 at reduced crop) through the HIP path on the GPU, against the
same network on the CPU with torch BatchNorm, the oracle focal loss and identical weights.
Tolerances (fp32): loss 5e-4 relative; gradients 2e-2 in relative L2 over all parameters (the
1x1 / 2x2 pooled BN layers normalise over 2-8 values, so MIOpen-vs-CPU conv rounding is
amplified; BiSeNet's smoke bound is 3e-3)."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _build(kind, norm, gpu):
    torch.manual_seed(77)
    if kind == "dfn":
        from torchseg_amd.workloads.dfn import DFN
        if gpu:
            from torchseg_amd.losses import SigmoidFocalLoss
        else:
            from oracle.focal_ref import SigmoidFocalLoss
        return DFN(19, nn.CrossEntropyLoss(reduction='mean', ignore_index=255),
                   SigmoidFocalLoss(255, 2.0, 0.25), 0.1, None, norm)
    from torchseg_amd.workloads.pspnet import PSANet, PSPNet
    cls = PSPNet if kind == "pspnet" else PSANet
    net = cls(150, nn.CrossEntropyLoss(reduction='mean', ignore_index=-1), None, norm, depth=50)
    for m in net.modules():
        if isinstance(m, nn.Dropout2d):
            m.p = 0.0               # CPU and GPU RNG streams differ
    return net


def _batch(kind):
    g = torch.Generator().manual_seed(5)
    if kind == "dfn":
        B, S = 2, 64
        x = torch.randn(B, 3, S, S, generator=g)
        y = torch.randint(0, 19, (B, S, S), generator=g)
        y[:, :4] = 255
        e = torch.randint(0, 2, (B, S, S), generator=g)
        e[:, :, :4] = 255
        return (x, y, e)
    B, S = (2, 96) if kind == "pspnet" else (1, 480)
    x = torch.randn(B, 3, S, S, generator=g)
    y = torch.randint(0, 150, (B, S, S), generator=g)
    y[:, :4] = -1
    return (x, y)


@pytest.mark.parametrize("kind", ["pspnet", "dfn", "psanet"])
def test_family_step_matches_cpu(cuda, kind):
    from torchseg_amd.ddp import DistributedDataParallel
    from torchseg_amd.syncbn import SyncBatchNorm
    ref = _build(kind, nn.BatchNorm2d, False)
    net = _build(kind, SyncBatchNorm, True)
    net.load_state_dict(ref.state_dict())
    net = DistributedDataParallel(net.to(cuda), compute_dtype=torch.float32)
    if kind == "psanet":
        assert net.fuse_psa, "PSANet must run its attention through tsg_psa_*"
    batch = _batch(kind)
    loss_ref = ref(*batch)
    loss_ref.backward()
    loss = net(*[t.to(cuda) for t in batch])
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - loss_ref.item()) <= 5e-4 * max(1.0, abs(loss_ref.item())), (loss.item(), loss_ref.item())
    num = den = 0.0
    for (n, p), (_, q) in zip(net.module.named_parameters(), ref.named_parameters()):
        assert (p.grad is None) == (q.grad is None), n
        if q.grad is None:
            continue
        d = p.grad.cpu().double() - q.grad.double()
        num += float((d * d).sum())
        den += float((q.grad.double() ** 2).sum())
    rel = (num / den) ** 0.5
    print("%s: loss %.6f (cpu %.6f) grad rel-L2 %.2e" % (kind, loss.item(), loss_ref.item(), rel))
    assert rel <= 2e-2, rel
