"""One training step of each BASELINE.json model family other than BiSeNet (configs 3-5:
PSPNet-R50, DFN-R101, PSANet-R50 at reduced crop) through the HIP path on the GPU, against the
same network on the CPU with torch BatchNorm, the oracle focal loss and identical weights.

Tolerances (fp32).  Loss: 1e-5 relative.  Gradients: a randomly initialised 50/101-layer
network is ill-conditioned in fp32 (the error energy sits in the deep-stem weights; stock torch
on the same GPU -- MIOpen convs + torch BatchNorm/losses -- is 2.5e-2 .. 8.3e-2 away from the
CPU over all parameters and 0.7e-2 .. 1.6e-2 over the heads: tools/debug_families.py, DESIGN.md
section 4a), so the bound is stated against that: over the heads (everything outside `backbone.`)
and over all parameters, our path may be at most 2x as far from the CPU in relative L2 as stock
torch on the same device is.  Measured: ours 0.8e-2 .. 1.3e-2 (heads), 2.1e-2 .. 6.5e-2 (all),
i.e. closer to the CPU than stock torch in 5 of the 6 numbers."""
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
        B, S = 8, 128
        x = torch.randn(B, 3, S, S, generator=g)
        y = torch.randint(0, 19, (B, S, S), generator=g)
        y[:, :4] = 255
        e = torch.randint(0, 2, (B, S, S), generator=g)
        e[:, :, :4] = 255
        return (x, y, e)
    B, S = (8, 64) if kind == "pspnet" else (1, 480)
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
    from torchseg_amd import kernels as K
    kp = K.provider()
    calls = {"ohem_fwd": 0, "psa_fwd": 0}

    def counted(name):
        fn = getattr(kp, name)

        def f(*a, **k):
            calls[name] += 1
            return fn(*a, **k)
        return f
    for name in calls:
        setattr(kp, name, counted(name))
    stock = _build(kind, nn.BatchNorm2d, False)
    stock.load_state_dict(ref.state_dict())
    stock = stock.to(cuda)
    batch = _batch(kind)
    dbatch = [t.to(cuda) for t in batch]
    loss_ref = ref(*batch)
    loss_ref.backward()
    try:
        loss = net(*dbatch)
        loss.backward()
    finally:
        for name in calls:
            delattr(kp, name)
    # the plain nn.CrossEntropyLoss heads (a5) and the PSA contraction (a9) must have run on the HIP kernels
    assert calls["ohem_fwd"] == (4 if kind == "dfn" else 2), calls
    assert calls["psa_fwd"] == (2 if kind == "psanet" else 0), calls
    stock(*dbatch).backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - loss_ref.item()) <= 1e-5 * max(1.0, abs(loss_ref.item())), (loss.item(), loss_ref.item())

    def rel(model, keep):
        num = den = 0.0
        for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            assert (p.grad is None) == (q.grad is None), n
            if q.grad is None or not keep(n):
                continue
            d = p.grad.cpu().double() - q.grad.double()
            num += float((d * d).sum())
            den += float((q.grad.double() ** 2).sum())
        return (num / den) ** 0.5

    is_head = lambda n: not n.startswith("backbone.")
    ours_head, ours_all = rel(net.module, is_head), rel(net.module, lambda n: True)
    stock_head, stock_all = rel(stock, is_head), rel(stock, lambda n: True)
    print("%s: loss %.6f (cpu %.6f)  grad rel-L2 vs cpu: heads %.2e (stock torch %.2e), all %.2e (stock torch %.2e)"
          % (kind, loss.item(), loss_ref.item(), ours_head, stock_head, ours_all, stock_all))
    assert ours_head <= 2.0 * stock_head + 1e-3, (ours_head, stock_head)
    assert ours_all <= 2.0 * stock_all + 1e-3, (ours_all, stock_all)
