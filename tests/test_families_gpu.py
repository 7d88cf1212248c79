"""One training step of each BASELINE.json model family other than BiSeNet AT THE PER-RANK SHAPE bench.py --config
times (configs 3-5: PSPNet-R50_v1c 2 x 720^2, DFN-R101_v1c 2 x 1024^2, PSANet-R101_v1c 2 x 480^2; round 2 tested 64^2 -
128^2 crops and PSANet-R50) through the HIP path on the GPU, in fp32, against the same network on the CPU (the oracle:
torch BatchNorm, nn.CrossEntropyLoss, the loss_opr.py focal restatement) with identical weights and inputs.  The
models, inputs and optimizer groups come from bench.build_model / bench.synthetic_batch, so what is checked is what
is timed.  A CPU step of these networks takes 10-60 s on the GPU box's host cores.

Tolerances (fp32).  Loss: 1e-4 relative (north_star), 1e-5 measured.  Gradients: a randomly initialised 50/101-layer
network is ill-conditioned in fp32 (the error energy sits in the deep-stem weights; stock torch
on the same GPU -- MIOpen convs + torch BatchNorm/losses -- is 2.5e-2 .. 8.3e-2 away from the
CPU over all parameters and 0.7e-2 .. 1.6e-2 over the heads: tools/debug_families.py, DESIGN.md
section 4a), so the bound is stated against that: over the heads (everything outside `backbone.`)
and over all parameters, our path may be at most 2x as far from the CPU in relative L2 as stock
torch on the same device is.  Measured: ours 0.8e-2 .. 1.3e-2 (heads), 2.1e-2 .. 6.5e-2 (all),
i.e. closer to the CPU than stock torch in 5 of the 6 numbers."""
import os
import sys

import pytest
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.gpu


def _build(kind, norm, gpu):
    import bench
    from torchseg_amd.workloads import ensure_furnace_on_path
    ensure_furnace_on_path()
    if gpu:
        from torchseg_amd.losses import SigmoidFocalLoss
    else:
        from oracle.focal_ref import SigmoidFocalLoss
    cfg = bench.CONFIGS[kind]
    # dropout off: CPU and GPU RNG streams differ
    net, _, _ = bench.build_model(torch.device("cpu"), cfg["batch"], cfg["size"], None, norm, seed=77, config=kind,
                                  focal_cls=SigmoidFocalLoss, dropout=False)
    return net


def _batch(kind):
    import bench
    cfg = bench.CONFIGS[kind]
    return bench.synthetic_batch(torch.device("cpu"), cfg["batch"], cfg["size"], seed=5, config=kind)


@pytest.mark.parametrize("kind", ["pspnet", "dfn", "psanet"])
def test_family_step_matches_cpu(cuda, kind):
    from torchseg_amd.ddp import DistributedDataParallel
    from torchseg_amd.syncbn import SyncBatchNorm
    torch.set_num_threads(min(os.cpu_count() or 1, 64))      # torch's CPU convolutions stop scaling beyond this
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
    assert abs(loss.item() - loss_ref.item()) <= 1e-4 * max(1.0, abs(loss_ref.item())), (loss.item(), loss_ref.item())

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
