"""Separates conv-rounding from our kernels in tests/test_families_gpu.py: gradients of (a) the
HIP path and (b) stock torch on the same GPU, both against the CPU run, per family."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, torch.nn as nn
import test_families_gpu as T
from torchseg_amd.ddp import DistributedDataParallel
from torchseg_amd.syncbn import SyncBatchNorm

dev = torch.device("cuda:0")


def relerr(named_a, named_b, top=4):
    num = den = 0.0
    rows = []
    for (n, p), (_, q) in zip(named_a, named_b):
        if q.grad is None:
            continue
        d = p.grad.cpu().double() - q.grad.double()
        e, r = float((d * d).sum()), float((q.grad.double() ** 2).sum())
        num += e; den += r
        rows.append((e, n, (e / (r + 1e-300)) ** 0.5))
    rows.sort(reverse=True)
    return (num / den) ** 0.5, [(n, "%.1e" % (e / num), "%.1e" % r) for e, n, r in rows[:top]]


for kind in sys.argv[1:] or ["pspnet", "dfn", "psanet"]:
    ref = T._build(kind, nn.BatchNorm2d, False)
    batch = T._batch(kind)
    lr = ref(*batch); lr.backward()
    ours = T._build(kind, SyncBatchNorm, True); ours.load_state_dict(ref.state_dict())
    ours = DistributedDataParallel(ours.to(dev), compute_dtype=torch.float32)
    lo = ours(*[t.to(dev) for t in batch]); lo.backward()
    stock = T._build(kind, nn.BatchNorm2d, False); stock.load_state_dict(ref.state_dict()); stock = stock.to(dev)
    ls = stock(*[t.to(dev) for t in batch]); ls.backward()
    torch.cuda.synchronize()
    print(kind, "loss cpu %.6f ours %.6f stock-gpu %.6f" % (lr.item(), lo.item(), ls.item()))
    print("  ours  vs cpu:", relerr(ours.module.named_parameters(), ref.named_parameters()))
    print("  stock vs cpu:", relerr(stock.named_parameters(), ref.named_parameters()))
    print("  ours  vs stock-gpu:", relerr(ours.module.named_parameters(), [(n, type("P", (), {"grad": None if p.grad is None else p.grad.cpu()})) for n, p in stock.named_parameters()]))
