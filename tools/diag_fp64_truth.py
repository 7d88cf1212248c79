"""How far is the reference CPU path itself from the exact answer?  north_star asks for fp32 logits within 1e-4 of the
reference CPU path; tests/test_headline_gpu.py measures 0.8-1.5e-3 between ANY GPU implementation and the CPU at 1024^2.
This script adds the missing yardstick: the same network evaluated in float64 on the host (the "truth" both fp32 paths
approximate).  Printed per head: max |cpu fp32 - fp64|, max |ours fp32 - fp64|, max |stock torch GPU fp32 - fp64| and the
pairwise |ours - cpu fp32|.  If the CPU fp32 path sits as far from the truth as ours does, then 1e-4 between the two is not
a property either implementation can have: both are one fp32 summation order away from the same exact value.
usage: python tools/diag_fp64_truth.py            (SIZE=1024 BATCH=2 by default; the CPU legs run without a GPU)"""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from torchseg_amd.workloads import ensure_furnace_on_path
ensure_furnace_on_path()
from torchseg_amd.workloads.bisenet import BiSeNet
B, S, C = int(os.environ.get("BATCH", "2")), int(os.environ.get("SIZE", "1024")), 19
torch.set_num_threads(min(os.cpu_count() or 1, 64))
torch.manual_seed(12345)
ref = BiSeNet(C, True, None, None, nn.BatchNorm2d); ref.train()
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 3, S, S, generator=g)
t0 = time.time()
with torch.no_grad():
    cpu32 = [t.clone() for t in ref.logits(x)]
t1 = time.time()
ref64 = copy.deepcopy(ref).double(); ref64.train()
with torch.no_grad():
    truth = [t.clone() for t in ref64.logits(x.double())]
print("cpu fp32 forward %.1f s, fp64 forward %.1f s (%d threads), B=%d S=%d" % (t1 - t0, time.time() - t1, torch.get_num_threads(), B, S), flush=True)


def dist(a, b):
    d = (a.double().cpu() - b.double().cpu()).abs()
    return d.max().item(), (d.pow(2).mean().sqrt() / b.double().pow(2).mean().sqrt()).item()


def report(tag, got, base):
    print(f"{tag:38s}", " | ".join("max %.2e rms-rel %.2e" % dist(a, b) for a, b in zip(got, base)), flush=True)


print("logit scale", ["%.2f" % t.abs().max().item() for t in truth])
report("cpu fp32 (reference path) vs fp64", cpu32, truth)
if torch.cuda.is_available():
    from torchseg_amd import workloads
    from torchseg_amd.ddp import DistributedDataParallel, apply_channels_last
    from torchseg_amd.syncbn import SyncBatchNorm
    dev = torch.device("cuda:0")
    m = BiSeNet(C, True, None, None, SyncBatchNorm); m.load_state_dict(ref.state_dict())
    m = DistributedDataParallel(m.to(dev), compute_dtype=torch.float32).train()
    with torch.no_grad():
        ours = m.module.logits(x.to(dev))
    workloads.NATIVE_FUSIONS = False
    s = BiSeNet(C, True, None, None, nn.BatchNorm2d); s.load_state_dict(ref.state_dict()); s = s.to(dev).train()
    apply_channels_last(s)
    with torch.no_grad():
        stock = s.logits(x.to(dev))
    workloads.NATIVE_FUSIONS = True
    from torchseg_amd import exactconv
    exactconv.ENABLED = False
    with torch.no_grad():
        ours_lib = m.module.logits(x.to(dev))
    exactconv.ENABLED = True
    report("ours fp32, exact convolutions vs fp64", ours, truth)
    report("ours fp32, vendor-library convs vs fp64", ours_lib, truth)
    report("ours fp32 (exact) vs cpu fp32", ours, cpu32)
    report("stock torch GPU fp32 vs fp64", stock, truth)
    report("ours fp32 (vendor convs) vs cpu fp32", ours_lib, cpu32)
    report("stock torch GPU fp32 vs cpu fp32", stock, cpu32)
