"""How often is the weight-shadow refresh launched per steady-state step, and from where?  (round 6: rocprof showed ~4
weight_shadow_k launches of 75 us per step where one is needed)"""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from torchseg_amd import shadow
from torchseg_amd.ddp import DistributedDataParallel
from torchseg_amd.losses import ProbOhemCrossEntropy2d, SigmoidFocalLoss
from torchseg_amd.syncbn import SyncBatchNorm
from torchseg_amd.workloads import ensure_furnace_on_path
ensure_furnace_on_path()
from engine.lr_policy import PolyLR
os.environ["TSG_DTYPE"] = "bf16"
from torchseg_amd.tuning import use_shipped_miopen_db
use_shipped_miopen_db(rank=0)
dev = torch.device("cuda:0")
model, opt, base_lr = bench.build_model(dev, 16, 1024, ProbOhemCrossEntropy2d, SyncBatchNorm, fused_sgd=True,
                                        focal_cls=SigmoidFocalLoss)
model = DistributedDataParallel(model); model.train()
batch = bench.synthetic_batch(dev, 16, 1024, label_dtype=torch.uint8)
pol = PolyLR(base_lr, 0.9, 80000)
calls = collections.Counter()
orig = shadow.bank.refresh_all
step = [0]
def traced(device):
    st = traceback.extract_stack(limit=8)
    who = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(st[:-1]))
    stale = [tuple(e.ref().shape) for e in shadow.bank.entries.values() if e.ref() is not None and e.version != e.ref()._version]
    calls[(step[0], who[:300], len(stale), str(stale[:3]))] += 1
    return orig(device)
shadow.bank.refresh_all = traced
for it in range(4):
    step[0] = it
    bench.train_step(model, opt, batch, pol, it, 1)
torch.cuda.synchronize()
for k, v in sorted(calls.items()):
    print(v, k)
