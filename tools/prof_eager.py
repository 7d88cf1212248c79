"""Which framework (ATen) operators still launch kernels inside the bench step, with their input shapes and the Python
frames that call them?  torch.profiler over 3 steps of the headline configuration (same model / batch / optimizer as
bench.py); prints the operators that own device time, largest first.  usage: python tools/prof_eager.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import ProfilerActivity, profile
from torchseg_amd.tuning import use_shipped_miopen_db
use_shipped_miopen_db(0)
os.environ["TSG_DTYPE"] = "bf16"
from torchseg_amd.ddp import DistributedDataParallel
from torchseg_amd.losses import ProbOhemCrossEntropy2d, SigmoidFocalLoss
from torchseg_amd.syncbn import SyncBatchNorm
from torchseg_amd.workloads import ensure_furnace_on_path
ensure_furnace_on_path()
from engine.lr_policy import PolyLR
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = False
model, opt, base_lr = bench.build_model(dev, 16, 1024, ProbOhemCrossEntropy2d, SyncBatchNorm, fused_sgd=True, focal_cls=SigmoidFocalLoss)
model = DistributedDataParallel(model); model.train()
batch = bench.synthetic_batch(dev, 16, 1024, label_dtype=torch.uint8)
pol = PolyLR(base_lr, 0.9, 80000)
for it in range(6):
    bench.train_step(model, opt, batch, pol, it, 1)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for it in range(n):
        bench.train_step(model, opt, batch, pol, 6 + it, 1)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=8):
    dt = getattr(e, "self_device_time_total", None)
    if dt is None:
        dt = getattr(e, "self_cuda_time_total", 0)
    if dt <= 0 or not e.key.startswith("aten::"):
        continue
    rows.append((dt / n, e.count / n, e.key, str(e.input_shapes)[:120], [s for s in e.stack if "torchseg_amd" in s or "bench.py" in s or "furnace" in s][:4]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("ATen operators with device time: %.1f us per step in %d rows" % (tot, len(rows)))
for dt, cnt, key, shapes, stack in rows[:60]:
    print("%8.1f us  x%-4.1f %-34s %s" % (dt, cnt, key, shapes))
    for s in stack:
        print("              " + s.strip()[:150])
