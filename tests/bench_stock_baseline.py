"""Stock PyTorch-ROCm sanity baseline: the same BiSeNet-R18 step with
nn.BatchNorm2d (== SyncBN at world 1), ATen upsample, and the reference-Python OHEM
(oracle restatement, run on the GPU) — the "before" img/s each hand-written kernel must beat.
    python tests/bench_stock_baseline.py [--steps K --warmup W --batch B --size S --dtype bf16|fp32 --nchw]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repo root
import torch, torch.nn as nn
import bench
from oracle.ohem_ref import ProbOhemCrossEntropy2d as RefOhem
from torchseg_amd.workloads import ensure_furnace_on_path
ensure_furnace_on_path()
from engine.lr_policy import PolyLR

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10); ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--batch", type=int, default=16); ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--dtype", default="bf16"); ap.add_argument("--nchw", action="store_true")
ap.add_argument("--miopen-find", type=int, default=0)
a = ap.parse_args()
torch.backends.cudnn.benchmark = bool(a.miopen_find)
dev = torch.device("cuda:0")
model, opt, base_lr = bench.build_model(dev, a.batch, a.size, RefOhem, nn.BatchNorm2d)
if not a.nchw:
    from torchseg_amd.ddp import apply_channels_last
    apply_channels_last(model)
model.train()
imgs, gts = bench.synthetic_batch(dev, a.batch, a.size)
pol = PolyLR(base_lr, 0.9, 80000)
def step(it):
    opt.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=a.dtype == "bf16"):
        loss = model(imgs, gts)
    lr = pol.get_lr(it)
    for i, g in enumerate(opt.param_groups): g['lr'] = lr if i < 2 else lr * 10
    loss.backward(); opt.step(); return loss
for it in range(a.warmup): step(it)
torch.cuda.synchronize(); t0 = time.perf_counter()
for it in range(a.steps): loss = step(a.warmup + it)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({"stock_img_per_s": round(a.batch * a.steps / dt, 2), "ms_per_step": round(dt / a.steps * 1e3, 2),
                  "dtype": a.dtype, "layout": "nchw" if a.nchw else "channels_last", "loss": loss.item()}))
