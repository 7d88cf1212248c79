"""hipGraph root-causing, part 3 (bench shape): eager warm-up, capture of zero_grad+forward+backward, then replays with
the optimizer eager in between; per replay: loss, gradient norms per group of layers, first non-finite gradient /
parameter / buffer, and the same quantities from a twin model trained eagerly (same seed) for reference."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
import bench
from torchseg_amd.syncbn import SyncBatchNorm
from torchseg_amd.losses import ProbOhemCrossEntropy2d
from torchseg_amd.ddp import DistributedDataParallel
from torchseg_amd.workloads import ensure_furnace_on_path
ensure_furnace_on_path()
from engine.lr_policy import PolyLR

ap = argparse.ArgumentParser()
ap.add_argument("--opt", default="fused"); ap.add_argument("--batch", type=int, default=16); ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--warm", type=int, default=4); ap.add_argument("--steps", type=int, default=8); ap.add_argument("--tag", default="")
ap.add_argument("--side", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda:0")
os.environ.setdefault("TSG_DTYPE", "bf16")
from torchseg_amd.tuning import use_shipped_miopen_db
use_shipped_miopen_db(rank=0)

def make():
    model, opt, base_lr = bench.build_model(dev, a.batch, a.size, ProbOhemCrossEntropy2d, SyncBatchNorm, fused_sgd=a.opt == "fused")
    model = DistributedDataParallel(model); model.train()
    return model, opt, PolyLR(base_lr, 0.9, 80000)
imgs, gts = bench.synthetic_batch(dev, a.batch, a.size)

def stats(model, loss):
    tot = 0.0; bad_g = []; bad_p = []; big = (0.0, "")
    for n, p in model.named_parameters():
        if p.grad is None: continue
        g = p.grad.float()
        if not torch.isfinite(g).all(): bad_g.append(n)
        else:
            v = g.norm().item(); tot += v * v
            if v > big[0]: big = (v, n)
        if not torch.isfinite(p).all(): bad_p.append(n)
    bad_b = [n for n, b in model.named_buffers() if b.is_floating_point() and not torch.isfinite(b).all()]
    return f"loss {loss.item():.4f} |g| {tot ** 0.5:.3e} max {big[0]:.2e} ({big[1][-40:]}) nonfinite grads {len(bad_g)} {bad_g[:2]} params {len(bad_p)} {bad_p[:2]} buffers {len(bad_b)} {bad_b[:2]}"

# eager twin
model, opt, pol = make()
ref = []
for it in range(a.warm + a.steps):
    loss = bench.train_step(model, opt, imgs, gts, pol, it, 1)
    if it >= a.warm:
        torch.cuda.synchronize(); ref.append(stats(model, loss))
del model, opt
torch.cuda.empty_cache()

model, opt, pol = make()
side = {0: torch.cuda.current_stream(), 1: torch.cuda.Stream(), 2: bench.GraphedStep.capture_stream()}[a.side]
if a.side == 1:
    bench.GraphedStep.stream = torch.cuda.Stream()       # capture on a stream other than the warm-up's (the old behaviour)
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for it in range(a.warm):
        bench.train_step(model, opt, imgs, gts, pol, it, 1)
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
gr = bench.GraphedStep(model, opt, imgs, gts, 1)
for it in range(a.steps):
    bench.set_lr(opt, pol, a.warm + it)
    gr.graph.replay(); torch.cuda.synchronize()
    line = stats(model, gr.loss)
    opt.step(); torch.cuda.synchronize()
    print(f"[{a.tag}] step {it}: GRAPH {line}\n[{a.tag}]          EAGER {ref[it]}", flush=True)
