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
    loss = bench.train_step(model, opt, (imgs, gts), pol, it, 1)
    if it >= a.warm:
        torch.cuda.synchronize(); ref.append(stats(model, loss))
del model, opt
torch.cuda.empty_cache()

model, opt, pol = make()
Y = bench.GraphedStep.capture_stream()
D = torch.cuda.current_stream()
def warm(stream, its):
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        for it in its:
            bench.train_step(model, opt, (imgs, gts), pol, it, 1)
    torch.cuda.current_stream().wait_stream(stream); torch.cuda.synchronize()
if a.side == 0:   warm(D, range(a.warm))                                  # everything eager on the default stream
elif a.side == 1: warm(torch.cuda.Stream(), range(a.warm))               # warm-up on X, capture on Y
elif a.side == 2: warm(Y, range(a.warm))                                 # warm-up on Y, capture on Y
elif a.side == 3: warm(D, range(1)); warm(Y, range(1, a.warm))           # first step (lazy state) on D, rest on Y
elif a.side == 4: warm(Y, range(a.warm))                                 # as 2, and the replays + optimizer run on Y too
elif a.side == 5: warm(Y, range(a.warm - 1)); warm(D, range(a.warm - 1, a.warm))   # last eager step on D
gr = bench.GraphedStep(model, opt, (imgs, gts), 1)
run_stream = Y if a.side == 4 else D
for it in range(a.steps):
    with torch.cuda.stream(run_stream):
        bench.set_lr(opt, pol, a.warm + it)
        gr.graph.replay(); torch.cuda.synchronize()
        line = stats(model, gr.loss)
        opt.step(); torch.cuda.synchronize()
    print(f"[{a.tag}] step {it}: GRAPH {line}\n[{a.tag}]          EAGER {ref[it]}", flush=True)
