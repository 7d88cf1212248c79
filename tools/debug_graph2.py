"""Root-causing the hipGraph divergence: per step, gradients produced by the REPLAY of the captured
forward+backward against gradients of an EAGER forward+backward on the very same weights, per parameter, under each
optimizer variant.  Prints the first parameters that disagree and the loss trajectories.
    python tools/debug_graph2.py [--opt torch|fused] [--steps 10] [--batch 4 --size 256]"""
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
ap.add_argument("--opt", default="torch"); ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--batch", type=int, default=4); ap.add_argument("--size", type=int, default=256)
ap.add_argument("--inside", type=int, default=0, help="capture optimizer.step() inside the graph")
a = ap.parse_args()
dev = torch.device("cuda:0")
os.environ["TSG_DTYPE"] = "bf16"
model, opt, base_lr = bench.build_model(dev, a.batch, a.size, ProbOhemCrossEntropy2d, SyncBatchNorm, fused_sgd=a.opt == "fused")
model = DistributedDataParallel(model); model.train()
imgs, gts = bench.synthetic_batch(dev, a.batch, a.size)
pol = PolyLR(base_lr, 0.9, 100)
params = [(n, p) for n, p in model.named_parameters()]

side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for it in range(3):
        bench.train_step(model, opt, imgs, gts, pol, it, 1)
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()

g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    loss_g = bench.step_body(model, opt, imgs, gts, 1, with_optimizer=bool(a.inside))
torch.cuda.synchronize()
graph_grads = [p.grad for _, p in params]
print("params with grad after capture:", sum(gg is not None for gg in graph_grads), "of", len(params))
bufs = [b for b in model.buffers()]

for it in range(a.steps):
    bench.set_lr(opt, pol, 3 + it)
    w_before = [p.detach().clone() for _, p in params]
    b_before = [b.detach().clone() for b in bufs]
    g.replay(); torch.cuda.synchronize()
    lg = loss_g.item()
    gr = [None if gg is None else gg.detach().clone() for gg in graph_grads]
    w_after_replay = [p.detach().clone() for _, p in params]
    if not a.inside:
        moved = [n for (n, p), w0, w1 in zip(params, w_before, w_after_replay) if not torch.equal(w0, w1)]
        if moved: print("  !! weights changed by a replay WITHOUT optimizer:", moved[:5])
    # eager fwd/bwd on the weights the replay saw
    with torch.no_grad():
        for (_, p), w0 in zip(params, w_before): p.copy_(w0)
        for b, b0 in zip(bufs, b_before): b.copy_(b0)
    for _, p in params: p.grad = None
    le = model(imgs, gts); le.backward(); torch.cuda.synchronize()
    worst = []
    for (n, p), r in zip(params, gr):
        if r is None or p.grad is None: continue
        d = (r.float() - p.grad.float()).abs().max().item(); s = p.grad.float().abs().max().item() + 1e-12
        if d / s > 2e-2: worst.append((d / s, n))
    worst.sort(reverse=True)
    print(f"step {it}: loss replay {lg:.5f} eager {le.item():.5f}  params off by >2e-2: {len(worst)}  {worst[:4]}")
    # restore the graph's grad tensors, the post-replay weights, then the optimizer under test
    with torch.no_grad():
        for (_, p), w1 in zip(params, w_after_replay): p.copy_(w1)
    for (_, p), gg in zip(params, graph_grads): p.grad = gg
    if not a.inside:
        opt.step()
torch.cuda.synchronize()
print("done")
