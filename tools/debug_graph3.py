"""hipGraph root-causing, part 2: does a replay of forward+backward (NO optimizer in the graph, none run between
replays) leave the weights alone, is it deterministic, and does it agree with eager?  One variant per process; toggles
through the TSG_* environment (feature bisect) and --warm default|side, --mode global|thread_local|relaxed."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
import bench
from torchseg_amd.syncbn import SyncBatchNorm
from torchseg_amd.losses import ProbOhemCrossEntropy2d
from torchseg_amd.ddp import DistributedDataParallel

ap = argparse.ArgumentParser()
ap.add_argument("--opt", default="fused"); ap.add_argument("--batch", type=int, default=2); ap.add_argument("--size", type=int, default=256)
ap.add_argument("--warm", default="side"); ap.add_argument("--mode", default="global"); ap.add_argument("--tag", default="")
a = ap.parse_args()
dev = torch.device("cuda:0")
os.environ.setdefault("TSG_DTYPE", "bf16")
model, opt, base_lr = bench.build_model(dev, a.batch, a.size, ProbOhemCrossEntropy2d, SyncBatchNorm, fused_sgd=a.opt == "fused")
model = DistributedDataParallel(model); model.train()
imgs, gts = bench.synthetic_batch(dev, a.batch, a.size)
params = list(model.named_parameters())

def eager_step(with_opt):
    opt.zero_grad()
    loss = model(imgs, gts); loss.backward()
    if with_opt: opt.step()
    return loss

if a.warm == "side":
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): eager_step(True)
    torch.cuda.current_stream().wait_stream(side)
else:
    for _ in range(3): eager_step(True)
torch.cuda.synchronize()

# eager reference gradients on the current weights
eager_step(False); torch.cuda.synchronize()
g_eager = {n: p.grad.detach().float().clone() for n, p in params}
g2 = None
eager_step(False); torch.cuda.synchronize()
eager_noise = max(((p.grad.float() - g_eager[n]).abs().max() / (g_eager[n].abs().max() + 1e-20)).item() for n, p in params)

graph = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(graph, capture_error_mode=a.mode):
    opt.zero_grad()
    loss_g = model(imgs, gts); loss_g.backward()
torch.cuda.synchronize()
W0 = {n: p.detach().clone() for n, p in params}
reps = []
for r in range(3):
    graph.replay(); torch.cuda.synchronize()
    changed = [(n, (p.detach().float() - W0[n].float()).abs().max().item(), W0[n].float().abs().max().item()) for n, p in params
               if not torch.equal(p.detach(), W0[n])]
    reps.append({n: p.grad.detach().float().clone() for n, p in params})
    bad = sorted([((reps[-1][n] - g_eager[n]).abs().max().item() / (g_eager[n].abs().max().item() + 1e-20), n) for n, _ in params],
                 reverse=True)
    print(f"[{a.tag}] replay {r}: loss {loss_g.item():.5f}  weights changed: {len(changed)} {changed[:2]}  "
          f"grads vs eager: worst {bad[0][0]:.2e} ({bad[0][1]}), #>2e-2: {sum(b[0] > 2e-2 for b in bad)}, "
          f"nonfinite params: {sum(not torch.isfinite(reps[-1][n]).all().item() for n, _ in params)}")
det = max(((reps[1][n] - reps[2][n]).abs().max() / (reps[2][n].abs().max() + 1e-20)).item() for n, _ in params)
print(f"[{a.tag}] replay-to-replay max rel diff {det:.2e}; eager-to-eager {eager_noise:.2e}")
worst = sorted([((reps[2][n] - g_eager[n]).abs().max().item() / (g_eager[n].abs().max().item() + 1e-20), n) for n, _ in params], reverse=True)[:6]
print(f"[{a.tag}] worst params (replay vs eager):", [(f"{v:.1e}", n) for v, n in worst])
