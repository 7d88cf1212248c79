"""Which component breaks under hipGraph capture/replay?  Compares eager vs replayed outputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn, torch.nn.functional as F
import bench
from torchseg_amd.syncbn import SyncBatchNorm
from torchseg_amd.losses import ProbOhemCrossEntropy2d
from torchseg_amd.ddp import DistributedDataParallel
from torchseg_amd import syncbn
dev = torch.device("cuda:0")

def run_case(name, fn, n_warm=3):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(n_warm):
            ref = fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            out = fn()
        g.replay(); torch.cuda.synchronize()
        r1 = [o.float().clone() for o in out]
        g.replay(); torch.cuda.synchronize()
        r2 = [o.float() for o in out]
        eager = [o.float() for o in fn()]
        print(f"{name:28s} replay finite={[bool(torch.isfinite(t).all()) for t in r2]} "
              f"replay={[round(t.flatten()[0].item(), 5) for t in r2]} eager={[round(t.flatten()[0].item(), 5) for t in eager]}")
    except Exception as e:
        print(f"{name:28s} EXC {type(e).__name__}: {str(e)[:200]}")

torch.manual_seed(0)
x = torch.randn(4, 64, 64, 64, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
bn = SyncBatchNorm(64).to(dev)
def f_bn():
    xx = x.clone().requires_grad_(True)
    y = bn(xx, relu=True); y.float().mean().backward()
    return [y.float().mean(), xx.grad.float().abs().mean()]
run_case("syncbn nhwc fwd+bwd", f_bn)

logits = torch.randn(2, 19, 256, 256, device=dev)
lab = torch.randint(0, 19, (2, 256, 256), device=dev)
crit = ProbOhemCrossEntropy2d(255, thresh=0.7, min_kept=2 * 256 * 256 // 16)
def f_ohem():
    l = logits.clone().requires_grad_(True)
    loss = crit(l, lab); loss.backward()
    return [loss, l.grad.abs().mean()]
run_case("ohem fwd+bwd", f_ohem)

from torchseg_amd.upsample import install_aten_overrides
install_aten_overrides()
z = torch.randn(2, 19, 32, 32, device=dev)
def f_up():
    zz = z.clone().requires_grad_(True)
    y = F.interpolate(zz, scale_factor=8, mode="bilinear", align_corners=True); y.mean().backward()
    return [y.mean(), zz.grad.abs().mean()]
run_case("upsample fwd+bwd", f_up)

conv = nn.Conv2d(64, 64, 3, padding=1, bias=False).to(dev).to(memory_format=torch.channels_last)
def f_conv():
    xx = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = conv(xx)
    y.float().mean().backward()
    return [y.float().mean(), conv.weight.grad.abs().mean()]
run_case("miopen conv autocast", f_conv)

from torchseg_amd.pool import GlobalAvgPool, channel_scale, MaxPool2d
def f_pool():
    xx = x.clone().requires_grad_(True)
    g = torch.sigmoid(GlobalAvgPool(1)(xx)); y = channel_scale(xx, g, True); y = MaxPool2d(3, 2, 1)(y)
    y.float().mean().backward()
    return [y.float().mean(), xx.grad.float().abs().mean()]
run_case("gap+chanscale+maxpool", f_pool)

# whole model, forward+backward only, then with optimizer
model, opt, base_lr = bench.build_model(dev, 4, 256, ProbOhemCrossEntropy2d, SyncBatchNorm, fused_sgd=True)
model = DistributedDataParallel(model); model.train()
imgs, gts = bench.synthetic_batch(dev, 4, 256)
def f_model():
    opt.zero_grad()
    loss = model(imgs, gts); loss.backward()
    gn = sum(p.grad.float().abs().sum() for p in model.parameters() if p.grad is not None)
    return [loss, gn]
run_case("bisenet fwd+bwd", f_model)
def f_step():
    opt.zero_grad()
    loss = model(imgs, gts); loss.backward(); opt.step()
    return [loss]
run_case("bisenet full step", f_step)
