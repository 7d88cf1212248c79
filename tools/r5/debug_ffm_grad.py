"""where does the 1e-3 relative error of the spatial path's gradients come from (fp32 parity mode vs float64)?"""
import os, sys, torch, torch.nn as nn
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.ohem_ref import ProbOhemCrossEntropy2d as OracleOhem
from torchseg_amd.ddp import DistributedDataParallel
from torchseg_amd.losses import ProbOhemCrossEntropy2d
from torchseg_amd.syncbn import SyncBatchNorm
from torchseg_amd.workloads.bisenet import BiSeNet
dev = torch.device("cuda:0")
B, S = 4, 256
mk = B * S * S // 16
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 3, S, S, generator=g); y = torch.randint(0, 19, (B, S, S), generator=g); y[:, :8] = 255
torch.manual_seed(12345)
ref = BiSeNet(19, True, OracleOhem(255, thresh=0.7, min_kept=mk), None, nn.BatchNorm2d)
sd = {k: v.clone() for k, v in ref.state_dict().items()}
ref = ref.double()
net = BiSeNet(19, True, ProbOhemCrossEntropy2d(255, thresh=0.7, min_kept=mk), None, SyncBatchNorm)
net.load_state_dict(sd)
net = DistributedDataParallel(net.to(dev), compute_dtype=torch.float32)
names = ["spatial_path", "ffm", "ffm.conv_1x1", "ffm.channel_attention", "heads.2", "heads.2.conv_3x3", "heads.2.conv_1x1", "refines.1", "arms.1",
         "ffm.conv_1x1.conv", "ffm.conv_1x1.bn", "spatial_path.conv_1x1.bn", "spatial_path.conv_1x1.conv"]
def instrument(model, store):
    mods = dict(model.named_modules())
    for n in names:
        def fh(m, inp, out, n=n):
            if isinstance(out, torch.Tensor) and out.requires_grad:
                store["out:" + n] = out.detach().double().cpu()
                out.register_hook(lambda gr, n=n: store.__setitem__("grad:" + n, gr.detach().double().cpu()))
        mods[n].register_forward_hook(fh)
sr, sn = {}, {}
instrument(ref, sr); instrument(net.module, sn)
ref(x.double(), y).backward()
net(x.to(dev), y.to(dev)).backward()
torch.cuda.synchronize()
for k in sorted(sr):
    if k in sn:
        a, b = sn[k], sr[k]
        print("%-34s rel-L2 %.2e  max|d|/max %.2e   |ref| %.3e" % (k, float((a - b).norm() / b.norm()), float((a - b).abs().max() / b.abs().max()), float(b.norm())))
    else:
        print(k, "not captured on the GPU net")

print("---- isolated: SyncBatchNorm(+ReLU) backward of spatial_path.conv_1x1.bn on the captured float64 tensors")
import torch.nn.functional as F
xin = sr["out:spatial_path.conv_1x1.conv"]; dyin = sr["grad:spatial_path"]
bnr = ref.spatial_path.conv_1x1.bn
xr = xin.clone().requires_grad_(True)
yr = F.relu(F.batch_norm(xr, None, None, bnr.weight.detach().double(), bnr.bias.detach().double(), True, 0.1, 1e-5))
yr.backward(dyin)
for fmt in (torch.channels_last, torch.contiguous_format):
    for sliced in (False, True):
        bn = SyncBatchNorm(128).to(dev); bn.train()
        with torch.no_grad():
            bn.weight.copy_(bnr.weight.float()); bn.bias.copy_(bnr.bias.float())
        xg = xin.float().to(dev).contiguous(memory_format=fmt).requires_grad_(True)
        yg = bn(xg, relu=True)
        if sliced:
            big = torch.zeros(B, 256, dyin.shape[2], dyin.shape[3], device=dev).contiguous(memory_format=torch.channels_last)
            big[:, :128] = dyin.float().to(dev)
            dyg = big[:, :128]
        else:
            dyg = dyin.float().to(dev).contiguous(memory_format=fmt)
        yg.backward(dyg)
        d = xg.grad.double().cpu() - xr.grad
        print("fmt", "cl" if fmt == torch.channels_last else "nchw", "sliced dy" if sliced else "dense dy", "dx rel-L2 %.2e max %.2e ; dgamma %.2e dbeta %.2e" % (
            float(d.norm() / xr.grad.norm()), float(d.abs().max() / xr.grad.abs().max()),
            float((bn.weight.grad.double().cpu() - 0).norm()), float(bn.bias.grad.double().cpu().norm())))
# the same with torch's own fp32 batch_norm on the GPU and on the CPU
for devn in ("cuda:0", "cpu"):
    xt = xin.float().to(devn).requires_grad_(True)
    yt = F.relu(F.batch_norm(xt, None, None, bnr.weight.detach().float().to(devn), bnr.bias.detach().float().to(devn), True, 0.1, 1e-5))
    yt.backward(dyin.float().to(devn))
    d = xt.grad.double().cpu() - xr.grad
    print("torch fp32 batch_norm on", devn, "dx rel-L2 %.2e max %.2e" % (float(d.norm() / xr.grad.norm()), float(d.abs().max() / xr.grad.abs().max())))
a, b = sn["out:spatial_path"], sr["out:spatial_path"]
flip = ((a > 0) != (b > 0))
print("ReLU sign flips at spatial_path output:", int(flip.sum()), "of", flip.numel(), "; values there:", a[flip][:5].tolist(), b[flip][:5].tolist())
