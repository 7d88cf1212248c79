"""debug: which BatchNorm buffers differ between the reference network.py and the native builder after one bf16 step"""
import json, os, sys, torch, torch.nn as nn
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import stage_reference
net_mod, config, _ = stage_reference.import_experiment("bisenet", "cityscapes.bisenet.R18")
from torchseg_amd import fusion, kernels as K
from torchseg_amd.tuning import use_shipped_miopen_db
use_shipped_miopen_db()
from torchseg_amd.ddp import DistributedDataParallel
from torchseg_amd.losses import ProbOhemCrossEntropy2d
from torchseg_amd.syncbn import SyncBatchNorm
from torchseg_amd.workloads.bisenet import BiSeNet as Native
from utils.init_func import init_weight
B, S = int(os.environ.get("DB", 16)), int(os.environ.get("DS", 1024))
dev = torch.device("cuda:0")
def build(cls):
    torch.manual_seed(config.seed)
    crit = ProbOhemCrossEntropy2d(ignore_label=255, thresh=0.7, min_kept=B * S * S // 16, use_weight=False)
    m = cls(config.num_classes, is_training=True, criterion=crit, pretrained_model=None, norm_layer=SyncBatchNorm)
    init_weight(m.business_layer, nn.init.kaiming_normal_, SyncBatchNorm, config.bn_eps, config.bn_momentum, mode='fan_in', nonlinearity='relu')
    return DistributedDataParallel(m.to(dev), compute_dtype=torch.bfloat16)
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 3, S, S, generator=g).to(dev)
y = torch.randint(0, 19, (B, S, S), generator=g); y[:, :8] = 255; y = y.to(dev)
def run(cls):
    net = build(cls); net.train()
    loss = net(x, y); loss.backward(); torch.cuda.synchronize()
    return loss.item(), {k: v.detach().float().cpu() for k, v in net.module.named_buffers()}, {k: p.grad.float().cpu() for k, p in net.module.named_parameters()}
runs = [("nat0", Native), ("nat1", Native), ("ref0", net_mod.BiSeNet), ("ref1", net_mod.BiSeNet), ("nat2", Native)]
res = {n: run(c) for n, c in runs}
for a, b in (("nat0", "nat1"), ("nat1", "nat2"), ("ref0", "ref1"), ("nat1", "ref0")):
    la, ba, ga = res[a]; lb, bb, gb = res[b]
    diff = [k for k in ba if not torch.equal(ba[k], bb[k])]
    gd = [k for k in ga if not torch.equal(ga[k], gb[k])]
    print(a, b, "loss", la, lb, "buffers differing:", len(diff), diff[:6], "grads differing:", len(gd), gd[:4])
