"""Where does the fp32 logits gap to the CPU oracle at 1024^2 come from?  CPU oracle vs (a) stock PyTorch-ROCm ops on the
GPU (nn.BatchNorm2d, MIOpen convs, ATen upsample) with allow_tf32 on/off, (b) our HIP path; also bf16 autocast, stock vs ours."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from torchseg_amd import workloads
from torchseg_amd.workloads.bisenet import BiSeNet
from torchseg_amd.ddp import DistributedDataParallel, apply_channels_last
from torchseg_amd.syncbn import SyncBatchNorm
B, S, C = 2, int(os.environ.get("SIZE", "1024")), 19
dev = torch.device("cuda:0")
torch.manual_seed(12345)
ref = BiSeNet(C, True, None, None, nn.BatchNorm2d); ref.train()
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 3, S, S, generator=g)
with torch.no_grad():
    want = [t.clone() for t in ref.logits(x)]
def report(tag, got):
    out = []
    for a, b in zip(got, want):
        d = a.float().cpu() - b
        out.append("max %.2e rms-rel %.2e" % (d.abs().max().item(), (d.pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()))
    print(f"{tag:34s}", " | ".join(out), " scale", ["%.2f" % b.abs().max().item() for b in want], flush=True)
def stock(dtype, tf32, cl):
    torch.backends.cudnn.allow_tf32 = tf32; torch.backends.cuda.matmul.allow_tf32 = tf32
    workloads.NATIVE_FUSIONS = False
    torch.manual_seed(12345)
    m = BiSeNet(C, True, None, None, nn.BatchNorm2d); m.load_state_dict(ref.state_dict()); m = m.to(dev).train()
    if cl: apply_channels_last(m)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
        got = m.logits(x.to(dev))
    workloads.NATIVE_FUSIONS = True
    return got
def ours(dtype):
    torch.manual_seed(12345)
    m = BiSeNet(C, True, None, None, SyncBatchNorm); m.load_state_dict(ref.state_dict())
    m = DistributedDataParallel(m.to(dev), compute_dtype=dtype).train()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
        return m.module.logits(x.to(dev))
print("allow_tf32 defaults:", torch.backends.cudnn.allow_tf32, torch.backends.cuda.matmul.allow_tf32, " MIOPEN env:", {k: v for k, v in os.environ.items() if "MIOPEN" in k})
report("stock fp32 tf32=default nchw", stock(torch.float32, torch.backends.cudnn.allow_tf32, False))
report("stock fp32 tf32=False nchw", stock(torch.float32, False, False))
report("stock fp32 tf32=False channels_last", stock(torch.float32, False, True))
report("ours  fp32", ours(torch.float32))
if os.environ.get("DIAG_BF16", "1") == "1":
    report("ours  bf16", ours(torch.bfloat16))
    report("stock bf16 channels_last", stock(torch.bfloat16, False, True))
