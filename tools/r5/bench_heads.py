"""fused head kernels at the bench shapes (bf16 logits, uint8 labels): forward / backward time per launch"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from torchseg_amd import kernels as K
kp = K.provider()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
B, C, OH = 16, 19, 1024
t = torch.randint(0, C, (B, OH, OH), device=dev, generator=g).to(torch.uint8); t[:, :8] = 255
k = B * OH * OH // 16
def timeit(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
for IH in (128, 64):
    z = torch.randn(B, C, IH, IH, device=dev, generator=g).to(torch.bfloat16)
    out = kp.ohem_up_fwd(z, t, OH, OH, 255, 0.7, k, None)
    loss, nll, lse, sel = out[0], out[1], out[2], out[3]
    gs = torch.ones(1, device=dev)
    tf = timeit(lambda: kp.ohem_up_fwd(z, t, OH, OH, 255, 0.7, k, None))
    tb = timeit(lambda: kp.ohem_up_bwd(z, t, OH, OH, 255, None, nll, lse, sel, gs))
    dz = kp.ohem_up_bwd(z, t, OH, OH, 255, None, nll, lse, sel, gs)
    print("IH %3d: fwd (incl. selection tail) %7.1f us   bwd %7.1f us   loss %.6f  |dz| %.6e  FWD=%s NT=%s" % (
        IH, tf, tb, float(loss), float(dz.float().norm()), os.environ.get("TSG_HEAD_FWD", "2"), os.environ.get("TSG_HEAD_BWD_NT", "-")))
