"""GPU microbench of the data gradients that emit the BatchNorm backward sums (round 6): plain launch, fused launch, and the
separate tsg_bn_bwd_reduce pass the fused launch replaces — at the bench shapes of BiSeNet-R18 (16 x 64 x 256^2 stride 1;
512^2 / 256^2 inputs of the stride-2 SpatialPath layers)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchseg_amd import kernels as K
dev = torch.device("cuda:0")
kp = K.provider()
def timeit(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
cl = dict(memory_format=torch.channels_last)
for stride, H in ((1, 256), (2, 512), (2, 256)):
    OH = (H - 1) // stride + 1
    x = torch.randn(16, 64, H, H, device=dev).bfloat16().contiguous(**cl)
    dy = torch.randn(16, 64, OH, OH, device=dev).bfloat16().contiguous(**cl)
    w = (torch.randn(64, 64, 3, 3, device=dev) * 0.05).bfloat16().contiguous(**cl)
    rot = kp.conv3x3_weight_rot180_t(w)
    fp = torch.stack([torch.rand(64) + 0.5, torch.randn(64) * 0.3, torch.randn(64) * 0.1]).to(dev).contiguous()
    layout, N, C, HW = K.bn_layout(x)
    if stride == 1:
        plain = lambda: kp.conv3x3_c64_fwd(dy, rot)
        fused = lambda: kp.conv3x3_c64_fwd(dy, rot, bsum=(x, fp))
    else:
        plain = lambda: kp.conv3x3_c64_s2_dgrad(dy, rot, (H, H))
        fused = lambda: kp.conv3x3_c64_s2_dgrad(dy, rot, (H, H), bsum=(x, fp))
    dx = plain()
    red = lambda: kp.bn_bwd_reduce(dx, x, None, layout, N, C, HW, fp, True)
    t0, t1, t2 = timeit(plain), timeit(fused), timeit(red)
    mb = (x.numel() + dy.numel()) * 2 / 1e6
    print("stride %d, %d^2: data gradient %.1f us, with the BN sums %.1f us (%.0f GB/s incl. x), separate reduce pass %.1f us -> %.1f us saved"
          % (stride, H, t0, t1, (mb + x.numel() * 2 / 1e6) / t1 * 1e3, t2, t0 + t2 - t1), flush=True)
