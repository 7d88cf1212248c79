"""tsg_stem_conv_fwd/_wrw at the BASELINE geometry (16 x 3 x 1024^2 bf16) against MIOpen's convolution
under autocast; HIP-event timing on the launch stream, algorithmic bytes = x + y (fwd), x + dy (wgrad)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from torchseg_amd import kernels as K
from torchseg_amd.tuning import use_shipped_miopen_db
use_shipped_miopen_db(0)
dev = torch.device("cuda:0")
kp = K.provider()
B, S = 16, 1024
x = torch.randn(B, 3, S, S, device=dev).bfloat16()
w = (torch.randn(64, 3, 7, 7, device=dev) * 0.05).requires_grad_()
dy = torch.randn(B, 64, S // 2, S // 2, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


wd = w.detach()
tf = timeit(lambda: kp.stem_conv_fwd(x, wd))
tw = timeit(lambda: kp.stem_conv_wrw(x, dy))
bx, by = x.numel() * 2, dy.numel() * 2
print("stem fwd  %.1f us  %.0f GB/s algorithmic" % (tf, (bx + by) / tf / 1e3))
print("stem wrw  %.1f us  %.0f GB/s algorithmic" % (tw, (bx + by) / tw / 1e3))


def stock():
    w.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = F.conv2d(x, w, None, 2, 3)
    y.backward(dy)


print("MIOpen fwd+wrw (nchw, autocast) %.1f us" % timeit(stock, 10))
