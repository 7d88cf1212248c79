"""GPU microbench of the 64 -> 64 3x3 forward at layer1's bench shape ([16, 64, 256, 256] bf16 channels_last): ours
(tsg_conv3x3_c64_fwd, with and without the statistics epilogue) against the vendor library's forward (F.conv2d, shipped
find-db); HIP-event timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from torchseg_amd import kernels as K
from torchseg_amd.tuning import use_shipped_miopen_db
use_shipped_miopen_db(0)
dev = torch.device("cuda:0")
kp = K.provider()
def timeit(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
for H in (256, 128):
    x = torch.randn(16, 64, H, H, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 64, 3, 3, device=dev) * 0.05).bfloat16().contiguous(memory_format=torch.channels_last)
    fl = 2.0 * 16 * 64 * 64 * 9 * H * H
    t0 = timeit(lambda: kp.conv3x3_c64_fwd(x, w))
    t1 = timeit(lambda: kp.conv3x3_c64_fwd(x, w, True))
    t2 = timeit(lambda: F.conv2d(x, w, None, 1, 1))
    d = (kp.conv3x3_c64_fwd(x, w).float() - F.conv2d(x, w, None, 1, 1).float()).abs().max().item()
    print("H=%d  ours %.1f us (%.2f PF, %.0f GB/s)  +stats %.1f us   vendor %.1f us (%.2f PF)   max |diff| %.3g" %
          (H, t0, fl / t0 / 1e9, 2 * x.numel() * 2 / t0 / 1e3, t1, t2, fl / t2 / 1e9, d), flush=True)

for H in (512, 256):                                     # SpatialPath.conv_3x3_1 / conv_3x3_2: stride 2
    x = torch.randn(16, 64, H, H, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 64, 3, 3, device=dev) * 0.05).bfloat16().contiguous(memory_format=torch.channels_last)
    dy = torch.randn(16, 64, H // 2, H // 2, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    wt = kp.conv3x3_weight_rot180_t(w)
    mb = (x.numel() + dy.numel()) * 2 / 1e6
    t0 = timeit(lambda: kp.conv3x3_c64_fwd(x, w, stride=2))
    t1 = timeit(lambda: kp.conv3x3_c64_fwd(x, w, True, stride=2))
    t2 = timeit(lambda: F.conv2d(x, w, None, 2, 1))
    t3 = timeit(lambda: kp.conv3x3_c64_s2_dgrad(dy, wt, (H, H)))
    t4 = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1,
                                                            [True, False, False]), 10)
    print("stride 2, H=%d  fwd ours %.1f us (%.0f GB/s) +stats %.1f  vendor %.1f us | dgrad ours %.1f us (%.0f GB/s)  vendor %.1f us" %
          (H, t0, mb / t0 * 1e3, t1, t2, t3, mb / t3 * 1e3, t4), flush=True)
