"""tsg_conv3x3_wrw at ResNet-18 layer1's geometry (16 x 64 x 256^2 bf16 channels_last) against MIOpen's
backward-filter (aten::convolution_backward, weight gradient only); HIP-event timing on the launch stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchseg_amd import kernels as K
from torchseg_amd.tuning import use_shipped_miopen_db
use_shipped_miopen_db(0)
dev = torch.device("cuda:0")
kp = K.provider()
x = torch.randn(16, 64, 256, 256, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
dy = torch.randn(16, 64, 256, 256, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
w = torch.randn(64, 64, 3, 3, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


nbytes = (x.numel() + dy.numel()) * 2
for variant in ("tr", "v1"):
    t = timeit(lambda: kp.conv3x3_wrw(x, dy, variant=variant))
    print("conv3x3 wrw ours (%s) %.1f us  %.0f GB/s algorithmic" % (variant, t, nbytes / t / 1e3))
t2 = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                        [False, True, False]), 20)
print("conv3x3 wrw MIOpen %.1f us (incl. its zero fill / cast)" % t2)
a = kp.conv3x3_wrw(x, dy)
b = torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
print("rel diff vs MIOpen %.2e" % ((a - b.float()).norm() / b.float().norm()).item())
