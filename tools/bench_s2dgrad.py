"""Data gradient of the three stride-2 3x3 layers of ResNet-18 at the bench shape (16 x 1024^2 crops): tsg_conv3x3_s2_dgrad
(incl. its filter preparation) with and without the shortcut addend, against the vendor library's backward-data
(aten::convolution_backward, data gradient only) plus the accumulation add it needs; HIP-event timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchseg_amd import kernels as K
from torchseg_amd.tuning import use_shipped_miopen_db
use_shipped_miopen_db(0)
dev = torch.device("cuda:0")
kp = K.provider()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for name, Cin, Cout, S in [("layer2.0", 64, 128, 256), ("layer3.0", 128, 256, 128), ("layer4.0", 256, 512, 64)]:
    x = torch.randn(16, Cin, S, S, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * (2.0 / (9 * Cout)) ** 0.5).contiguous(memory_format=torch.channels_last)
    wb = w.bfloat16().contiguous(memory_format=torch.channels_last)
    dy = torch.randn(16, Cout, S // 2, S // 2, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    skip = torch.randn_like(x)
    fl = 2.0 * 9 * Cin * dy.numel()
    t0 = timeit(lambda: kp.conv3x3_s2_dgrad(dy, wb, (S, S)))
    t1 = timeit(lambda: kp.conv3x3_s2_dgrad(dy, wb, (S, S), addend=skip))
    lib = lambda: torch.ops.aten.convolution_backward(dy, x, wb, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]
    t2 = timeit(lib, 10)
    t3 = timeit(lambda: lib().add_(skip), 10)
    d = (kp.conv3x3_s2_dgrad(dy, wb, (S, S)).float() - lib().float()).abs().max().item()
    print("%-9s %3d -> %3d @%3d  ours %6.1f us (%.2f PF)  +addend %6.1f | vendor %6.1f  + add %6.1f | max|diff| %.4f" %
          (name, Cin, Cout, S, t0, fl / t0 / 1e9, t1, t2, t3, d), flush=True)
