"""Classifier-head kernels (csrc/clshead.hip) at the three BiSeNet head shapes, bf16: forward / data gradient / weight
gradient in us and GB/s of the algorithmic bytes, next to the vendor convolution + its layout copy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from torchseg_amd import kernels as K
kp = K.provider(); dev = torch.device("cuda:0")


def timeit(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3


for (B, C, S) in [(16, 256, 128), (16, 64, 128), (16, 256, 64)]:
    x = torch.randn(B, C, S, S, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn(19, C, 1, 1, device=dev) * 0.05
    b = torch.randn(19, device=dev)
    dz = torch.randn(B, 19, S, S, device=dev).bfloat16()
    wb = w.bfloat16()
    tf = timeit(lambda: kp.cls_head_fwd(x, w, b))
    td = timeit(lambda: kp.cls_head_bwd(dz, x, w, need_dx=True, need_db=False)[0])
    tb = timeit(lambda: kp.cls_head_bwd(dz, x, w, need_dx=False, need_db=True))
    vf = timeit(lambda: F.conv2d(x, wb, None).contiguous())
    dzc = dz.contiguous(memory_format=torch.channels_last)
    vb = timeit(lambda: torch.ops.aten.convolution_backward(dzc, x, wb, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, True, False]))
    mb = x.numel() * 2 / 1e6
    print("%dx%dx%d^2 -> 19: fwd %6.1f us (%5.0f GB/s)  dgrad %6.1f us (%5.0f GB/s)  wgrad+dbias %6.1f us (%5.0f GB/s) | vendor fwd+copy %6.1f  bwd %6.1f"
          % (B, C, S, tf, mb / tf * 1e3, td, mb / td * 1e3, tb, mb / tb * 1e3, vf, vb), flush=True)
