"""Per-layer, per-direction cost of the MIOpen convolutions of BiSeNet-R18 at the bench shape (bf16, channels_last, the
shipped find-db in immediate mode, exactly what bench.py runs), next to each direction's HBM and MFMA floors:
where is the slack?   python tools/probe_conv2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchseg_amd.tuning import use_shipped_miopen_db
use_shipped_miopen_db(rank=0)
torch.backends.cudnn.benchmark = False
dev = torch.device("cuda:0")
B = 16
# (name, count per step, cin, cout, k, stride, Hin, needs_dgrad)
L = [("S1 sp.conv_3x3_1", 1, 64, 64, 3, 2, 512, True), ("S2 sp.conv_3x3_2", 1, 64, 64, 3, 2, 256, True),
     ("S3 sp.conv_1x1", 1, 64, 128, 1, 1, 128, True), ("L1 layer1 3x3", 4, 64, 64, 3, 1, 256, True),
     ("L2.0 3x3 s2", 1, 64, 128, 3, 2, 256, True), ("L2 3x3", 3, 128, 128, 3, 1, 128, True), ("L2 ds 1x1 s2", 1, 64, 128, 1, 2, 256, True),
     ("L3.0 3x3 s2", 1, 128, 256, 3, 2, 128, True), ("L3 3x3", 3, 256, 256, 3, 1, 64, True), ("L3 ds", 1, 128, 256, 1, 2, 128, True),
     ("L4.0 3x3 s2", 1, 256, 512, 3, 2, 64, True), ("L4 3x3", 3, 512, 512, 3, 1, 32, True), ("L4 ds", 1, 256, 512, 1, 2, 64, True),
     ("ARM32 3x3", 1, 512, 128, 3, 1, 32, True), ("ARM16 3x3", 1, 256, 128, 3, 1, 64, True),
     ("refine32->64 3x3", 1, 128, 128, 3, 1, 64, True), ("refine 3x3 @128", 1, 128, 128, 3, 1, 128, True),
     ("head0 3x3 128->256 @64", 1, 128, 256, 3, 1, 64, True), ("head1 3x3 128->256 @128", 1, 128, 256, 3, 1, 128, True),
     ("head2 3x3 256->64 @128", 1, 256, 64, 3, 1, 128, True), ("FFM 1x1 256->256 @128", 1, 256, 256, 1, 1, 128, True),
     ("cls 1x1 256->19 @64", 1, 256, 19, 1, 1, 64, True), ("cls 1x1 256->19 @128", 1, 256, 19, 1, 1, 128, True), ("cls 1x1 64->19 @128", 1, 64, 19, 1, 1, 128, True)]

def timeit(fn, n=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3

tot = [0.0, 0.0, 0.0, 0.0]
print("%-26s %2s %9s %9s %9s | floors: mem us (x / y), mfma us @2.5PF" % ("layer", "n", "fwd us", "dgrad us", "wgrad us"))
for name, cnt, cin, cout, k, s, H, dg in L:
    p = k // 2
    x = torch.randn(B, cin, H, H, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, k, k, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    y = torch.ops.aten.convolution(x, w, None, [s, s], [p, p], [1, 1], False, [0, 0], 1)
    dy = torch.randn_like(y)
    f = timeit(lambda: torch.ops.aten.convolution(x, w, None, [s, s], [p, p], [1, 1], False, [0, 0], 1))
    d = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [s, s], [p, p], [1, 1], False, [0, 0], 1, [True, False, False]))
    g = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [s, s], [p, p], [1, 1], False, [0, 0], 1, [False, True, False]))
    xb, yb = x.numel() * 2 / 1e6, y.numel() * 2 / 1e6
    fl = 2.0 * B * cin * cout * k * k * (H // s) ** 2
    print("%-26s %2d %9.1f %9.1f %9.1f | %6.1f (%5.0f MB / %5.0f MB)  %6.1f   fwd %.2f PF" %
          (name, cnt, f, d, g, (xb + yb) / 8.0, xb, yb, fl / 2.5e15 * 1e6, fl / f / 1e9), flush=True)
    tot[0] += cnt * f; tot[1] += cnt * d; tot[2] += cnt * g; tot[3] += cnt * (xb + yb) / 8.0
print("per step: fwd %.2f ms, dgrad %.2f ms, wgrad %.2f ms (incl. MIOpen's own zero-fill / cast helpers); memory floor per direction %.2f ms"
      % (tot[0] / 1e3, tot[1] / 1e3, tot[2] / 1e3, tot[3] / 1e3))
