"""Weight gradient of every stride-1 3x3 convolution of BiSeNet-R18 at the bench shape (bf16 channels_last): ours
(tsg_conv3x3_wrw / _gen, incl. the fold) against MIOpen's backward-filter incl. its zero fill and cast
(aten::convolution_backward, weight gradient only, shipped find-db); HIP-event timing on the launch stream."""
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


# (name, count per step, Cin, Cout, Hin[, stride])
LAYERS = [("sp.conv_3x3_1 s2", 1, 64, 64, 512, 2), ("sp.conv_3x3_2 s2", 1, 64, 64, 256, 2), ("layer2.0 s2", 1, 64, 128, 256, 2),
          ("layer3.0 s2", 1, 128, 256, 128, 2), ("layer4.0 s2", 1, 256, 512, 64, 2), ("layer1", 4, 64, 64, 256), ("layer2", 3, 128, 128, 128), ("refine @128", 1, 128, 128, 128),
          ("layer3", 3, 256, 256, 64), ("layer4", 3, 512, 512, 32), ("head1 128->256 @128", 1, 128, 256, 128),
          ("head2 256->64 @128", 1, 256, 64, 128), ("head0 128->256 @64", 1, 128, 256, 64), ("arm16 256->128 @64", 1, 256, 128, 64),
          ("arm32 512->128 @32", 1, 512, 128, 32), ("refine @64", 1, 128, 128, 64)]
only = os.environ.get("ONLY")
tot_o = tot_m = 0.0
for row in LAYERS:
    name, cnt, cin, cout, H = row[:5]
    st = row[5] if len(row) > 5 else 1
    if only and only not in name:
        continue
    x = torch.randn(16, cin, H, H, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    dy = torch.randn(16, cout, H // st, H // st, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 3, 3, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    t = timeit(lambda: kp.conv3x3_wrw(x, dy, stride=st))
    t2 = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [st, st], [1, 1], [1, 1], False, [0, 0], 1,
                                                            [False, True, False]), 10)
    a = kp.conv3x3_wrw(x, dy, stride=st)
    b = torch.ops.aten.convolution_backward(dy, x, w, None, [st, st], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    fl = 2.0 * 16 * cin * cout * 9 * (H // st) ** 2
    print("%-22s x%d  ours %7.1f us (%.2f PF)   MIOpen %7.1f us   rel diff %.1e" %
          (name, cnt, t, fl / t / 1e9, t2, ((a - b.float()).norm() / b.float().norm()).item()), flush=True)
    tot_o += cnt * t; tot_m += cnt * t2
print("per step: ours %.2f ms, MIOpen %.2f ms" % (tot_o / 1e3, tot_m / 1e3))
