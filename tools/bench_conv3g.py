"""GPU microbench of the general stride-1 3x3 forward (tsg_conv3x3_gen_fwd, csrc/conv3g.hip) at BiSeNet-R18's layer
shapes of BASELINE config 2 (batch 16 x 1024^2 crops; bf16 channels_last), forward and data gradient (the same kernel on
the mode-1 filter), against the vendor library's forward (F.conv2d) and backward-data kernels; HIP-event timing.
TSG_CONV3G_BLOCKS selects the number of persistent blocks."""
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
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


# (name, launches per step fwd, Cin, Cout, H)
LAYERS = [("layer2 3x3", 3, 128, 128, 128), ("layer3 3x3", 3, 256, 256, 64), ("layer4 3x3", 3, 512, 512, 32),
          ("refine @128", 1, 128, 128, 128), ("refine @64", 1, 128, 128, 64), ("ARM16", 1, 256, 128, 64),
          ("ARM32", 1, 512, 128, 32), ("head0 @64", 1, 128, 256, 64), ("head1 @128", 1, 128, 256, 128),
          ("head2 @128", 1, 256, 64, 128)]
only = os.environ.get("LAYER")
tot = {"ours_f": 0.0, "vend_f": 0.0, "ours_d": 0.0, "vend_d": 0.0}
for name, n, Cin, Cout, H in LAYERS:
    if only and only not in name:
        continue
    x = torch.randn(16, Cin, H, H, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * (2.0 / (9 * Cin)) ** 0.5).contiguous(memory_format=torch.channels_last)
    wb = w.bfloat16().contiguous(memory_format=torch.channels_last)
    dy = torch.randn(16, Cout, H, H, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    wf, wt = kp.conv3x3_gen_prep_filter(w, 0, x), kp.conv3x3_gen_prep_filter(w, 1, dy)
    fl = 2.0 * 16 * Cin * Cout * 9 * H * H
    t_prep = timeit(lambda: kp.conv3x3_gen_prep_filter(w, 0, x))
    t0 = timeit(lambda: kp.conv3x3_gen_fwd(x, wf, Cout))
    t1 = timeit(lambda: kp.conv3x3_gen_fwd(x, wf, Cout, with_stats=True))
    t2 = timeit(lambda: F.conv2d(x, wb, None, 1, 1))
    t3 = timeit(lambda: kp.conv3x3_gen_fwd(dy, wt, Cin))
    t4 = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, wb, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                            [True, False, False]), 10)
    rot = kp.conv3x3_weight_rot180_t(wb)
    t5 = timeit(lambda: F.conv2d(dy, rot, None, 1, 1))
    d = (kp.conv3x3_gen_fwd(x, wf, Cout).float() - F.conv2d(x, wb, None, 1, 1).float()).abs().max().item()
    print("%-12s %3d->%3d @%3d  fwd ours %6.1f us (%.2f PF) +stats %6.1f  vendor %6.1f (%.2f PF) | dgrad ours %6.1f  "
          "vendor bwd %6.1f  vendor fwd(rot) %6.1f | prep %4.1f us  max|diff| %.3g" %
          (name, Cin, Cout, H, t0, fl / t0 / 1e9, t1, t2, fl / t2 / 1e9, t3, t4, t5, t_prep, d), flush=True)
    tot["ours_f"] += n * t1; tot["vend_f"] += n * t2; tot["ours_d"] += n * t3; tot["vend_d"] += n * min(t4, t5)
print("per step: forward ours(+stats) %.0f us vs vendor %.0f us; data gradient ours %.0f us vs vendor %.0f us" %
      (tot["ours_f"], tot["vend_f"], tot["ours_d"], tot["vend_d"]))
