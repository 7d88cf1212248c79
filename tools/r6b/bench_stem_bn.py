"""tsg_stem_conv_wrw_bn / _wrw at 16 x 3 x 1024^2 (HIP events on the launch stream), and the unfused pair it replaces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torchseg_amd import kernels as K
dev = torch.device("cuda:0")
kp = K.provider()
B, S = 16, 1024
g = torch.Generator().manual_seed(1)
img = torch.randn(B, 3, S, S, generator=g).to(dev).bfloat16()
w = (torch.randn(64, 3, 7, 7, generator=g) * 0.1).to(dev)
xc = kp.stem_conv_fwd(img, w)
da = torch.randn(xc.shape, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
bp = torch.stack([torch.randn(64, generator=g) * 0.5 + 1.0, torch.randn(64, generator=g) * 0.3,
                  torch.randn(64, generator=g) * 0.1, torch.randn(64, generator=g) * 0.05,
                  torch.randn(64, generator=g) * 0.05]).to(dev).contiguous()
layout, n, c, hw = K.bn_layout(xc)


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


dy, _ = kp.bn_bwd_apply(da, xc, None, layout, n, c, hw, bp, True, False)
ref = kp.stem_conv_wrw(img, dy)
got = kp.stem_conv_wrw_bn(img, da, xc, bp)
print("grid env", os.environ.get("TSG_STEM_BN_GRID"), "max rel diff vs unfused", ((got - ref).abs().max() / ref.abs().max()).item(),
      "equal", torch.equal(got, ref))
t_bn = timeit(lambda: kp.stem_conv_wrw_bn(img, da, xc, bp))
t_plain = timeit(lambda: kp.stem_conv_wrw(img, dy))
t_apply = timeit(lambda: kp.bn_bwd_apply(da, xc, None, layout, n, c, hw, bp, True, False))
nb = img.numel() * 2 + 2 * da.numel() * 2
print("wrw_bn %.1f us (%.0f GB/s)   plain wrw %.1f us   bn_bwd_apply %.1f us" % (t_bn, nb / t_bn / 1e3, t_plain, t_apply))
