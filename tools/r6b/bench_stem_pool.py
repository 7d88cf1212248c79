"""The recomputing ResNet stem against the materialising path at 16 x 3 x 1024^2 (HIP events on the launch stream)."""
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
y, partial = kp.stem_conv_fwd_stats(img, w)
gamma = torch.ones(64, device=dev); beta = torch.zeros(64, device=dev)
OH = y.shape[2]
_, invstd, fp = kp.bn_finalize(partial, partial.shape[0], 64, float(B * OH * OH), None, 1e-5, 0.1, gamma, beta, None, None, None)
ypool, idx = kp.bn_relu_pool_fwd(y, fp)
dpool = torch.randn(ypool.shape, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
p2, S2 = kp.bn_relu_pool_bwd_reduce(dpool, idx, y, fp)
_, _, bp = kp.bn_bwd_coeffs(p2, S2, 64, float(B * OH * OH), None, True, invstd, fp, True, True)
dy = kp.bn_relu_pool_bwd_apply(dpool, idx, y, bp)


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


old = [("stem_conv_fwd_stats", lambda: kp.stem_conv_fwd_stats(img, w)), ("bn_relu_pool_fwd", lambda: kp.bn_relu_pool_fwd(y, fp)),
       ("bn_relu_pool_bwd_reduce", lambda: kp.bn_relu_pool_bwd_reduce(dpool, idx, y, fp)),
       ("bn_relu_pool_bwd_apply", lambda: kp.bn_relu_pool_bwd_apply(dpool, idx, y, bp)), ("stem_conv_wrw", lambda: kp.stem_conv_wrw(img, dy))]
new = [("stem_conv_stats", lambda: kp.stem_conv_stats(img, w)), ("stem_conv_bn_relu_pool_fwd", lambda: kp.stem_conv_bn_relu_pool_fwd(img, w, fp)),
       ("stem_conv_bn_relu_pool_bwd_reduce", lambda: kp.stem_conv_bn_relu_pool_bwd_reduce(img, w, dpool, idx, fp)),
       ("stem_conv_wrw_bn_pool", lambda: kp.stem_conv_wrw_bn_pool(img, w, dpool, idx, bp))]
print("  %-36s %7.1f us" % ("stem_conv_wrw_bn_pool(xc=y) waves=" + str(os.environ.get("TSG_STEM_POOL_WAVES")), timeit(lambda: kp.stem_conv_wrw_bn_pool(img, w, dpool, idx, bp, xc=y))))
for name, lst in (("materialising", old), ("recomputing", new)):
    tot = 0.0
    for n, f in lst:
        t = timeit(f); tot += t
        print("  %-36s %7.1f us" % (n, t))
    print("%s stem chain: %.1f us" % (name, tot))
