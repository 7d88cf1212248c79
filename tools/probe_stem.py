"""Stem conv (3->64, 7x7, stride 2, pad 3, [16,3,1024,1024] bf16, weight gradient only): MIOpen
directly vs the same convolution restated as space-to-depth(2) + 4x4 stride-1 conv on 12 (16)
channels in NHWC.  Prints ms per fwd+bwd and the max abs difference of output / weight grad."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from torchseg_amd.tuning import use_shipped_miopen_db
use_shipped_miopen_db(0)
dev = torch.device("cuda:0")
B, S = 16, 1024
torch.manual_seed(0)
x = torch.randn(B, 3, S, S, device=dev).bfloat16()
w = (torch.randn(64, 3, 7, 7, device=dev) * 0.05).requires_grad_()
dy = torch.randn(B, 64, S // 2, S // 2, device=dev).bfloat16()
dy_cl = dy.contiguous(memory_format=torch.channels_last)


def s2d_input(x, cpad):
    b, c, h, w_ = x.shape
    xp = F.pad(x, (4, 2, 4, 2))                                    # 2 s2d cells left/top, 1 right/bottom
    hp, wp = (h + 6) // 2, (w_ + 6) // 2
    t = xp.view(b, c, hp, 2, wp, 2).permute(0, 2, 4, 3, 5, 1).reshape(b, hp, wp, 4 * c)
    if cpad > 4 * c:
        t = F.pad(t, (0, cpad - 4 * c))
    return t.permute(0, 3, 1, 2)                                    # NCHW view of NHWC memory


def s2d_weight(w, cpad):
    o, c, _, _ = w.shape
    w8 = F.pad(w, (1, 0, 1, 0))                                     # kh' = kh + 1
    t = w8.view(o, c, 4, 2, 4, 2).permute(0, 3, 5, 1, 2, 4).reshape(o, 4 * c, 4, 4)
    if cpad > 4 * c:
        t = F.pad(t, (0, 0, 0, 0, 0, cpad - 4 * c))
    return t.contiguous(memory_format=torch.channels_last)


def run_direct():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = F.conv2d(x, w, None, 2, 3)
    y.backward(dy)
    return y


def make_s2d(cpad, cache_x):
    xs = s2d_input(x, cpad) if cache_x else None
    def run():
        xi = xs if cache_x else s2d_input(x, cpad)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = F.conv2d(xi, s2d_weight(w, cpad), None, 1, 0)
        y.backward(dy_cl)
        return y
    return run


def timeit(fn, n=20):
    for _ in range(5):
        w.grad = None; fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        w.grad = None; fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


w.grad = None; y0 = run_direct(); g0 = w.grad.clone()
print("direct nchw: %.3f ms" % timeit(run_direct))
for cpad in (12, 16):
    for cache in (True, False):
        fn = make_s2d(cpad, cache)
        w.grad = None; y1 = fn(); g1 = w.grad.clone()
        print("s2d C=%d cache_x=%d: %.3f ms   out max|d| %.3e (max|y| %.2f)  dW rel %.3e   y strides %s"
              % (cpad, cache, timeit(fn), (y1.float() - y0.float()).abs().max().item(), y0.float().abs().max().item(),
                 ((g1 - g0).norm() / g0.norm()).item(), y1.stride()))
