import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from torchseg_amd import kernels as K
import torch.nn.functional as F
kp = K.provider()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (B, Ci, Co, H, k, s, p) in [(4, 64, 128, 16, 1, 1, 0), (4, 64, 128, 32, 1, 1, 0), (4, 64, 128, 64, 1, 1, 0), (4, 64, 64, 64, 3, 2, 1), (2, 128, 128, 64, 3, 1, 1), (16, 64, 128, 32, 1, 1, 0)]:
    x = torch.randn(B, Ci, H, H, generator=g); w = torch.randn(Co, Ci, k, k, generator=g) * 0.1
    OH = (H + 2 * p - k) // s + 1
    dy = torch.randn(B, Co, OH, OH, generator=g)
    xw = x.double().requires_grad_(True); ww = w.double().requires_grad_(True)
    y = F.conv2d(xw, ww, None, s, p); y.backward(dy.double())
    for fmt in (torch.contiguous_format, torch.channels_last):
        xd = x.to(dev).contiguous(memory_format=fmt); dyd = dy.to(dev).contiguous(memory_format=fmt)
        wd = w.to(dev).contiguous(memory_format=fmt)
        dw = kp.conv2d_f32_exact_wgrad(xd, dyd, wd, (s, s), (p, p), (1, 1))
        dx = kp.conv2d_f32_exact_dgrad(dyd, wd, xd, (s, s), (p, p), (1, 1))
        yy = kp.conv2d_f32_exact_fwd(xd, wd, (s, s), (p, p), (1, 1))
        e = lambda a, b: float((a.cpu().double() - b).abs().max() / b.abs().max())
        print((B, Ci, Co, H, k, s, p), "cl" if fmt == torch.channels_last else "nchw", "dw %.2e dx %.2e y %.2e" % (e(dw, ww.grad), e(dx, xw.grad), e(yy, y.detach())))
