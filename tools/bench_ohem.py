"""GPU microbench of the OHEM kernels at the headline size (16 x 19 x 1024 x 1024 bf16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchseg_amd import kernels as K
kp = K.provider(); dev = torch.device("cuda:0")
B, C, H, W = 16, 19, 1024, 1024
g = torch.Generator(device=dev).manual_seed(0)
logits = torch.randn(B, C, H, W, device=dev, generator=g).bfloat16()
lab = torch.randint(0, C, (B, H, W), device=dev, generator=g); lab[:, :8] = 255
def timeit(fn, n=10):
    for _ in range(2): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
loss, nll, lse, sel = kp.ohem_fwd(logits, lab, 255, 0.7, B * H * W // 16, None)
gs = torch.ones(1, device=dev)
t_f = timeit(lambda: kp.ohem_fwd(logits, lab, 255, 0.7, B * H * W // 16, None))
t_b = timeit(lambda: kp.ohem_bwd(logits, lab, 255, None, nll, lse, sel, gs))
fb, bb = 906e6, 1543.5e6
print(f"TSG_OHEM_BWD_LDS={os.environ.get('TSG_OHEM_BWD_LDS','0'):>6}  fwd {t_f:7.1f} us {fb/t_f/1e3:6.0f} GB/s   bwd {t_b:7.1f} us {bb/t_b/1e3:6.0f} GB/s  kept={int(sel[1])}")
