"""GPU microbench of one BiSeNet head's loss path at the bench shape (B=16, C=19, 128x128 -> 1024x1024, bf16 logits):
materialised (upsample kernel + OHEM kernels) against the fused upsample->OHEM kernels, forward and backward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchseg_amd import kernels as K
from torchseg_amd.losses import ohem_cross_entropy
from torchseg_amd.upsample import DeferredUpsample, upsample_bilinear_ac, install_aten_overrides
install_aten_overrides()
dev = torch.device("cuda:0")
B, C, IH, OH = int(os.environ.get("HB", 16)), 19, 128, 1024
g = torch.Generator(device=dev).manual_seed(0)
def timeit(fn, n=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
for ldt in (torch.int64, torch.uint8):
    t = torch.randint(0, C, (B, OH, OH), device=dev, generator=g).to(ldt)
    t[:, :40] = 255
    for dtype in (torch.bfloat16, torch.float32):
        z = torch.randn(B, C, IH, IH, device=dev, generator=g).to(dtype).requires_grad_(True)
        k = B * OH * OH // 16
        res = {}
        for name, mk in (("materialised", lambda: upsample_bilinear_ac(z, size=(OH, OH))), ("fused", lambda: DeferredUpsample(z, (OH, OH)))):
            def fwd():
                return ohem_cross_entropy(mk(), t, 255, 0.7, k)
            def fb():
                z.grad = None
                fwd().backward()
            with torch.no_grad():
                tf = timeit(fwd)
            tfb = timeit(fb)
            fb(); res[name] = (fwd().item(), z.grad.float().clone())
            print(f"{str(dtype).split('.')[-1]:9s} labels {str(ldt).split('.')[-1]:6s} {name:13s} fwd {tf:8.1f} us   fwd+bwd {tfb:8.1f} us", flush=True)
        la, ga = res["materialised"]; lb, gb = res["fused"]
        print(f"    loss {la:.6f} vs {lb:.6f}; grad max |diff| / max |g| = {(ga - gb).abs().max().item() / ga.abs().max().item():.3e}")
