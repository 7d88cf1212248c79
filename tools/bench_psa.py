"""GPU microbench of the PSA attention kernels at PSANet's size (B=2, Cx=512, L=3600): forward / backward time,
TFLOP/s against the 2.5 PF dense bf16 MFMA peak, and the HBM rate of the unavoidable traffic (A read twice forward)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchseg_amd import kernels as K
kp = K.provider(); dev = torch.device("cuda:0")
B, Cx, Lk = 2, 512, 3600
def timeit(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
g = torch.Generator(device=dev).manual_seed(2)
QUICK = os.environ.get("PSA_QUICK", "0") == "1"      # bf16 only, no torch reference timings
for dtype in ((torch.bfloat16,) if QUICK else (torch.bfloat16, torch.float32)):
    X = torch.relu(torch.randn(B, Cx, Lk, device=dev, generator=g)).to(dtype)
    A = torch.randn(B, Lk, Lk, device=dev, generator=g).to(dtype)
    dout = torch.randn(B, Cx, Lk, device=dev, generator=g).to(dtype)
    out, lse = kp.psa_fwd(X, A)
    t_f = timeit(lambda: kp.psa_fwd(X, A))
    t_b = timeit(lambda: kp.psa_bwd(X, A, out, dout, lse))
    flop_f = 2.0 * B * Cx * Lk * Lk
    t_ref_f = 0.0 if QUICK else timeit(lambda: torch.bmm(X, torch.softmax(A, dim=1)), n=5)
    Xr, Ar = X.clone().requires_grad_(True), A.clone().requires_grad_(True)
    def ref_fb():
        Xr.grad = None; Ar.grad = None
        torch.bmm(Xr, torch.softmax(Ar, dim=1)).backward(dout)
    t_ref_fb = 0.0 if QUICK else timeit(ref_fb, n=5)
    mult = 3 if dtype == torch.float32 else 1     # split-precision passes
    print(f"{str(dtype).split('.')[-1]:9s} fwd {t_f:8.1f} us  ({flop_f*mult/t_f/1e6:7.1f} TFLOP/s issued, {flop_f/t_f/1e6:6.1f} useful, "
          f"{flop_f*mult/t_f/1e6/2500:.3f} of the 2.5 PF bf16 peak)   bwd {t_b:8.1f} us ({2*flop_f/t_b/1e6:6.1f} useful TFLOP/s, "
          f"{2*flop_f*mult/t_b/1e6/2500:.3f} of peak)   torch fwd {t_ref_f:8.1f} us  torch fwd+bwd {t_ref_fb:8.1f} us  ours fwd+bwd {t_f+t_b:8.1f} us")
