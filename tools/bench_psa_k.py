"""PSA forward / backward at a varying contraction length (K rows of A) with N = 3600 columns: time = a + b * K tiles
separates psa_mm's fixed cost (launch, prologue, epilogue) from its K-loop cost.  Run under rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchseg_amd import kernels as K
kp = K.provider(); dev = torch.device("cuda:0")
B, Cx, N = 2, 512, 3600
Kr = int(os.environ.get("PSA_K", "3600"))
g = torch.Generator(device=dev).manual_seed(2)
X = torch.relu(torch.randn(B, Cx, Kr, device=dev, generator=g)).to(torch.bfloat16)
A = torch.randn(B, Kr, N, device=dev, generator=g).to(torch.bfloat16)
dout = torch.randn(B, Cx, N, device=dev, generator=g).to(torch.bfloat16)
out, lse = kp.psa_fwd(X, A)
for _ in range(20):
    out, lse = kp.psa_fwd(X, A)
    kp.psa_bwd(X, A, out, dout, lse)
torch.cuda.synchronize()
