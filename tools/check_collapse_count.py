import torch, sys
sys.path.insert(0, "/root/repo")
from torchseg_amd import kernels as K
kp = K.provider(); dev = torch.device("cuda:0")
x = torch.randn(4, 24, 8, 8, device=dev)
partial, S = kp.bn_stats(x, K.L.NCHW, 4, 24, 64)
a = torch.empty(50, device=dev); b = torch.full((50,), -1.0, device=dev)
kp.bn_collapse(partial, S, 24, a); kp.bn_collapse(partial, S, 24, b, count=4 * 64 + 4096 * 3 + 5)
torch.cuda.synchronize()
assert torch.equal(a[:48], b[:48]) and b[48].item() == 3.0 and b[49].item() == 261.0, (b[48:].tolist())
assert torch.allclose(a[:24], x.sum((0, 2, 3)), atol=1e-3)
print("collapse_count ok")
