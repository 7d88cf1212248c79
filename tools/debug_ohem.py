import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_ohem_gpu import _make
from torchseg_amd import kernels as K
from torchseg_amd.losses import ohem_cross_entropy
cuda = torch.device("cuda:0")
B, C, H, W = 2, 19, 128, 128
pred, t = _make(B, C, H, W, "confident", seed=5)
k = B * H * W // 2
kp = K.provider()
for trial in range(3):
    loss, nll, lse, sel = kp.ohem_fwd(pred.to(cuda), t.to(cuda), 255, 0.7, k, None)
    torch.cuda.synchronize()
    print("direct", trial, sel.cpu().tolist(), loss.item())
p = pred.to(cuda); tt = t.to(cuda)
loss, nll, lse, sel = kp.ohem_fwd(p, tt, 255, 0.7, k, None)
print("direct kept refs", sel.cpu().tolist(), loss.item())
l2, s2 = ohem_cross_entropy(p, tt, 255, 0.7, k, None, return_selection=True)
print("via fn", s2.cpu().tolist(), l2.item())
pr = torch.exp(-nll)
print("count p<=0.7 all:", int((torch.where(tt.view(-1) != 255, pr, torch.ones_like(pr)) <= 0.7).sum()), "k", k)
