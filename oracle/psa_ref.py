"""PSA collect/distribute attention oracle (torch CPU).  Test infrastructure only.

Restates model/psanet/ade.psanet.R101_v1c/network.py:125-126 (and :135-136):
    fm = torch.bmm(reduce_x, torch.softmax(attention, dim=1))
with X [B, Cx, K], A [B, K, N]; gradients come from torch autograd on the same
expression.  No golden vectors exist for it in the reference ("parity unpinned"
beyond torch's own softmax/bmm, which ARE what the reference calls)."""
import torch


def psa_attention(X, A):
    return torch.bmm(X, torch.softmax(A, dim=1))


def psa_attention_with_grads(X, A, dout):
    X = X.detach().double().requires_grad_(True)
    A = A.detach().double().requires_grad_(True)
    out = psa_attention(X, A)
    out.backward(dout.double())
    return out.detach(), X.grad, A.grad
