"""GPU parity of the PSA attention MFMA kernels vs the torch-CPU oracle
(= the reference's own softmax + bmm).  fp32: 1e-4 relative to the output scale
(north_star); bf16: bf16 resolution."""
import numpy as np
import pytest
import torch

from oracle import psa_ref

pytestmark = pytest.mark.gpu

SHAPES = [(2, 64, 72, 72), (1, 128, 200, 136), (2, 512, 256, 320), (1, 24, 8, 8)]


def _inputs(B, Cx, K, N, seed):
    g = torch.Generator().manual_seed(seed)
    X = torch.relu(torch.randn(B, Cx, K, generator=g))       # post-ReLU features
    # asymmetric, structured A so that a row/col swap cannot pass
    A = torch.randn(B, K, N, generator=g) * 2 + torch.arange(K).view(1, K, 1) * 0.01 - torch.arange(N).view(1, 1, N) * 0.02
    dout = torch.randn(B, Cx, N, generator=g)
    return X, A, dout


@pytest.mark.parametrize("shape", SHAPES)
def test_psa_fp32(cuda, shape):
    from torchseg_amd.psa import psa_attention
    X, A, dout = _inputs(*shape, seed=1)
    out_ref, dX_ref, dA_ref = psa_ref.psa_attention_with_grads(X, A, dout)
    Xd, Ad = X.to(cuda).requires_grad_(True), A.to(cuda).requires_grad_(True)
    out = psa_attention(Xd, Ad)
    out.backward(dout.to(cuda))
    for got, ref, name in ((out, out_ref, "out"), (Xd.grad, dX_ref, "dX"), (Ad.grad, dA_ref, "dA")):
        scale = ref.abs().max().item()
        err = (got.detach().cpu().double() - ref).abs().max().item()
        assert err <= 1e-4 * scale, (name, err, scale)


@pytest.mark.parametrize("shape", SHAPES)
def test_psa_bf16(cuda, shape):
    from torchseg_amd.psa import psa_attention
    X, A, dout = _inputs(*shape, seed=2)
    X, A, dout = X.bfloat16(), A.bfloat16(), dout.bfloat16()
    out_ref, dX_ref, dA_ref = psa_ref.psa_attention_with_grads(X.float(), A.float(), dout.float())
    Xd, Ad = X.to(cuda).requires_grad_(True), A.to(cuda).requires_grad_(True)
    out = psa_attention(Xd, Ad)
    assert out.dtype == torch.bfloat16
    out.backward(dout.to(cuda))
    for got, ref, name in ((out, out_ref, "out"), (Xd.grad, dX_ref, "dX"), (Ad.grad, dA_ref, "dA")):
        scale = ref.abs().max().item()
        err = (got.detach().float().cpu().double() - ref).abs().max().item()
        assert err <= 2e-2 * scale, (name, err, scale)


def test_psa_full_size_matches_gpu_fp32_reference(cuda):
    """PSANet's real size (Cx 512, 3600 x 3600), one sample: against torch's fp32
    softmax+bmm on the same device (CPU fp64 at this size is too slow for the suite)
    and a linearity property: psa(X1 + X2, A) == psa(X1, A) + psa(X2, A)."""
    from torchseg_amd.psa import psa_attention
    g = torch.Generator(device=cuda).manual_seed(3)
    X = torch.relu(torch.randn(1, 512, 3600, generator=g, device=cuda))
    X2 = torch.randn(1, 512, 3600, generator=g, device=cuda)
    A = torch.randn(1, 3600, 3600, generator=g, device=cuda)
    ref = torch.bmm(X, torch.softmax(A, dim=1))
    out = psa_attention(X, A)
    assert (out - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
    lin = psa_attention(X + X2, A) - psa_attention(X2, A)
    assert (lin - out).abs().max().item() <= 2e-4 * ref.abs().max().item()
    outb = psa_attention(X.bfloat16(), A.bfloat16()).float()
    refb = torch.bmm(X.bfloat16().float(), torch.softmax(A.bfloat16().float(), dim=1))
    assert (outb - refb).abs().max().item() <= 2e-2 * refb.abs().max().item()


def test_fuse_mode_intercepts_unchanged_call_pattern(cuda):
    """The reference writes torch.bmm(x, torch.softmax(a, dim=1)); under FusePsaMode
    that exact expression must run the fused kernel and stay differentiable."""
    from torchseg_amd import kernels as K
    from torchseg_amd.psa import FusePsaMode
    kp = K.provider()
    n = {"fwd": 0}
    orig = kp.psa_fwd
    kp.psa_fwd = lambda *a: (n.__setitem__("fwd", n["fwd"] + 1), orig(*a))[1]
    try:
        x = torch.randn(2, 64, 80, device=cuda, requires_grad=True)
        a = torch.randn(2, 80, 80, device=cuda, requires_grad=True)
        with FusePsaMode():
            y = torch.bmm(x, torch.softmax(a, dim=1))
            z = torch.softmax(a, dim=1).sum()          # non-bmm consumer: materialises
        (y.sum() + z).backward()
    finally:
        kp.psa_fwd = orig
    assert n["fwd"] == 1
    ref = torch.bmm(x.detach(), torch.softmax(a.detach(), dim=1))
    torch.testing.assert_close(y.detach(), ref, rtol=1e-4, atol=1e-5)
    assert x.grad is not None and a.grad is not None


def test_psa_bf16_full_size_fwd_bwd_vs_cpu_oracle(cuda):
    """PSANet's real size (Cx 512, 3600 x 3600: 14 full + 1 partial M tile, 56 + 1 N tiles, 56 + 1 K tiles of the fused
    kernels) in bf16, forward AND both gradients against the CPU oracle (the reference's own softmax + bmm,
    psanet network.py:125-126, in fp32 on the same bf16-rounded inputs)."""
    from torchseg_amd.psa import psa_attention
    g = torch.Generator().manual_seed(11)
    X = torch.relu(torch.randn(1, 512, 3600, generator=g)).bfloat16()
    A = (torch.randn(1, 3600, 3600, generator=g) * 2 + torch.arange(3600).view(1, 3600, 1) * 0.001
         - torch.arange(3600).view(1, 1, 3600) * 0.002).bfloat16()
    dout = torch.randn(1, 512, 3600, generator=g).bfloat16()
    Xr, Ar = X.float().requires_grad_(True), A.float().requires_grad_(True)
    ref = torch.bmm(Xr, torch.softmax(Ar, dim=1))
    ref.backward(dout.float())
    Xd, Ad = X.to(cuda).requires_grad_(True), A.to(cuda).requires_grad_(True)
    out = psa_attention(Xd, Ad)
    out.backward(dout.to(cuda))
    for got, want, name in ((out, ref.detach(), "out"), (Xd.grad, Xr.grad, "dX"), (Ad.grad, Ar.grad, "dA")):
        scale = want.abs().max().item()
        d = (got.detach().float().cpu() - want)
        assert d.abs().max().item() <= 2e-2 * scale, (name, d.abs().max().item(), scale)
        assert (d.pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item() <= 1e-2, name
        # the last rows / columns (partial tiles) carry real values
        assert got[..., -16:].float().abs().sum().item() > 0 and got[..., -16:, :].float().abs().sum().item() > 0


@pytest.mark.parametrize("regime", ["one_hot_column", "all_large", "all_very_negative"])
def test_psa_bf16_logits_outside_the_optimistic_range_take_the_guarded_path(cuda, regime):
    """Round 4: the bf16 forward contracts X with exp(A) and normalises by the column sums it collected on the way (no
    column-statistics pass).  That is only safe while a column's sum of exp stays in [1e-20, 1e20]; any tile that sees
    more raises a device flag and the classic three launches (column statistics, exp(a - lse)) redo the call.  Logits far
    outside the range — one column, every column, everything hugely negative — must give the oracle's result (the
    reference's torch.softmax subtracts the column maximum: psanet network.py:125-126), forward, lse and gradients."""
    from torchseg_amd import kernels as K
    from torchseg_amd.psa import psa_attention
    g = torch.Generator().manual_seed(5)
    B, Cx, Kd, N = 2, 512, 264, 200
    X = torch.relu(torch.randn(B, Cx, Kd, generator=g)).bfloat16()
    A = torch.randn(B, Kd, N, generator=g) * 2
    if regime == "one_hot_column":
        A[1, 7, 131] = 120.0                                # exp(120) overflows fp32: one column of one sample
    elif regime == "all_large":
        A = A * 40.0
    else:
        A = A - 200.0                                       # every exp underflows: the sums are 0
    A = A.bfloat16()
    dout = torch.randn(B, Cx, N, generator=g).bfloat16()
    out_ref, dX_ref, dA_ref = psa_ref.psa_attention_with_grads(X.float(), A.float(), dout.float())
    lse_ref = torch.logsumexp(A.double(), dim=1)
    Xd, Ad = X.to(cuda).requires_grad_(True), A.to(cuda).requires_grad_(True)
    out = psa_attention(Xd, Ad)
    out.backward(dout.to(cuda))
    assert torch.isfinite(out.float()).all()
    for got, ref, name in ((out, out_ref, "out"), (Xd.grad, dX_ref, "dX"), (Ad.grad, dA_ref, "dA")):
        scale = ref.abs().max().item()
        err = (got.detach().float().cpu().double() - ref).abs().max().item()
        assert err <= 2e-2 * scale, (regime, name, err, scale)
    _, lse = K.provider().psa_fwd(X.to(cuda), A.to(cuda))
    assert (lse.double().cpu() - lse_ref).abs().max().item() <= 1e-5 * lse_ref.abs().max().item() + 1e-5


def test_psa_bf16_lse_of_the_optimistic_forward(cuda):
    """In-range logits (the ordinary case): lse = log of the column sums the contraction collected equals
    torch.logsumexp to fp32 accuracy, and two calls are bit-identical."""
    from torchseg_amd import kernels as K
    g = torch.Generator().manual_seed(6)
    X = torch.relu(torch.randn(2, 512, 520, generator=g)).bfloat16().to(cuda)
    A = (torch.randn(2, 520, 456, generator=g) * 3).bfloat16().to(cuda)
    kp = K.provider()
    out, lse = kp.psa_fwd(X, A)
    out2, lse2 = kp.psa_fwd(X, A)
    assert torch.equal(out, out2) and torch.equal(lse, lse2)
    ref = torch.logsumexp(A.double().cpu(), dim=1)
    assert (lse.double().cpu() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item() + 2e-6
