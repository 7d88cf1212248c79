"""GPU: the N > 1 SyncBN kernel path on one device.  Two "ranks" with unequal batches run the per-rank kernels
(stats -> collapse_count, bwd_reduce -> collapse), the two exchange messages are summed by hand (what the RCCL
all-reduce does), and finalize / bwd_coeffs consume the summed message with the device-side count.  Result ==
torch BatchNorm2d (fp32) on the concatenated batch: y, running stats, dx, dgamma, dbeta."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_two_rank_exchange_matches_batchnorm_on_concatenated_batch(cuda, layout):
    from torchseg_amd import kernels as K
    kp = K.provider()
    torch.manual_seed(3)
    C, H, W = 24, 10, 12
    xs = [torch.randn(3, C, H, W, device=cuda) * 2 + 0.5, torch.randn(5, C, H, W, device=cuda) - 1.0]   # unequal batches
    dys = [torch.randn_like(t) for t in xs]
    if layout == "nhwc":
        xs = [t.contiguous(memory_format=torch.channels_last) for t in xs]
        dys = [t.contiguous(memory_format=torch.channels_last) for t in dys]
    lay = K.L.NHWC if layout == "nhwc" else K.L.NCHW
    gamma = torch.rand(C, device=cuda) + 0.5
    beta = torch.randn(C, device=cuda)
    eps, mom = 1e-5, 0.1
    HW = H * W

    # forward exchange
    msgs = []
    for t in xs:
        partial, S = kp.bn_stats(t, lay, t.shape[0], C, HW)
        m = torch.empty(2 * C + 2, device=cuda)
        kp.bn_collapse(partial, S, C, m, count=t.shape[0] * HW)
        msgs.append(m)
    msg = msgs[0] + msgs[1]                                   # the all-reduce(SUM)
    n_total = sum(t.shape[0] for t in xs) * HW
    assert msg[2 * C].item() * 4096 + msg[2 * C + 1].item() == n_total
    rm, rv = torch.zeros(C, device=cuda), torch.ones(C, device=cuda)
    nbt = torch.zeros((), dtype=torch.int64, device=cuda)
    mean, invstd, fp = kp.bn_finalize(msg, 1, C, 0.0, msg[2 * C:], eps, mom, gamma, beta, rm, rv, nbt)
    ys = [kp.bn_apply_fwd(t, None, lay, t.shape[0], C, HW, fp, False) for t in xs]

    # backward exchange
    sums, locals_ = [], []
    for t, d in zip(xs, dys):
        partial, S = kp.bn_bwd_reduce(d, t, None, lay, t.shape[0], C, HW, fp, False)
        sm = torch.empty(2 * C, device=cuda)
        kp.bn_collapse(partial, S, C, sm)
        dg, db, _ = kp.bn_bwd_coeffs(sm, 1, C, 1.0, None, True, invstd, fp, True, False)     # LOCAL parameter grads
        sums.append(sm); locals_.append((dg, db))
    tot = sums[0] + sums[1]                                   # the all-reduce(SUM)
    _, _, bp = kp.bn_bwd_coeffs(tot, 1, C, 0.0, msg[2 * C:], True, invstd, fp, False, True)
    dxs = [kp.bn_bwd_apply(d, t, None, lay, t.shape[0], C, HW, bp, False, False)[0] for t, d in zip(xs, dys)]

    # reference: one BatchNorm2d over the concatenated batch
    ref = nn.BatchNorm2d(C, eps=eps, momentum=mom).to(cuda)
    with torch.no_grad():
        ref.weight.copy_(gamma); ref.bias.copy_(beta)
    xall = torch.cat([t.contiguous() for t in xs]).requires_grad_()
    yall = ref(xall)
    yall.backward(torch.cat([d.contiguous() for d in dys]))
    torch.testing.assert_close(torch.cat([t.contiguous() for t in ys]), yall.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(rm, ref.running_mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rv, ref.running_var, rtol=1e-5, atol=1e-6)
    assert int(nbt) == 1
    torch.testing.assert_close(torch.cat([t.contiguous() for t in dxs]), xall.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(locals_[0][0] + locals_[1][0], ref.weight.grad, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(locals_[0][1] + locals_[1][1], ref.bias.grad, rtol=1e-4, atol=1e-3)
