"""CPU: pin the SyncBN oracle against torch.nn.BatchNorm2d applied to the
rank-concatenated batch in fp64 (identical maths incl. unbiased running var,
SURVEY.md §8c)."""
import numpy as np
import pytest
import torch

from oracle import syncbn_ref as R


@pytest.mark.parametrize("ranks", [1, 2, 3])
@pytest.mark.parametrize("relu,res", [(False, False), (True, False), (True, True)])
def test_oracle_matches_torch_batchnorm(ranks, relu, res):
    rng = np.random.default_rng(7 + ranks)
    C = 6
    xs = [rng.standard_normal((2 + r, C, 5, 4)) * 1.7 + 0.3 for r in range(ranks)]
    rs = [rng.standard_normal(x.shape) for x in xs] if res else None
    gamma = rng.standard_normal(C) + 1.0
    beta = rng.standard_normal(C)
    rm0, rv0 = rng.standard_normal(C), rng.random(C) + 0.5
    ys, mean, inv_std, rm, rv = R.forward(xs, gamma, beta, 1e-5, 0.1, rm0, rv0, rs, relu)

    bn = torch.nn.BatchNorm2d(C, eps=1e-5, momentum=0.1).double()
    with torch.no_grad():
        bn.weight.copy_(torch.from_numpy(gamma)); bn.bias.copy_(torch.from_numpy(beta))
        bn.running_mean.copy_(torch.from_numpy(rm0)); bn.running_var.copy_(torch.from_numpy(rv0))
    xt = torch.from_numpy(np.concatenate(xs, 0)).requires_grad_(True)
    yt = bn(xt)
    if res:
        rt = torch.from_numpy(np.concatenate(rs, 0)).requires_grad_(True)
        yt = yt + rt
    if relu:
        yt = torch.relu(yt)
    np.testing.assert_allclose(np.concatenate(ys, 0), yt.detach().numpy(), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(rm, bn.running_mean.numpy(), rtol=1e-12)
    np.testing.assert_allclose(rv, bn.running_var.numpy(), rtol=1e-12)

    dys = [rng.standard_normal(x.shape) for x in xs]
    yt.backward(torch.from_numpy(np.concatenate(dys, 0)))
    dxs, dres, dg_r, db_r = R.backward(xs, dys, ys, gamma, mean, inv_std, relu)
    np.testing.assert_allclose(np.concatenate(dxs, 0), xt.grad.numpy(), rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(sum(dg_r), bn.weight.grad.numpy(), rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(sum(db_r), bn.bias.grad.numpy(), rtol=1e-9, atol=1e-10)
    if res:
        np.testing.assert_allclose(np.concatenate(dres, 0), rt.grad.numpy(), rtol=1e-10, atol=1e-12)


def test_oracle_matches_reference_compute_mean_std_golden():
    """Pin to the REFERENCE itself: tests/golden/syncbn_golden.npz holds the outputs of _SyncBatchNorm._compute_mean_std
    (furnace/legacy/sync_bn/syncbn.py:86-98) executed verbatim by tests/golden/make_golden.py on cross-"GPU" sums of 1, 2,
    3 and 8 ranks (incl. a [B,C,1,1] activation and a non-default eps / momentum): mean, inv_std and both running
    statistics (unbiased variance)."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "syncbn_golden.npz"))
    names = sorted({k.split("/")[0] for k in z.files})
    assert len(names) == 4
    for name in names:
        C, ranks, size, eps, momentum = z[name + "/cfg"]
        xs = [z[name + "/x%d" % r] for r in range(int(ranks))]
        # the oracle's own sums reproduce what the reference was fed ...
        s = sum(R.sum_square(x)[0] for x in xs)
        q = sum(R.sum_square(x)[1] for x in xs)
        np.testing.assert_allclose(s, z[name + "/sum"], rtol=2e-6, atol=1e-3)
        np.testing.assert_allclose(q, z[name + "/ssum"], rtol=2e-6, atol=1e-3)
        assert sum(x.size // x.shape[1] for x in xs) == int(size)
        # ... and from the SAME fp32 sums it lands on the reference's fp32 results to fp32 rounding
        mean, inv_std, rm, rv = R.compute_mean_std(z[name + "/sum"].astype(np.float64), z[name + "/ssum"].astype(np.float64),
                                                   int(size), eps, momentum, z[name + "/rm0"], z[name + "/rv0"])
        np.testing.assert_allclose(mean, z[name + "/mean"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(inv_std, z[name + "/inv_std"], rtol=2e-5)       # ssum - sum*mean cancels in fp32
        np.testing.assert_allclose(rm, z[name + "/rm1"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(rv, z[name + "/rv1"], rtol=2e-5, atol=1e-6)
        # and the full forward of the oracle uses exactly these statistics
        ys, mean2, inv2, _, _ = R.forward(xs, np.ones(int(C)), np.zeros(int(C)), eps, momentum)
        np.testing.assert_allclose(mean2, z[name + "/mean"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(inv2, z[name + "/inv_std"], rtol=1e-4)
