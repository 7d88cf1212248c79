"""SyncBatchNorm oracle (numpy, float64 accumulate).  Test infrastructure only.

Restates furnace/legacy/sync_bn/syncbn.py:32-52 (forward), :86-98
(_compute_mean_std) and src/gpu/syncbn_kernel.cu:12-23,92-138,160-174
(backward pieces) with the autograd composition of functions.py:22-61.
All arrays are [N, C, *spatial]; "ranks" are modelled by passing a list of
per-rank arrays, the cross-GPU ReduceAddCoalesced (syncbn.py:75) being a sum.
"""
import numpy as np


def _red_axes(x):
    return (0,) + tuple(range(2, x.ndim))


def _bc(v, x):
    return v.reshape((1, -1) + (1,) * (x.ndim - 2))


def sum_square(x):
    """Sum_Square_Forward_kernel (syncbn_kernel.cu:142-157): per-channel sum x, sum x^2."""
    x = x.astype(np.float64)
    return x.sum(axis=_red_axes(x)), (x * x).sum(axis=_red_axes(x))


def compute_mean_std(sum_, ssum, size, eps, momentum, running_mean=None, running_var=None):
    """syncbn.py:86-98.  Returns mean, inv_std, new_running_mean, new_running_var."""
    assert size > 1, "BatchNorm computes unbiased standard-deviation, which requires size > 1."
    mean = sum_ / size
    sumvar = ssum - sum_ * mean
    unbias_var = sumvar / (size - 1)
    bias_var = sumvar / size
    rm = rv = None
    if running_mean is not None:
        rm = (1 - momentum) * running_mean.astype(np.float64) + momentum * mean
        rv = (1 - momentum) * running_var.astype(np.float64) + momentum * unbias_var
    return mean, (bias_var + eps) ** -0.5, rm, rv


def forward(xs, gamma, beta, eps=1e-5, momentum=0.1, running_mean=None, running_var=None,
            residuals=None, relu=False):
    """Training forward over a list of per-rank inputs (syncbn.py:32-52 +
    BatchNorm_Forward_kernel syncbn_kernel.cu:73-89), optionally followed by the
    residual add and ReLU of resnet.py:44-51 / seg_oprs.py:39-46.
    Returns (ys, mean, inv_std, running_mean, running_var)."""
    s = sum(sum_square(x)[0] for x in xs)
    q = sum(sum_square(x)[1] for x in xs)
    size = sum(x.size // x.shape[1] for x in xs)
    mean, inv_std, rm, rv = compute_mean_std(s, q, size, eps, momentum, running_mean, running_var)
    ys = []
    for i, x in enumerate(xs):
        x64 = x.astype(np.float64)
        y = _bc(gamma.astype(np.float64), x) * (x64 - _bc(mean, x)) * _bc(inv_std, x) + _bc(beta.astype(np.float64), x)
        if residuals is not None and residuals[i] is not None:
            y = y + residuals[i].astype(np.float64)
        if relu:
            y = np.maximum(y, 0.0)
        ys.append(y)
    return ys, mean, inv_std, rm, rv


def backward(xs, dys, ys, gamma, mean, inv_std, relu=False):
    """Backward over per-rank lists.  dy' = dy*[y>0] when relu (nn.ReLU backward).
    GradOp (syncbn_kernel.cu:12-23): sum dy', sum dy'*(x-mean) -> dgamma = dotP*invstd
    (:130), dbeta = sum dy' (:135), gradMean/gradStd (:118-119) chained through
    _compute_mean_std and Sum_Square_Backward (:170), which collapses to
      dx = gamma*invstd * (dy' - mean_n(dy') - xhat * mean_n(dy'*xhat)).
    Returns (dxs, dres_list, dgamma_per_rank, dbeta_per_rank)."""
    g = gamma.astype(np.float64)
    dps = []
    for x, dy, y in zip(xs, dys, ys):
        d = dy.astype(np.float64)
        if relu:
            d = d * (y > 0)
        dps.append(d)
    size = sum(x.size // x.shape[1] for x in xs)
    xhats = [(x.astype(np.float64) - _bc(mean, x)) * _bc(inv_std, x) for x in xs]
    sdy_r = [d.sum(axis=_red_axes(d)) for d in dps]
    sdx_r = [(d * xh).sum(axis=_red_axes(d)) for d, xh in zip(dps, xhats)]
    sdy, sdx = sum(sdy_r), sum(sdx_r)
    dxs = []
    for x, d, xh in zip(xs, dps, xhats):
        dxs.append(_bc(g * inv_std, x) * (d - _bc(sdy / size, x) - xh * _bc(sdx / size, x)))
    return dxs, dps, sdx_r, sdy_r
