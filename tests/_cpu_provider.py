"""TEST-ONLY stand-in kernel provider: implements the SyncBN provider methods with
the numpy oracle so that the HOST logic of torchseg_amd.syncbn (message layout,
collective sequencing, count handling, autograd wiring) can run under gloo on
CPU with world_size > 1.  Lives in tests/; the package never selects it."""
import numpy as np
import torch


def _to_np(t):
    return t.detach().double().numpy()


class OracleProvider:
    name = "oracle-test-standin"

    def bn_stats(self, x, layout, N, C, HW):
        x64 = x.detach().double().reshape(N, C, -1) if x.dim() > 2 else x.detach().double().reshape(N, C, 1)
        part = torch.stack([x64.sum((0, 2)), (x64 * x64).sum((0, 2))]).float().unsqueeze(0)
        return part.contiguous(), 1

    def bn_collapse(self, partial, S, C, out):
        out[:2 * C].copy_(partial[:S].double().sum(0).reshape(-1).float())

    def _sums(self, partial, S, C):
        p = partial.reshape(-1)[:S * 2 * C].reshape(S, 2, C).double().sum(0)
        return p[0], p[1]

    def bn_finalize(self, partial, S, C, count, count_dev, eps, momentum, rm, rv, nbt):
        s, q = self._sums(partial, S, C)
        n = float(count_dev[0]) * 4096.0 + float(count_dev[1]) if count_dev is not None else float(count)
        mean = s / n
        sumvar = (q - s * mean).clamp_min(0)
        invstd = (sumvar / n + eps) ** -0.5
        if rm is not None:
            rm.copy_(((1 - momentum) * rm.double() + momentum * mean).float())
        if rv is not None:
            rv.copy_(((1 - momentum) * rv.double() + momentum * sumvar / (n - 1)).float())
        if nbt is not None:
            nbt.add_(1)
        return mean.float(), invstd.float()

    @staticmethod
    def _bc(v, x):
        return v.reshape((1, -1) + (1,) * (x.dim() - 2))

    def bn_apply_fwd(self, x, residual, layout, N, C, HW, mean, invstd, gamma, beta, relu, out=None):
        g = gamma if gamma is not None else torch.ones_like(mean)
        b = beta if beta is not None else torch.zeros_like(mean)
        y = (x.float() - self._bc(mean, x)) * self._bc(invstd * g, x) + self._bc(b, x)
        if residual is not None:
            y = y + residual.float()
        if relu:
            y = y.clamp_min(0)
        return y.to(x.dtype)

    def _mask(self, dy, x, y, mean, invstd, gamma, beta, relu):
        d = dy.float()
        if relu:
            ref = y if y is not None else self.bn_apply_fwd(x, None, 0, 0, 0, 0, mean, invstd, gamma, beta, False)
            d = d * (ref > 0)
        return d

    def bn_bwd_reduce(self, dy, x, y, layout, N, C, HW, mean, invstd, gamma, beta, relu):
        d = self._mask(dy, x, y, mean, invstd, gamma, beta, relu).double()
        xh = (x.double() - self._bc(mean.double(), x)) * self._bc(invstd.double(), x)
        ax = (0,) + tuple(range(2, x.dim()))
        part = torch.stack([d.sum(ax), (d * xh).sum(ax)]).float().unsqueeze(0)
        return part.contiguous(), 1

    def bn_bwd_coeffs(self, partial, S, C, count, count_dev, want_param_grads, want_k):
        s, q = self._sums(partial, S, C)
        dg = q.float() if want_param_grads else None
        db = s.float() if want_param_grads else None
        k = None
        if want_k:
            n = float(count_dev[0]) * 4096.0 + float(count_dev[1]) if count_dev is not None else float(count)
            k = torch.stack([s / n, q / n]).float()
        return dg, db, k

    def bn_bwd_apply(self, dy, x, y, layout, N, C, HW, mean, invstd, gamma, beta, k, relu, want_dres):
        d = self._mask(dy, x, y, mean, invstd, gamma, beta, relu)
        g = gamma if gamma is not None else torch.ones_like(mean)
        xh = (x.float() - self._bc(mean, x)) * self._bc(invstd, x)
        dx = self._bc(g * invstd, x) * (d - self._bc(k[0], x) - xh * self._bc(k[1], x))
        return dx.to(x.dtype), (d.to(x.dtype) if want_dres else None)
