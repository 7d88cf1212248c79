"""TEST-ONLY stand-in kernel provider: implements the SyncBN provider methods in
plain torch (fp64) so that the HOST logic of torchseg_amd.syncbn (message
layout, collective sequencing, count handling, autograd wiring) can run under
gloo on CPU with world_size > 1.  Lives in tests/; the package never selects it."""
import torch


class OracleProvider:
    name = "oracle-test-standin"

    @staticmethod
    def _bc(v, x):
        return v.reshape((1, -1) + (1,) * (x.dim() - 2))

    @staticmethod
    def _axes(x):
        return (0,) + tuple(range(2, x.dim()))

    @staticmethod
    def _sums(partial, S, C):
        p = partial.reshape(-1)[:S * 2 * C].reshape(S, 2, C).double().sum(0)
        return p[0], p[1]

    @staticmethod
    def _count(count, count_dev):
        return float(count_dev[0]) * 4096.0 + float(count_dev[1]) if count_dev is not None else float(count)

    @staticmethod
    def _rows(part64, like):
        """What csrc/bn.hip hands on: one fp32 row for a bf16 tensor; a hi and a lo fp32 row (S = 2) for an fp32 tensor,
        whose sums it accumulates in fp64 (RedAcc<float, .>)."""
        hi = part64.float()
        if like.dtype != torch.float32:
            return hi.unsqueeze(0).contiguous(), 1
        return torch.stack([hi, (part64 - hi.double()).float()]).contiguous(), 2

    def bn_stats(self, x, layout, N, C, HW):
        x64 = x.detach().double()
        return self._rows(torch.stack([x64.sum(self._axes(x)), (x64 * x64).sum(self._axes(x))]), x)

    def bn_collapse(self, partial, S, C, out, count=None):
        s, q = self._sums(partial, S, C)
        out[:C].copy_(s.float())
        out[C:2 * C].copy_(q.float())
        if count is not None:
            out[2 * C] = float(count // 4096)
            out[2 * C + 1] = float(count % 4096)

    def _pack(self, mean, invstd, gamma, beta):
        g = gamma if gamma is not None else torch.ones_like(mean)
        b = beta if beta is not None else torch.zeros_like(mean)
        a = g * invstd
        return torch.stack([a, b - mean * a, mean]).float()

    def bn_finalize(self, partial, S, C, count, count_dev, eps, momentum, gamma, beta, rm, rv, nbt):
        s, q = self._sums(partial, S, C)
        n = self._count(count, count_dev)
        mean = s / n
        sumvar = (q - s * mean).clamp_min(0)
        invstd = (sumvar / n + eps) ** -0.5
        if rm is not None:
            rm.copy_(((1 - momentum) * rm.double() + momentum * mean).float())
        if rv is not None:
            rv.copy_(((1 - momentum) * rv.double() + momentum * sumvar / (n - 1)).float())
        if nbt is not None:
            nbt.add_(1)
        mean, invstd = mean.float(), invstd.float()
        return mean, invstd, self._pack(mean, invstd, gamma, beta)

    def bn_affine(self, mean, invstd, gamma, beta):
        return self._pack(mean, invstd, gamma, beta)

    def bn_apply_fwd(self, x, residual, layout, N, C, HW, fp, relu, out=None):
        y = x.float() * self._bc(fp[0], x) + self._bc(fp[1], x)
        if residual is not None:
            y = y + residual.float()
        if relu:
            y = y.clamp_min(0)
        return y.to(x.dtype)

    def _mask(self, dy, x, y, fp, relu):
        d = dy.float()
        if relu:
            ref = y if y is not None else (x.float() * self._bc(fp[0], x) + self._bc(fp[1], x))
            d = d * (ref > 0)
        return d

    def bn_bwd_reduce(self, dy, x, y, layout, N, C, HW, fp, relu):
        d = self._mask(dy, x, y, fp, relu).double()
        xc = x.double() - self._bc(fp[2].double(), x)
        return self._rows(torch.stack([d.sum(self._axes(x)), (d * xc).sum(self._axes(x))]), x)

    def bn_bwd_coeffs(self, partial, S, C, count, count_dev, batch_stats, invstd, fp, want_param_grads, want_pack):
        s, q = self._sums(partial, S, C)
        is_ = invstd.double()
        dg = (q * is_).float() if want_param_grads else None
        db = s.float() if want_param_grads else None
        bp = None
        if want_pack:
            a = fp[0].double()
            bc = torch.zeros_like(a)
            c2 = torch.zeros_like(a)
            if batch_stats:
                n = self._count(count, count_dev)
                bc = -a * (q * is_ / n) * is_
                c2 = -a * (s / n)
            bp = torch.stack([a, fp[1].double(), fp[2].double(), bc, c2]).float()
        return dg, db, bp

    def bn_bwd_apply(self, dy, x, y, layout, N, C, HW, bp, relu, want_dres):
        d = self._mask(dy, x, y, bp, relu)
        dx = self._bc(bp[0], x) * d + self._bc(bp[3], x) * (x.float() - self._bc(bp[2], x)) + self._bc(bp[4], x)
        return dx.to(x.dtype), (d.to(x.dtype) if want_dres else None)

    # ---- plain-CE / upsample stand-ins: only what tests/test_fusion_cpu.py needs to drive the HOST logic of
    # torchseg_amd.fusion (pattern recognition, deferred values, autograd wiring) without a GPU ----------------
    calls = None          # optional list: names of the provider methods that ran

    def _note(self, name):
        if self.calls is not None:
            self.calls.append(name)

    def ohem_fwd(self, logits, labels, ignore_label, thresh, min_kept, weight):
        assert min_kept == 0, "the CPU stand-in only implements the plain-CE mode"
        self._note("ohem_fwd")
        B, C = logits.shape[:2]
        x = logits.double().reshape(B, C, -1)
        lab = labels.reshape(B, -1).long()
        valid = (lab != ignore_label) & (lab >= 0) & (lab < C)
        n_bad = int(((lab != ignore_label) & ~valid).sum())
        t = torch.where(valid, lab, torch.zeros_like(lab))
        lse = torch.logsumexp(x, 1)
        nll = (lse - x.gather(1, t.unsqueeze(1)).squeeze(1)) * valid
        w = weight.double()[t] * valid if weight is not None else valid.double()
        denom = w.sum()
        loss = ((w * nll).sum() / denom).float().reshape(1)
        sel = torch.zeros(8, dtype=torch.int32)
        sel[1] = sel[2] = int(valid.sum())
        sel[3] = 2
        sel[4:5].view(torch.float32)[0] = float(denom)
        sel[5] = n_bad
        return loss, nll.float().reshape(-1), lse.float().reshape(-1), sel

    def ohem_bwd(self, logits, labels, ignore_label, weight, nll, lse, sel, gscale):
        self._note("ohem_bwd")
        B, C = logits.shape[:2]
        x = logits.double().reshape(B, C, -1)
        lab = labels.reshape(B, -1).long()
        valid = (lab != ignore_label) & (lab >= 0) & (lab < C)
        t = torch.where(valid, lab, torch.zeros_like(lab))
        p = torch.softmax(x, 1)
        p.scatter_add_(1, t.unsqueeze(1), -torch.ones_like(p[:, :1]))
        w = weight.double()[t] * valid if weight is not None else valid.double()
        denom = float(sel[4:5].view(torch.float32)[0])
        return (p * (w * float(gscale[0]) / denom).unsqueeze(1)).reshape(logits.shape).to(logits.dtype)

    def upsample_fwd(self, x, add, OH, OW):
        self._note("upsample_fwd")
        y = torch.nn.functional.interpolate(x, size=(OH, OW), mode="bilinear", align_corners=True)
        return y if add is None else y + add

    # fused upsample + criterion (tsg_ohem_up_*): the interpolation followed by the plain-CE stand-ins above
    def ohem_up_supported(self, z, OH, OW, thresh):
        return z.shape[1] <= 32 and OH >= 2 * z.shape[2] and OW >= 2 * z.shape[3]

    def ohem_up_fwd(self, z, labels, OH, OW, ignore_label, thresh, min_kept, weight):
        calls, self.calls = self.calls, None
        try:
            out = self.ohem_fwd(torch.nn.functional.interpolate(z.float(), size=(OH, OW), mode="bilinear", align_corners=True),
                                labels, ignore_label, thresh, min_kept, weight)
        finally:
            self.calls = calls
        self._note("ohem_up_fwd")
        return out

    def ohem_up_bwd(self, z, labels, OH, OW, ignore_label, weight, nll, lse, sel, gscale):
        self._note("ohem_up_bwd")
        calls, self.calls = self.calls, None
        try:
            full = torch.nn.functional.interpolate(z.float(), size=(OH, OW), mode="bilinear", align_corners=True)
            d = self.ohem_bwd(full, labels, ignore_label, weight, nll, lse, sel, gscale)
            return self._up_bwd(d, z.shape[2], z.shape[3]).to(z.dtype)
        finally:
            self.calls = calls

    def upsample_presum_fwd(self, x, x2, OH, OW):
        self._note("upsample_presum_fwd")
        return torch.nn.functional.interpolate(x + x2, size=(OH, OW), mode="bilinear", align_corners=True)

    def _up_bwd(self, dy, IH, IW):
        N, C = dy.shape[:2]
        return torch.ops.aten.upsample_bilinear2d_backward(dy, [dy.shape[2], dy.shape[3]], [N, C, IH, IW], True, None, None)

    def upsample_bwd(self, dy, IH, IW):
        self._note("upsample_bwd")
        return self._up_bwd(dy, IH, IW)

    def upsample_bwd_nhwc(self, dy, IH, IW):
        self._note("upsample_bwd")
        return self._up_bwd(dy, IH, IW)
