"""Fused SGD on the HIP kernel (csrc/sgd.hip), graph-capturable.

Same update as the `torch.optim.SGD(params_list, lr, momentum, weight_decay)` the
reference builds (model/bisenet/cityscapes.bisenet.R18/train.py:86-89): dampening 0,
no nesterov,  g += wd*p ; buf = momentum*buf + g ; p -= lr*buf.  One launch per
128 parameter tensors (tsg_sgd_multi_step_dev: the pointer table travels in the
kernel arguments; torch's foreach path issues ~7 launches per parameter group plus
scalar-list setup on the host), and the learning rate is read from a device vector so that the
reference's per-iteration `param_groups[i]['lr'] = ...` (train.py:133-139) keeps
working when the whole step is replayed from a hipGraph.
"""
import torch

from . import kernels as K


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-2, momentum=0.0, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self._lr_dev = None
        self._lr_host = None
        self._plans = {}

    def _sync_lr(self, device):
        """One [n_groups] device vector of learning rates, refreshed only when a group's lr changed."""
        lrs = [float(g["lr"]) for g in self.param_groups]
        if self._lr_dev is None:
            self._lr_dev = torch.tensor(lrs, dtype=torch.float32, device=device)
            self._lr_host = torch.tensor(lrs, dtype=torch.float32).pin_memory() if device.type == "cuda" else None
        elif lrs != self._last_lrs:
            if self._lr_host is not None:
                self._lr_host.copy_(torch.tensor(lrs, dtype=torch.float32))
                self._lr_dev.copy_(self._lr_host, non_blocking=True)
            else:
                self._lr_dev.copy_(torch.tensor(lrs, dtype=torch.float32))
        self._last_lrs = lrs

    def refresh_lr(self):
        """Call outside a captured graph after editing param_groups[i]['lr']."""
        p = next((p for g in self.param_groups for p in g["params"]), None)
        if p is not None:
            self._sync_lr(p.device)

    @torch.no_grad()
    def step(self, closure=None):
        kp = K.provider()
        capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
        first = next((p for g in self.param_groups for p in g["params"]), None)
        if first is None:
            return None
        if not capturing:
            self._sync_lr(first.device)
        if len(self.param_groups) > 16:
            raise K.L.TsgError("FusedSGD supports at most 16 parameter groups")
        segs = []                                   # (param view, grad view, buffer, group index)
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if "momentum_buffer" not in st:
                    st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                grad = p.grad
                if p.dtype != torch.float32:
                    raise K.L.TsgError("FusedSGD keeps fp32 master parameters (autocast casts them per op)")
                if grad.dtype != torch.float32 or grad.stride() != p.stride():
                    g2 = torch.empty_like(p)          # same dense layout as the parameter
                    g2.copy_(grad)
                    grad = g2
                buf = st["momentum_buffer"]
                if p.is_contiguous():
                    pv, gv, bv = p, grad, buf
                elif p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last):
                    # the update is element-wise: address the dense NHWC storage directly
                    pv, gv = p.permute(0, 2, 3, 1), grad.permute(0, 2, 3, 1)
                    if buf.shape != pv.shape:
                        st["momentum_buffer"] = buf = torch.zeros(pv.shape, dtype=p.dtype, device=p.device)
                    bv = buf
                else:
                    raise K.L.TsgError("FusedSGD: parameter is neither contiguous nor channels_last")
                segs.append((pv, gv, bv, gi))
        if not segs:
            return None
        import numpy as np
        mom = np.array([g["momentum"] for g in self.param_groups], dtype=np.float32)
        wd = np.array([g["weight_decay"] for g in self.param_groups], dtype=np.float32)
        cap = kp.SGD_MAX_SEGS
        for c0 in range(0, len(segs), cap):
            chunk = segs[c0:c0 + cap]
            numel = tuple(t[0].numel() for t in chunk)
            key = (c0, numel)
            plan = self._plans.get(key)
            if plan is None:                        # static per model: sizes, groups, block map
                plan = (np.array(numel, dtype=np.int64), np.array([t[3] for t in chunk], dtype=np.int32),
                        kp.sgd_multi_blockmap(numel, first.device))
                self._plans[key] = plan
            ptrs = np.array([[t[0].data_ptr() for t in chunk], [t[1].data_ptr() for t in chunk],
                             [t[2].data_ptr() for t in chunk]], dtype=np.uint64)
            kp.sgd_multi_step_dev(ptrs, plan[0], plan[1], self._lr_dev, mom, wd, plan[2])
        return None
