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


def _same_dense_order(a, b):
    """Same element order in memory?  Equal strides, or a [O, I, 1, 1] tensor, whose contiguous and
    channels_last forms coincide in memory while reporting different strides."""
    if a.stride() == b.stride():
        return True
    if a.dim() == 4 and a.shape == b.shape and a.shape[2] == 1 and a.shape[3] == 1:
        def dense(t):
            return t.is_contiguous() or t.is_contiguous(memory_format=torch.channels_last)
        return dense(a) and dense(b)
    return False


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-2, momentum=0.0, weight_decay=0.0):
        # the keys torch.optim.SGD keeps per group, so state dicts load in either direction (checkpoint compat);
        # the values below are the only ones implemented (and the ones the reference uses, train.py:86-89)
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=0, weight_decay=weight_decay,
                                      nesterov=False, maximize=False, foreach=None, differentiable=False,
                                      fused=None))
        self._lr_dev = None
        self._last_lrs = None
        self._plans = {}

    def _sync_lr(self, device):
        """One [n_groups] device vector of learning rates, refreshed only when a group's lr changed.  Every refresh
        stages the values in a FRESH pinned tensor: torch's host allocator does not hand a pinned block out again
        before the asynchronous copy that reads it has executed, so the host may run any number of iterations ahead
        without overwriting a staging buffer that an earlier, still queued H2D copy will read."""
        lrs = [float(g["lr"]) for g in self.param_groups]
        if self._lr_dev is None or self._lr_dev.numel() != len(lrs) or self._lr_dev.device != device:
            self._lr_dev = torch.tensor(lrs, dtype=torch.float32, device=device)     # also after add_param_group
        elif lrs != self._last_lrs:
            host = torch.tensor(lrs, dtype=torch.float32)
            if device.type == "cuda":
                host = host.pin_memory()
            self._lr_dev.copy_(host, non_blocking=True)
        self._last_lrs = lrs

    def refresh_lr(self):
        """Call outside a captured graph after editing param_groups[i]['lr']."""
        p = next((p for g in self.param_groups for p in g["params"]), None)
        if p is not None:
            self._sync_lr(p.device)

    def prepare(self, only=None):
        """Build, OUTSIDE a captured graph, the static tables a later `step(only=...)` over the same subset launches with
        (block maps, the weight-shadow table: host -> device copies, which stream capture refuses).  Every parameter of
        the subset that requires a gradient is assumed to have one at step time."""
        kp = K.provider()
        ids = None if only is None else {id(p) for p in only}
        chosen = [(p, gi) for gi, g in enumerate(self.param_groups) for p in g["params"]
                  if p.requires_grad and (ids is None or id(p) in ids)]
        if not chosen:
            return
        dev = chosen[0][0].device
        self._sync_lr(dev)
        for c0 in range(0, len(chosen), kp.SGD_MAX_SEGS):
            self._plan(c0, chosen[c0:c0 + kp.SGD_MAX_SEGS], dev)
        if dev.type == "cuda" and ids is not None:
            from . import shadow
            shadow.bank.prepare_subset(dev, [p for p, _ in chosen])

    def _plan(self, c0, chunk, device):
        """(numel, group index, block map) of the launch over `chunk` = [(parameter, group index)]: static per model"""
        import numpy as np
        numel = tuple(p.numel() for p, _ in chunk)
        groups = tuple(gi for _, gi in chunk)
        key = (c0, numel, groups, len(self.param_groups))   # the plan stores the group of every segment
        plan = self._plans.get(key)
        if plan is None:
            plan = (np.array(numel, dtype=np.int64), np.array(groups, dtype=np.int32),
                    K.provider().sgd_multi_blockmap(numel, device))
            self._plans[key] = plan
        return plan

    @torch.no_grad()
    def step(self, closure=None, only=None):
        """`only`: an iterable of parameters — update just these (bench.SegmentedStep: the parameters whose gradients are
        final after the first part of the context backward step early, on a side stream beside the rest of the backward;
        a second call with the remaining parameters closes the step).  The update is per element, so a step taken in
        several calls over disjoint subsets equals the one-call step bit for bit."""
        kp = K.provider()
        only_ids = None if only is None else {id(p) for p in only}
        capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
        first = next((p for g in self.param_groups for p in g["params"]), None)
        if first is None:
            return None
        if not capturing:
            self._sync_lr(first.device)
            from .convwrw import join_wrw_stream          # weight gradients still running on the side stream (the engine
            join_wrw_stream()                             # callback has joined them already; this is the belt to its braces:
            #                                               a backward pass that ended in an exception never ran it)
        if len(self.param_groups) > kp.SGD_MAX_GROUPS:      # DFN's train.py builds 18 (dfn train.py:66-73)
            raise K.L.TsgError(f"FusedSGD supports at most {kp.SGD_MAX_GROUPS} parameter groups")
        segs = []                                   # (param view, grad view, buffer, group index)
        for gi, group in enumerate(self.param_groups):
            if group.get("dampening", 0) != 0 or group.get("nesterov", False) or group.get("maximize", False):
                raise K.L.TsgError("FusedSGD implements dampening=0, nesterov=False, maximize=False only")
            for p in group["params"]:
                if p.grad is None or (only_ids is not None and id(p) not in only_ids):
                    continue
                st = self.state[p]
                if p.dtype != torch.float32:
                    raise K.L.TsgError("FusedSGD keeps fp32 master parameters (autocast casts them per op)")
                if not (p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))):
                    raise K.L.TsgError("FusedSGD: parameter is neither contiguous nor channels_last")
                # The update is element-wise, so parameter, gradient and momentum buffer only have to share one
                # dense layout; the buffer keeps the parameter's logical shape (what torch.optim.SGD stores), so
                # optimizer state dicts are interchangeable with the reference's (engine.py:103-137).
                buf = st.get("momentum_buffer")
                if buf is None:
                    buf = st["momentum_buffer"] = torch.zeros_like(p)
                elif not _same_dense_order(buf, p) or buf.dtype != p.dtype or buf.device != p.device:
                    b2 = torch.empty_like(p)          # e.g. restored from a checkpoint written in another layout
                    b2.copy_(buf)
                    buf = st["momentum_buffer"] = b2
                grad = p.grad
                if grad.dtype != torch.float32 or not _same_dense_order(grad, p):
                    g2 = torch.empty_like(p)          # same dense layout as the parameter
                    g2.copy_(grad)
                    grad = g2
                pv, gv, bv = p, grad, buf
                segs.append((pv, gv, bv, gi))
        if not segs:
            return None
        import numpy as np
        mom = np.array([g["momentum"] for g in self.param_groups], dtype=np.float32)
        wd = np.array([g["weight_decay"] for g in self.param_groups], dtype=np.float32)
        cap = kp.SGD_MAX_SEGS
        for c0 in range(0, len(segs), cap):
            chunk = segs[c0:c0 + cap]
            plan = self._plan(c0, [(t[0], t[3]) for t in chunk], first.device)
            ptrs = np.array([[t[0].data_ptr() for t in chunk], [t[1].data_ptr() for t in chunk],
                             [t[2].data_ptr() for t in chunk]], dtype=np.uint64)
            kp.sgd_multi_step_dev(ptrs, plan[0], plan[1], self._lr_dev, mom, wd, plan[2])
        if first.is_cuda:
            from . import shadow                     # the kernel above changed parameters behind torch's back
            shadow.after_external_update(first.device, None if only_ids is None else [t[0] for t in segs])
        return None
