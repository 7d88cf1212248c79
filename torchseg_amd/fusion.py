"""The TorchFunctionMode our DistributedDataParallel wrapper enters around an UNCHANGED network.py
forward (SURVEY.md 8b "the one choke point we own").  The reference's model files call ATen directly, so
the operator fusions of the hot path have to be recognised at the torch-function level:

  * a5  `nn.CrossEntropyLoss(ignore_index=...)` / `F.cross_entropy` on [B,C,H,W] logits (dfn train.py:48-49,
        network.py:140-143) and the same criterion applied to `F.log_softmax(x, dim=1)` (pspnet / psanet
        network.py:50-56) -> the OHEM kernels in plain-CE mode (`tsg_ohem_fwd/bwd`, min_kept = 0).  CE of a
        log-softmaxed input equals CE of the logits (log_softmax is idempotent), so the deferred log_softmax is
        never evaluated on the training path: the full-resolution logits are read once forward, once backward.
  * a8  `fm += last_fm` followed by `F.interpolate(fm, ..., 'bilinear', align_corners=True)` (bisenet
        network.py:91-95) -> one upsample kernel that sums the two maps while it reads the taps
        (`tsg_upsample_bilinear_ac*_presum_fwd`); and `last_fm + F.interpolate(fm)` (dfn network.py:130-133)
        stays `tsg_upsample_bilinear_ac*_fwd(x, add)`.
  * a9  `torch.bmm(x, torch.softmax(a, dim=1))` (psanet network.py:125-126,135-136) -> `tsg_psa_*` (psa.py).
  * a3  consecutive `ConvBnRelu` modules called one after the other from a network.py (bisenet network.py:131-137,
        SpatialPath) -> the BatchNorm + ReLU of the first is applied while the convolution of the second loads its
        input (`PendingCbr`: what `seg_oprs.cbr_chain` does for our own builders); the image stem, its BatchNorm and the
        next convolution become the one autograd node of convwrw.stem_bn_relu_conv.

Every deferred value materialises itself (with the eager semantics, including the in-place update of `fm`) the
moment anything other than its fusing consumer touches it.  The `+=` deferral changes WHEN (and, on the fused path,
whether) `fm` itself is updated, so it is only taken when nothing else can observe `fm`: the tensor must be referenced
by nothing but the statement's own variable (an alias in a list, an attribute or a second name keeps the eager in-place
add), and the fused consumer checks the version counters of both addends — an in-place change of either between the
`+=` and the `F.interpolate` raises instead of producing a silently different sum.
"""
import dis
import sys

import torch
import torch.nn.functional as F
from torch.overrides import TorchFunctionMode

from . import kernels as K

_FLOATS = (torch.float32, torch.bfloat16)
_TARGET_ON_DEVICE = True      # tests/ flip this (together with _is_map) to drive the host logic on CPU


class _Deferred(object):
    """A value that has not been computed yet; any torch function applied to it computes it first."""

    _value = None

    def materialize(self):
        raise NotImplementedError

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        return func(*_unwrap(args), **_unwrap(kwargs or {}))

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return getattr(self.materialize(), name)

    def __getitem__(self, idx):
        return self.materialize()[idx]


def _forward_dunder(name):
    def op(self, *others):
        return getattr(self.materialize(), name)(*_unwrap(others))
    op.__name__ = name
    return op


# Python looks operators up on the TYPE (never through __getattr__): `fm += x; y = fm * 2` must find them
for _n in ("add", "sub", "mul", "truediv", "floordiv", "mod", "pow", "matmul", "and", "or", "xor"):
    for _fmt in ("__%s__", "__r%s__", "__i%s__"):
        setattr(_Deferred, _fmt % _n, _forward_dunder(_fmt % _n))
for _n in ("__neg__", "__pos__", "__abs__", "__invert__", "__lt__", "__le__", "__gt__", "__ge__", "__eq__", "__ne__",
           "__len__", "__iter__", "__bool__", "__float__", "__int__", "__setitem__", "__contains__"):
    setattr(_Deferred, _n, _forward_dunder(_n))
_Deferred.__hash__ = object.__hash__


def _unwrap(v):
    if isinstance(v, _Deferred):
        return v.materialize()
    if isinstance(v, (list, tuple)):
        return type(v)(_unwrap(u) for u in v)
    if isinstance(v, dict):
        return {k: _unwrap(u) for k, u in v.items()}
    return v


def materialize(v):
    """Public form of _unwrap (the DDP wrapper applies it to whatever the wrapped forward returns)."""
    from .psa import _DeferredColSoftmax
    from .upsample import DeferredUpsample
    if isinstance(v, (_DeferredColSoftmax, DeferredUpsample)):
        return v.materialize()
    if isinstance(v, (list, tuple)):
        return type(v)(materialize(u) for u in v)
    return _unwrap(v)


class DeferredLogSoftmax(_Deferred):
    """F.log_softmax(x, dim=1) of a [B,C,H,W] tensor (or of a pending up-sampling of one), pending."""

    def __init__(self, x):
        self.x = x

    def materialize(self):
        if self._value is None:
            from .upsample import DeferredUpsample
            x = self.x.materialize() if isinstance(self.x, DeferredUpsample) else self.x
            self._value = torch._log_softmax(x, 1, False) if x.dtype == torch.float32 else torch.log_softmax(x, 1)
        return self._value


class DeferredSum(_Deferred):
    """`a += b` of two same-shaped [N,C,H,W] maps, pending.  Materialising performs the in-place add on `a`."""

    def __init__(self, a, b):
        self.a, self.b = a, b
        self.versions = (a._version, b._version)

    def unchanged(self):
        return (self.a._version, self.b._version) == self.versions

    def materialize(self):
        if self._value is None:
            if not self.unchanged():
                # FuseMode performs the pending add BEFORE any in-place operation it sees on either addend (eager
                # semantics are kept for `fm += x; x.zero_(); use(fm)`); what reaches this line changed a tensor behind
                # the mode's back (`.data`, a raw kernel), and the eager result can no longer be reconstructed
                raise RuntimeError("torchseg_amd.fusion: a tensor of a pending `a += b` was modified in place before the "
                                   "sum was used, outside the view of the fusion mode; set TSG_FUSE_ADD_UP=0 for this model")
            self._value = self.a.add_(self.b)
        return self._value


class PendingCbr(_Deferred):
    """The output of a furnace `ConvBnRelu` (conv -> SyncBatchNorm -> ReLU, seg_oprs.py:24-46) whose BatchNorm + ReLU —
    and, while `y` is None, whose convolution — has not run yet.  The NEXT ConvBnRelu consumes it (`feed`): the pair then
    runs exactly what seg_oprs.cbr_chain runs for the same two modules.  Any other consumer (a torch function, an
    attribute, another module: FuseMode's forward pre-hook) materialises it with the eager module sequence."""

    def __init__(self, mod, x=None, y=None):
        self.mod, self.x, self.y = mod, x, y
        self.consumed = False

    def materialize(self):
        if self._value is None:
            if self.consumed:
                # its BatchNorm already ran inside the consumer's fused node (running statistics updated once, the
                # normalised activation never stored): a second evaluation would update them twice
                raise RuntimeError("torchseg_amd.fusion: the output of a ConvBnRelu was consumed by the next ConvBnRelu "
                                   "(BatchNorm applied on its load path) and is used again; set TSG_FUSE_CHAIN=0 for this model")
            from .furnace_glue import norm_act
            m = self.mod
            y = m.conv(self.x) if self.y is None else self.y
            self._value = norm_act(m.bn, m.relu, y)
            self.x = self.y = None
            stats["cbr_materialized"] += 1
        return self._value

    def feed(self, nxt):
        """-> nxt.conv(relu(bn(conv(x)))), the BatchNorm + ReLU applied on the load path of nxt.conv where covered."""
        if self._value is not None or self.consumed:
            return nxt.conv(self.materialize())
        from .convwrw import bn_relu_conv, stem_bn_relu_conv
        m = self.mod
        y = self.y
        if y is None:
            out = stem_bn_relu_conv(m.conv, m.bn, m.relu, self.x, nxt.conv)
            if out is not None:
                self.consumed = True
                self.x = None
                stats["cbr_stem_fused"] += 1
                return out
            y = m.conv(self.x)
        out = bn_relu_conv(m.bn, m.relu, y, nxt.conv)
        self.consumed = True
        self.x = self.y = None
        stats["cbr_fed"] += 1
        return out


def _on_device(t):
    """tests/ replace this to drive the chain logic on CPU tensors."""
    return t.is_cuda


CHAIN_ACTIVE = False       # set while a FuseMode(chain=True) is entered: ConvBnRelu.forward may return a PendingCbr
MODE_DEPTH = 0             # FuseModes entered: `outside_mode` forwards step out of torch-function dispatch while it is > 0


def outside_mode(fn=None, keeps_chain=False):
    """Decorator for the forward of OUR furnace modules (ResNet, ConvBnRelu, AttentionRefinement, FeatureFusion, the
    criteria): nothing inside them is a call pattern FuseMode has to see — the patterns live in the reference's network.py —
    but the mode is consulted for every torch call they make (tensor attributes and the allocations inside the kernel
    wrappers included: ~5 000 per BiSeNet forward, ~0.4 us each).  While a FuseMode is entered the wrapped forward runs
    under torch._C.DisableTorchFunction(); deferred arguments have been materialised by the mode's forward pre-hook before
    that.  Without a FuseMode (our own builders, evaluation) the wrapper costs one global read.

    keeps_chain (ConvBnRelu.forward only): the module is itself a link of a ConvBnRelu chain and reads CHAIN_ACTIVE.  Every
    other wrapped forward clears it for its duration: a 64-output ConvBnRelu NESTED in one of our modules must not hand a
    PendingCbr to code that runs with torch-function dispatch disabled (channel_scale, an autograd.Function.apply — the
    non-tensor argument would drop the gradient or raise) — chains are a pattern of network.py level."""
    import functools

    def deco(fn):
        @functools.wraps(fn)
        def wrapper(*args, **kwargs):
            global CHAIN_ACTIVE
            if MODE_DEPTH <= 0:
                return fn(*args, **kwargs)
            if kwargs:                                   # the mode's forward pre-hook sees positional arguments only
                kwargs = {k: (v.materialize() if isinstance(v, (PendingCbr, DeferredSum)) else v) for k, v in kwargs.items()}
            chain = CHAIN_ACTIVE
            if not keeps_chain:
                CHAIN_ACTIVE = False
            try:
                with torch._C.DisableTorchFunction():
                    return fn(*args, **kwargs)
            finally:
                CHAIN_ACTIVE = chain
        return wrapper
    return deco if fn is None else deco(fn)


# ---- a sub-module of the wrapped model on a side stream (TSG_FORK_MODULES, opt-in) --------------------------------
# BiSeNet's two paths share nothing until the fusion module (bisenet network.py:76-99): the detail branch — large maps,
# HBM-bound BatchNorm passes and stems — runs on a side HIP stream beside the context path's deep layers (small maps,
# matrix-core bound, too few tiles to fill the chip on their own); autograd replays every node on its forward stream, so the
# backward overlaps the same way.  The wrapper (ddp.py) registers the two hooks below on the named sub-modules; the output
# is joined the moment a module (the FuseMode's global forward pre-hook) or a torch function at network.py level (the
# mode's slow path) receives it.
_FORKED = {}               # id(output tensor) -> (weak reference, side stream, the stream the forward came from)
_FORK_SIDE = {}
_FORK_STACK = []


def fork_pre_hook(mod, args):
    if MODE_DEPTH <= 0 or not args or not isinstance(args[0], torch.Tensor) or not args[0].is_cuda \
            or torch.cuda.is_current_stream_capturing():
        _FORK_STACK.append(None)
        return None
    dev = args[0].device
    cur = torch.cuda.current_stream(dev)
    side = _FORK_SIDE.get(dev)
    if side is None:
        side = _FORK_SIDE[dev] = torch.cuda.Stream(device=dev)
    side.wait_stream(cur)
    for a in args:
        if isinstance(a, torch.Tensor) and a.is_cuda:
            a.record_stream(side)
    torch.cuda.set_stream(side)
    _FORK_STACK.append((cur, side))
    return None


def fork_post_hook(mod, args, out):
    st = _FORK_STACK.pop() if _FORK_STACK else None
    if st is None:
        return None
    cur, side = st
    torch.cuda.set_stream(cur)
    if isinstance(out, _Deferred):                      # its kernels go to `cur`, its operands were made on `side`
        cur.wait_stream(side)
    out = _unwrap(out)
    if isinstance(out, torch.Tensor):
        import weakref
        _FORKED[id(out)] = (weakref.ref(out), side, cur)
        stats["forked"] += 1
    else:                                               # not a single tensor: nothing to track, join at once
        cur.wait_stream(side)
    return out


def _join_forked(args):
    for a in args:
        ent = _FORKED.get(id(a)) if isinstance(a, torch.Tensor) else None
        if ent is not None and ent[0]() is a:
            del _FORKED[id(a)]
            cur = torch.cuda.current_stream(a.device)
            cur.wait_stream(ent[1])
            a.record_stream(cur)


def _unwrap_pending_inputs(mod, args):
    """Global forward pre-hook while a FuseMode is entered: a pending ConvBnRelu output or a pending `a += b` handed to a
    MODULE is the real tensor by the time the module's forward runs (which may run outside torch-function dispatch:
    `outside_mode`) — except a PendingCbr handed to the next ConvBnRelu, which feeds on it."""
    if _FORKED:
        _join_forked(args)
    for a in args:
        if isinstance(a, (PendingCbr, DeferredSum)):
            break
    else:
        return None
    accepts = getattr(mod, "tsg_accepts_pending", False)
    return tuple(a.materialize() if (isinstance(a, DeferredSum) or (isinstance(a, PendingCbr) and not accepts)) else a
                 for a in args)


class _PresumUpFn(torch.autograd.Function):
    """up(a + b); both addends get the same gradient (one gather kernel)."""

    @staticmethod
    def forward(ctx, a, b, OH, OW):
        from .upsample import _is_cl_dense, _vec_ok
        ctx.in_hw = (a.shape[2], a.shape[3])
        if _is_cl_dense(a) and _vec_ok(a):
            b = b.contiguous(memory_format=torch.channels_last)
        else:
            a, b = a.contiguous(), b.contiguous()
        return K.provider().upsample_presum_fwd(a, b, OH, OW)

    @staticmethod
    def backward(ctx, dy):
        from .upsample import _backward_any_layout
        d = _backward_any_layout(dy, *ctx.in_hw)
        return (d if ctx.needs_input_grad[0] else None), (d if ctx.needs_input_grad[1] else None), None, None


def upsample_presum(a, b, size=None, scale_factor=None):
    """F.interpolate(a + b, bilinear, align_corners=True) without materialising the sum (a8)."""
    from .upsample import _out_size
    OH, OW = _out_size(a, size, scale_factor)
    return _PresumUpFn.apply(a, b, OH, OW)


def _is_map(t):
    return isinstance(t, torch.Tensor) and t.is_cuda and t.dim() == 4 and t.dtype in _FLOATS


def _is_logits(t):
    """A [B,C,H,W] map or the pending bilinear up-sampling of one (the criteria take both)."""
    from .upsample import DeferredUpsample
    return _is_map(t) or (isinstance(t, DeferredUpsample) and _is_map(t.z))


_OP_INPLACE_ADD = dis.opmap.get("INPLACE_ADD")            # <= 3.10
_OP_BINARY = dis.opmap.get("BINARY_OP")                   # >= 3.11: argument 13 = NB_INPLACE_ADD


def _caller_runs_augmented_add(depth=2):
    """True when the Python frame that triggered the current torch function is executing `x += y` — the statement that
    rebinds `x` to whatever we return.  `x.add_(y)` dispatches to the same torch function but drops the result, so a
    deferred value would be lost: only the augmented assignment may be deferred."""
    try:
        f = sys._getframe(depth)
        code, i = f.f_code.co_code, f.f_lasti
        if i < 0 or i >= len(code):
            return False
        if _OP_INPLACE_ADD is not None and code[i] == _OP_INPLACE_ADD:
            return True
        return _OP_BINARY is not None and code[i] == _OP_BINARY and code[i + 1] == 13
    except Exception:                                       # noqa: BLE001 - no frame introspection: never defer
        return False


def _calibrate_iadd_refs():
    """sys.getrefcount of the left operand of `t += u`, seen from a TorchFunctionMode, when `t` is referenced by the
    statement's own local variable only.  Anything above this count means an alias exists somewhere."""
    seen = []

    class _Probe(TorchFunctionMode):
        def __torch_function__(self, func, types, args=(), kwargs=None):
            if func in _IADD_FUNCS:
                seen.append(sys.getrefcount(args[0]))
            return func(*args, **(kwargs or {}))

    def stmt():
        t = torch.zeros(1) * 1.0
        u = torch.zeros(1)
        t += u
        return t

    with _Probe():
        stmt()
    return seen[0] if seen else 0


def _storage_ptr(t):
    try:
        return t.untyped_storage().data_ptr()
    except Exception:                                       # noqa: BLE001 - meta / wrapper tensors: identity only
        return -1 - id(t)


def _mutates(func, kwargs):
    """An in-place torch function: `x.zero_()`, `x += y` / `x.add_(y)`, `x[i] = v`, or anything called with out=."""
    name = getattr(func, "__name__", "") or ""
    if kwargs and (kwargs.get("out") is not None or kwargs.get("inplace")):
        return True                                      # F.relu(x, inplace=True) dispatches as 'relu' 
    if name.startswith("__") and name.endswith("__"):
        return name == "__setitem__" or (name.startswith("__i") and name not in ("__index__", "__int__", "__invert__",
                                                                                "__init__", "__iter__"))
    return name.endswith("_")


# What the `+=` deferral did, for logs and tests: it turns itself off silently whenever the interpreter does not behave
# like the one it was calibrated on (reference counts of the augmented assignment, the BINARY_OP bytecode), so the
# counts say whether the fused path is actually taken
stats = {"iadd_deferred": 0, "iadd_declined_alias": 0, "iadd_declined_not_augmented": 0, "presum_fused": 0,
         "materialized_before_mutation": 0, "head_deferred": 0, "ce_fused": 0, "psa_deferred": 0,
         "cbr_deferred": 0, "cbr_fed": 0, "cbr_stem_fused": 0, "cbr_materialized": 0, "forked": 0}


_LOG_SOFTMAX_FUNCS = (F.log_softmax, torch.log_softmax, torch.Tensor.log_softmax)
_IADD_FUNCS = (torch.Tensor.__iadd__, torch.Tensor.add_)
_ADD_FUNCS = (torch.Tensor.__add__, torch.Tensor.__radd__, torch.Tensor.add, torch.add)
_IADD_BASE_REFS = _calibrate_iadd_refs()
if _IADD_BASE_REFS <= 0 or (_OP_INPLACE_ADD is None and _OP_BINARY is None):
    import logging as _logging
    _logging.getLogger(__name__).warning(
        "torchseg_amd.fusion: the `+=` -> interpolate fusion is OFF on this interpreter (reference-count calibration %r, "
        "no INPLACE_ADD / BINARY_OP opcode): `fm += x; F.interpolate(fm)` runs as two kernels", _IADD_BASE_REFS)


def _ce_args(args, kwargs):
    """Normalised (input, target, weight, ignore_index) of an F.cross_entropy / F.nll_loss call, or None when the
    call uses something the kernels do not implement (then the stock op runs, on materialised inputs)."""
    names = ("input", "target", "weight", "size_average", "ignore_index", "reduce", "reduction", "label_smoothing")
    a = dict(zip(names, args))
    a.update(kwargs)
    if a.get("size_average") is not None or a.get("reduce") is not None or a.get("reduction", "mean") != "mean":
        return None
    if a.get("label_smoothing", 0.0) != 0.0:
        return None
    inp, tgt, w = a.get("input"), a.get("target"), a.get("weight")
    x = inp.x if isinstance(inp, DeferredLogSoftmax) else inp
    if not _is_logits(x) or not isinstance(tgt, torch.Tensor) or tgt.dim() != 3 or (_TARGET_ON_DEVICE and not tgt.is_cuda):
        return None
    if tgt.dtype not in (torch.int64, torch.uint8) or tuple(tgt.shape) != (x.shape[0], x.shape[2], x.shape[3]):
        return None
    if w is not None and (not isinstance(w, torch.Tensor) or w.numel() != x.shape[1]):
        return None
    return x, tgt, w, int(a.get("ignore_index", -100))


class FuseMode(TorchFunctionMode):
    """See the module docstring.  `psa`: defer column softmaxes for the PSA contraction; `loss`: plain CE heads on
    the HIP kernels; `add_up`: `+=` -> interpolate fusion; `head`: bilinear up-sampling of <= 32-channel logits by >= 4 is
    left pending for the criterion (fused upsample + CE / OHEM kernels); `chain`: consecutive ConvBnRelu modules hand
    their BatchNorm + ReLU to the next convolution (PendingCbr).

    Host cost: the mode sees EVERY torch function of the forward (a few thousand per step: tensor attributes and the
    allocations inside our own kernel wrappers included), so the first thing it does is one set lookup that sends
    everything it has no business with straight on."""

    def __init__(self, psa=False, loss=True, add_up=True, head=False, chain=False):
        super().__init__()
        self.psa, self.loss, self.add_up, self.head = bool(psa), bool(loss), bool(add_up), bool(head)
        self.chain = bool(chain)
        self._pending = []           # weak references to DeferredSums that have not been used yet
        self._cbrs = []              # PendingCbrs handed out (settled at exit: a BatchNorm must not be skipped)
        self._hook = None
        watch = set()
        if self.psa:
            from .psa import _SOFTMAX_FUNCS
            watch.update(_SOFTMAX_FUNCS)
        if self.loss:
            watch.update(_LOG_SOFTMAX_FUNCS)
            watch.update((F.cross_entropy, F.nll_loss))
        if self.add_up:
            watch.update(_IADD_FUNCS)
        if self.add_up or self.head:
            watch.add(F.interpolate)
        self._watch = frozenset(watch)

    def __enter__(self):
        global CHAIN_ACTIVE, MODE_DEPTH
        import torch.nn.modules.module as _mm
        self._hook = _mm.register_module_forward_pre_hook(_unwrap_pending_inputs)
        MODE_DEPTH += 1
        if self.chain:
            self._chain_before = CHAIN_ACTIVE
            CHAIN_ACTIVE = self
        return super().__enter__()

    def __exit__(self, *exc):
        global CHAIN_ACTIVE, MODE_DEPTH
        cbrs, self._cbrs = self._cbrs, []
        MODE_DEPTH -= 1
        if self.chain:
            CHAIN_ACTIVE = self._chain_before
        if self._hook is not None:
            self._hook.remove()
            self._hook = None
        out = super().__exit__(*exc)
        if exc[0] is not None and _FORK_STACK:           # an exception inside a forked module: its post-hook never ran
            for st in _FORK_STACK:
                if st is not None:
                    torch.cuda.set_stream(st[0])
                    st[0].wait_stream(st[1])
                    break
            del _FORK_STACK[:]
        if _FORKED and MODE_DEPTH <= 0:                  # never consumed inside the forward: join before anybody else can
            for ref, side, cur in list(_FORKED.values()):
                cur.wait_stream(side)
            _FORKED.clear()
        if self.chain and exc[0] is None:
            for c in cbrs:                               # never consumed: the eager program had run its BatchNorm
                if c._value is None and not c.consumed:
                    c.materialize()
        return out

    def defer_cbr(self, mod, x=None, y=None):
        c = PendingCbr(mod, x, y)
        self._cbrs.append(c)
        stats["cbr_deferred"] += 1
        return c

    def _settle_before_mutation(self, func, args, kwargs):
        """The pending `a += b` happens NOW if `func` is about to change `a` or `b` in place: the eager program had
        already consumed them at the `+=`."""
        live = []
        for ref in self._pending:
            sm = ref()
            if sm is None or sm._value is not None:
                continue
            live.append(ref)
        self._pending = live
        if not live or not _mutates(func, kwargs):
            return
        flat = [t for t in list(args) + list(kwargs.values()) if isinstance(t, torch.Tensor)]
        for ref in live:
            sm = ref()
            if sm is None:
                continue
            # identity, or the same storage (a view of an addend changed in place changes the addend: ADVICE r4)
            sa, sb = _storage_ptr(sm.a), _storage_ptr(sm.b)
            if any(t is sm.a or t is sm.b or _storage_ptr(t) in (sa, sb) for t in flat):
                sm.materialize()
                stats["materialized_before_mutation"] += 1

    def __torch_function__(self, func, types, args=(), kwargs=None):
        if func not in self._watch and not self._pending and not _FORKED:
            return func(*args, **kwargs) if kwargs else func(*args)
        kwargs = kwargs or {}
        if _FORKED:
            _join_forked(args)
            if kwargs:
                _join_forked(tuple(kwargs.values()))
            for a in args:
                if isinstance(a, (list, tuple)):
                    _join_forked(a)
        if self.add_up and self._pending:
            self._settle_before_mutation(func, args, kwargs)
        if self.psa:
            from .psa import _SOFTMAX_FUNCS, _DeferredColSoftmax
            if func in _SOFTMAX_FUNCS and args and isinstance(args[0], torch.Tensor):
                dim = kwargs.get("dim", args[1] if len(args) > 1 else None)
                a = args[0]
                if dim == 1 and a.dim() == 3 and a.is_cuda and kwargs.get("dtype") is None and a.dtype in _FLOATS:
                    stats["psa_deferred"] += 1
                    return _DeferredColSoftmax(a)
        if self.loss:
            if func in _LOG_SOFTMAX_FUNCS and args and _is_logits(args[0]) and torch.is_grad_enabled() \
                    and args[0].requires_grad:
                dim = kwargs.get("dim", args[1] if len(args) > 1 else None)
                if dim == 1 and kwargs.get("dtype") is None:
                    return DeferredLogSoftmax(args[0])
            if func is F.cross_entropy or (func is F.nll_loss and args and isinstance(args[0], DeferredLogSoftmax)):
                ce = _ce_args(args, kwargs)
                if ce is not None:
                    from .losses import cross_entropy_2d
                    x, tgt, w, ignore = ce
                    stats["ce_fused"] += 1
                    return cross_entropy_2d(x, tgt, ignore_index=ignore, weight=w)
        if self.add_up:
            if func in _IADD_FUNCS and len(args) == 2 and not kwargs and _is_map(args[0]) and _is_map(args[1]) \
                    and args[0].shape == args[1].shape and args[0].dtype == args[1].dtype \
                    and args[0].grad_fn is not None and torch.is_grad_enabled():
                # an augmented assignment whose target is referenced by its own variable only: nobody can observe when
                # (or whether) the tensor is updated.  Aliased tensors, attributes, `.add_()` calls keep the eager add.
                if sys.getrefcount(args[0]) > _IADD_BASE_REFS:
                    stats["iadd_declined_alias"] += 1
                elif not _caller_runs_augmented_add():
                    stats["iadd_declined_not_augmented"] += 1
                else:
                    import weakref
                    sm = DeferredSum(args[0], args[1])
                    self._pending.append(weakref.ref(sm))
                    stats["iadd_deferred"] += 1
                    return sm
            if func is F.interpolate and args and isinstance(args[0], DeferredSum):
                s = args[0]
                size = kwargs.get("size", args[1] if len(args) > 1 else None)
                scale = kwargs.get("scale_factor", args[2] if len(args) > 2 else None)
                mode = kwargs.get("mode", args[3] if len(args) > 3 else "nearest")
                ac = kwargs.get("align_corners", args[4] if len(args) > 4 else None)
                if s._value is None and s.unchanged() and mode == "bilinear" and ac \
                        and (size is not None or scale is not None) and not kwargs.get("antialias", False):
                    stats["presum_fused"] += 1
                    return upsample_presum(s.a, s.b, size=size, scale_factor=scale)
        if self.head and func is F.interpolate and args and _is_map(args[0]) and torch.is_grad_enabled() \
                and args[0].requires_grad and kwargs.get("mode") == "bilinear" and kwargs.get("align_corners") \
                and not kwargs.get("antialias", False):
            # the last statement of a head: logits up-sampled by >= 4 for the criterion.  Returned pending: our
            # criteria evaluate the interpolation inside their kernels (tsg_ohem_up_*), anything else materialises it
            from .upsample import DeferredUpsample, _out_size
            x = args[0]
            OH, OW = _out_size(x, kwargs.get("size", args[1] if len(args) > 1 else None),
                               kwargs.get("scale_factor", args[2] if len(args) > 2 else None))
            if x.shape[1] <= 32 and OH >= 4 * x.shape[2] and OW >= 4 * x.shape[3]:
                stats["head_deferred"] += 1
                return DeferredUpsample(x, (OH, OW))
        return func(*args, **kwargs)
