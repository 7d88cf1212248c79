"""PSANet collect / distribute attention on the MFMA kernels.

`psa_attention(X, A)` == `torch.bmm(X, torch.softmax(A, dim=1))`
(model/psanet/ade.psanet.R101_v1c/network.py:125-126,135-136) with autograd.

The reference's network.py calls torch.softmax and torch.bmm directly, so
`FusePsaMode` (a TorchFunctionMode entered by our DistributedDataParallel
wrapper when the model contains a PointwiseSpatialAttention block) defers
`softmax(A, dim=1)` on a 3-D HIP tensor and fuses it into the `torch.bmm` that
consumes it; any other consumer simply materialises the softmax.
"""
import torch
from torch.overrides import TorchFunctionMode

from . import kernels as K


class _PsaFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, A):
        kp = K.provider()
        if X.dtype != A.dtype:
            A = A.to(X.dtype)
        X = X.contiguous()
        A = A.contiguous()
        out, lse = kp.psa_fwd(X, A)
        ctx.save_for_backward(X, A, out, lse)
        return out

    @staticmethod
    def backward(ctx, dout):
        kp = K.provider()
        X, A, out, lse = ctx.saved_tensors
        dX, dA = kp.psa_bwd(X, A, out, dout.contiguous().to(X.dtype), lse)
        return dX, dA


def psa_attention(X, A):
    """X [B, Cx, K], A [B, K, N] -> [B, Cx, N] = X @ softmax(A, dim=1)."""
    if X.dim() != 3 or A.dim() != 3 or X.shape[0] != A.shape[0] or X.shape[2] != A.shape[1]:
        raise ValueError(f"psa_attention: incompatible shapes {tuple(X.shape)} x {tuple(A.shape)}")
    return _PsaFn.apply(X, A)


def psa_supported(X, A):
    return (X.is_cuda and X.dtype in (torch.float32, torch.bfloat16) and X.dim() == 3 and A.dim() == 3
            and X.shape[1] % 8 == 0 and A.shape[1] % 8 == 0 and A.shape[2] % 8 == 0)


class _DeferredColSoftmax(object):
    """softmax(A, dim=1) that has not been computed yet."""

    def __init__(self, a):
        self.a = a
        self._value = None

    def materialize(self):
        if self._value is None:
            # aten-level entry point: not one of _SOFTMAX_FUNCS, so an active FusePsaMode lets it through
            self._value = torch._softmax(self.a, 1, False)
        return self._value

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is torch.bmm and len(args) == 2 and isinstance(args[1], cls) and isinstance(args[0], torch.Tensor) \
                and psa_supported(args[0], args[1].a):
            return psa_attention(args[0], args[1].a)
        unwrap = lambda v: v.materialize() if isinstance(v, cls) else v  # noqa: E731
        return func(*[unwrap(a) for a in args], **{k: unwrap(v) for k, v in kwargs.items()})

    def __getattr__(self, name):
        return getattr(self.materialize(), name)


_SOFTMAX_FUNCS = (torch.softmax, torch.Tensor.softmax, torch.nn.functional.softmax)


class FusePsaMode(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _SOFTMAX_FUNCS and args and isinstance(args[0], torch.Tensor):
            dim = kwargs.get("dim", args[1] if len(args) > 1 else None)
            a = args[0]
            if dim == 1 and a.dim() == 3 and a.is_cuda and kwargs.get("dtype") is None \
                    and a.dtype in (torch.float32, torch.bfloat16):
                return _DeferredColSoftmax(a)
        return func(*args, **kwargs)


def model_has_psa(module):
    return any(type(m).__name__ == "PointwiseSpatialAttention" for m in module.modules())
