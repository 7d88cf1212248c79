"""Tensor-level wrappers over the C-ABI (one method per libtsg_hip entry point).

`HipKernels` is the only kernel provider the product registers.  The host
logic (syncbn.py, ohem.py, ...) talks to a provider object so that the
multi-process CPU tests can drive the same host logic with a stand-in provider
that lives under tests/ — the package itself never falls back to anything.
"""
import ctypes as C
import os

import torch

from . import _lib as L


# re-exported so host modules can name layout codes as K.L.NCHW / K.L.NHWC


def bn_layout(x):
    """(layout, N, C, HW) of a dense activation, or None if it must be copied.

    [N,C,H,W] contiguous -> NCHW; channels_last -> NHWC; HW == 1 is both, and the
    NHWC kernels are the coalesced choice there.
    """
    if x.dim() == 2:
        return L.NHWC, x.shape[0], x.shape[1], 1
    n, c = x.shape[0], x.shape[1]
    hw = 1
    for s in x.shape[2:]:
        hw *= s
    if hw == 1 and x.is_contiguous():
        return L.NHWC, n, c, 1
    if x.is_contiguous():
        return L.NCHW, n, c, hw
    if x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last):
        return L.NHWC, n, c, hw
    return None


def _require_contiguous(*tensors):
    for t in tensors:
        if t is not None and not t.is_contiguous():
            raise L.TsgError("kernel wrapper needs a contiguous tensor, got strides %s for shape %s"
                             % (tuple(t.stride()), tuple(t.shape)))


def _label_code(t):
    if t.dtype == torch.int64:
        return L.I64
    if t.dtype == torch.uint8:
        return L.U8
    raise L.TsgError(f"labels must be int64 or uint8, got {t.dtype}")



# TSG_WEIGHT_SHADOW=1|0 (default 1 since round 5): prepared (fragment-order) filters of parameters come from
# torchseg_amd.shadow — ONE refresh launch per optimizer step instead of one preparation launch per convolution and direction
_SHADOW_PREP = os.environ.get("TSG_WEIGHT_SHADOW", "1") != "0"


class HipKernels:
    """libtsg_hip.so kernels on torch's current HIP stream."""

    name = "hip"

    def __init__(self):
        self.lib = L.lib()
        self._npart = {}

    def _count(self, key, fn, what):
        """A (cached) geometry query of the library whose non-negative result is a count."""
        v = self._npart.get(key)
        if v is None:
            v = fn()
            if v <= 0:
                L.check(v if v < 0 else -2, what)
            self._npart[key] = v
        return v

    def _num_partials(self, layout, N, Cc, HW):
        key = (layout, N, Cc, HW)
        v = self._npart.get(key)
        if v is None:
            v = self._npart[key] = self.lib.tsg_bn_num_partials(layout, N, Cc, HW)
        return v

    # ---- SyncBN -----------------------------------------------------------
    def bn_stats(self, x, layout, N, Cc, HW):
        """-> (partial fp32 [S,2,C], S)"""
        lib = self.lib
        smax = self._num_partials(layout, N, Cc, HW)
        partial = torch.empty((smax, 2, Cc), dtype=torch.float32, device=x.device)
        rows = C.c_int(0)
        L.check(lib.tsg_bn_stats(x.data_ptr(), L.dtype_code(x), layout, N, Cc, HW,
                                 partial.data_ptr(), C.byref(rows), L.stream_ptr(x)), "tsg_bn_stats")
        return partial, rows.value

    def bn_collapse(self, partial, S, Cc, out, count=None):
        """out[0:2C] = per-channel sums; with `count` also out[2C:2C+2] = the element count as two exact fp32 words"""
        if count is None:
            L.check(self.lib.tsg_bn_collapse(partial.data_ptr(), S, Cc, out.data_ptr(),
                                             L.stream_ptr(partial)), "tsg_bn_collapse")
        else:
            L.check(self.lib.tsg_bn_collapse_count(partial.data_ptr(), S, Cc, out.data_ptr(), int(count),
                                                   L.stream_ptr(partial)), "tsg_bn_collapse_count")

    def bn_finalize(self, partial, S, Cc, count, count_dev, eps, momentum, gamma, beta,
                    running_mean, running_var, nbt):
        """-> (mean[C], invstd[C], fwd_pack[3,C])"""
        dev = partial.device
        mean = torch.empty(Cc, dtype=torch.float32, device=dev)
        invstd = torch.empty(Cc, dtype=torch.float32, device=dev)
        fp = torch.empty((3, Cc), dtype=torch.float32, device=dev)
        L.check(self.lib.tsg_bn_finalize(partial.data_ptr(), S, Cc, float(count), L.ptr(count_dev),
                                         eps, momentum, L.ptr(gamma), L.ptr(beta), L.ptr(running_mean),
                                         L.ptr(running_var), L.ptr(nbt), mean.data_ptr(),
                                         invstd.data_ptr(), fp.data_ptr(), L.stream_ptr(partial)),
                "tsg_bn_finalize")
        return mean, invstd, fp

    def bn_affine(self, mean, invstd, gamma, beta):
        Cc = mean.numel()
        fp = torch.empty((3, Cc), dtype=torch.float32, device=mean.device)
        L.check(self.lib.tsg_bn_affine(mean.data_ptr(), invstd.data_ptr(), L.ptr(gamma), L.ptr(beta), Cc,
                                       fp.data_ptr(), L.stream_ptr(mean)), "tsg_bn_affine")
        return fp

    def bn_apply_fwd(self, x, residual, layout, N, Cc, HW, fp, relu, out=None):
        y = torch.empty_like(x) if out is None else out
        L.check(self.lib.tsg_bn_apply_fwd(x.data_ptr(), L.ptr(residual), y.data_ptr(),
                                          L.dtype_code(x), layout, N, Cc, HW, fp.data_ptr(),
                                          int(relu), L.stream_ptr(x)), "tsg_bn_apply_fwd")
        return y

    def bn_bwd_reduce(self, dy, x, y, layout, N, Cc, HW, fp, relu):
        """-> (partial fp32 [S,2,C] = {sum dy', sum dy'(x-mean)}, S)"""
        lib = self.lib
        smax = self._num_partials(layout, N, Cc, HW)
        partial = torch.empty((smax, 2, Cc), dtype=torch.float32, device=x.device)
        rows = C.c_int(0)
        L.check(lib.tsg_bn_bwd_reduce(dy.data_ptr(), x.data_ptr(), L.ptr(y), L.dtype_code(x),
                                      layout, N, Cc, HW, fp.data_ptr(), int(relu), partial.data_ptr(),
                                      C.byref(rows), L.stream_ptr(x)), "tsg_bn_bwd_reduce")
        return partial, rows.value

    def bn_bwd_coeffs(self, partial, S, Cc, count, count_dev, batch_stats, invstd, fp,
                      want_param_grads, want_pack):
        """-> (dgamma, dbeta, bwd_pack[5,C]) (None where not requested)"""
        dev = partial.device
        dgamma = torch.empty(Cc, dtype=torch.float32, device=dev) if want_param_grads else None
        dbeta = torch.empty(Cc, dtype=torch.float32, device=dev) if want_param_grads else None
        bp = torch.empty((5, Cc), dtype=torch.float32, device=dev) if want_pack else None
        L.check(self.lib.tsg_bn_bwd_coeffs(partial.data_ptr(), S, Cc, float(count), L.ptr(count_dev),
                                           int(batch_stats), invstd.data_ptr(), L.ptr(fp), L.ptr(dgamma),
                                           L.ptr(dbeta), L.ptr(bp), L.stream_ptr(partial)),
                "tsg_bn_bwd_coeffs")
        return dgamma, dbeta, bp

    def bn_bwd_apply(self, dy, x, y, layout, N, Cc, HW, bp, relu, want_dres):
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if want_dres else None
        L.check(self.lib.tsg_bn_bwd_apply(dy.data_ptr(), x.data_ptr(), L.ptr(y), dx.data_ptr(),
                                          L.ptr(dres), L.dtype_code(x), layout, N, Cc, HW,
                                          bp.data_ptr(), int(relu), L.stream_ptr(x)), "tsg_bn_bwd_apply")
        return dx, dres

    # ---- SyncBN block tail with the ReLU mask as one bit per element ------------
    def bn_maskbits_supported(self, x, layout, Cc, HW):
        return bool(self.lib.tsg_bn_maskbits_supported(L.dtype_code(x), layout, Cc, HW)) and x.data_ptr() % 16 == 0

    def bn_apply_fwd_bits(self, x, residual, layout, N, Cc, HW, fp):
        """relu(a x + b [+ residual]) -> (y, bits uint8 [N * HW * C / V]): bit j of byte (pixel * C + c) / V = (y > 0)"""
        y = torch.empty_like(x)
        V = 8 if x.dtype == torch.bfloat16 else 4
        bits = torch.empty(N * HW * Cc // V, dtype=torch.uint8, device=x.device)
        L.check(self.lib.tsg_bn_apply_fwd_maskbits(x.data_ptr(), L.ptr(residual), y.data_ptr(), bits.data_ptr(),
                                                   L.dtype_code(x), layout, N, Cc, HW, fp.data_ptr(), L.stream_ptr(x)),
                "tsg_bn_apply_fwd_maskbits")
        return y, bits

    def bn_bwd_reduce_bits(self, dy, x, bits, layout, N, Cc, HW, fp):
        smax = self._num_partials(layout, N, Cc, HW)
        partial = torch.empty((smax, 2, Cc), dtype=torch.float32, device=x.device)
        rows = C.c_int(0)
        L.check(self.lib.tsg_bn_bwd_reduce_maskbits(dy.data_ptr(), x.data_ptr(), bits.data_ptr(), L.dtype_code(x), layout, N,
                                                    Cc, HW, fp.data_ptr(), partial.data_ptr(), C.byref(rows),
                                                    L.stream_ptr(x)), "tsg_bn_bwd_reduce_maskbits")
        return partial, rows.value

    def bn_bwd_apply_bits(self, dy, x, bits, layout, N, Cc, HW, bp, want_dres):
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if want_dres else None
        L.check(self.lib.tsg_bn_bwd_apply_maskbits(dy.data_ptr(), x.data_ptr(), bits.data_ptr(), dx.data_ptr(), L.ptr(dres),
                                                   L.dtype_code(x), layout, N, Cc, HW, bp.data_ptr(), L.stream_ptr(x)),
                "tsg_bn_bwd_apply_maskbits")
        return dx, dres

    # ---- SyncBN, mixed layout (x NCHW, y/dy channels_last) -------------------
    def bn_mixed_supported(self, x):
        """4-D NCHW-contiguous activation that the stem kernels can take."""
        if x.dim() != 4 or not x.is_contiguous():
            return False
        return bool(self.lib.tsg_bn_mixed_supported(L.dtype_code(x), x.shape[1], x.shape[2] * x.shape[3]))

    def bn_apply_fwd_mixed(self, x, N, Cc, HW, fp, relu):
        y = torch.empty_like(x, memory_format=torch.channels_last)
        L.check(self.lib.tsg_bn_apply_fwd_mixed(x.data_ptr(), y.data_ptr(), L.dtype_code(x), N, Cc, HW,
                                                fp.data_ptr(), int(relu), L.stream_ptr(x)),
                "tsg_bn_apply_fwd_mixed")
        return y

    def bn_bwd_reduce_mixed(self, dy, x, N, Cc, HW, fp, relu):
        smax = self.lib.tsg_bn_mixed_num_partials(N, Cc, HW)
        partial = torch.empty((smax, 2, Cc), dtype=torch.float32, device=x.device)
        rows = C.c_int(0)
        L.check(self.lib.tsg_bn_bwd_reduce_mixed(dy.data_ptr(), x.data_ptr(), L.dtype_code(x), N, Cc, HW,
                                                 fp.data_ptr(), int(relu), partial.data_ptr(),
                                                 C.byref(rows), L.stream_ptr(x)), "tsg_bn_bwd_reduce_mixed")
        return partial, rows.value

    def bn_bwd_apply_mixed(self, dy, x, N, Cc, HW, bp, relu):
        dx = torch.empty_like(x)
        L.check(self.lib.tsg_bn_bwd_apply_mixed(dy.data_ptr(), x.data_ptr(), dx.data_ptr(), L.dtype_code(x),
                                                N, Cc, HW, bp.data_ptr(), int(relu), L.stream_ptr(x)),
                "tsg_bn_bwd_apply_mixed")
        return dx

    # ---- global average pool -------------------------------------------------
    def gap_fwd(self, x, layout, N, Cc, HW):
        wsb = self.lib.tsg_gap_ws_bytes(layout, N, Cc, HW)
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=x.device)
        out = torch.empty((N, Cc), dtype=x.dtype, device=x.device)
        L.check(self.lib.tsg_gap_fwd(x.data_ptr(), out.data_ptr(), L.dtype_code(x), layout, N, Cc, HW,
                                     ws.data_ptr(), ws.numel(), L.stream_ptr(x)), "tsg_gap_fwd")
        return out

    def gap_bwd(self, dout, like, layout, N, Cc, HW):
        dx = torch.empty_like(like)
        L.check(self.lib.tsg_gap_bwd(dout.data_ptr(), dx.data_ptr(), L.dtype_code(dout), layout, N, Cc, HW,
                                     L.stream_ptr(dout)), "tsg_gap_bwd")
        return dx

    def adaptive_avgpool_supported(self, x, OH, OW):
        """channels_last-dense [N,C,H,W] (f32 / bf16, C a multiple of the 16-byte vector) pooled to OH <= H, OW <= W"""
        return (x.dim() == 4 and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16)
                and x.is_contiguous(memory_format=torch.channels_last)
                and x.shape[1] % (8 if x.dtype == torch.bfloat16 else 4) == 0
                and 0 < OH <= x.shape[2] and 0 < OW <= x.shape[3])

    def adaptive_avgpool_fwd(self, x, OH, OW):
        """x [N,C,H,W] channels_last -> [N,C,OH,OW] channels_last (nn.AdaptiveAvgPool2d)"""
        N, Cc, H, W = x.shape
        wsb = self.lib.tsg_adaptive_avgpool_nhwc_ws_bytes(L.dtype_code(x), N, Cc, H, W, OH, OW)
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=x.device)
        out = torch.empty((N, Cc, OH, OW), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        L.check(self.lib.tsg_adaptive_avgpool_nhwc_fwd(x.data_ptr(), out.data_ptr(), L.dtype_code(x), N, Cc, H, W, OH, OW,
                                                       ws.data_ptr(), ws.numel(), L.stream_ptr(x)),
                "tsg_adaptive_avgpool_nhwc_fwd")
        return out

    def adaptive_avgpool_bwd(self, dout, H, W):
        """dout [N,C,OH,OW] channels_last -> dx [N,C,H,W] channels_last"""
        N, Cc, OH, OW = dout.shape
        dx = torch.empty((N, Cc, H, W), dtype=dout.dtype, device=dout.device, memory_format=torch.channels_last)
        L.check(self.lib.tsg_adaptive_avgpool_nhwc_bwd(dout.data_ptr(), dx.data_ptr(), L.dtype_code(dout), N, Cc, H, W, OH,
                                                       OW, L.stream_ptr(dout)), "tsg_adaptive_avgpool_nhwc_bwd")
        return dx

    def cat_channels(self, a, b):
        """a [B,Ca,H,W], b [B,Cb,H,W], same dtype, channels_last -> [B,Ca+Cb,H,W] channels_last"""
        B, Ca, H, W = a.shape
        Cb = b.shape[1]
        out = torch.empty((B, Ca + Cb, H, W), dtype=a.dtype, device=a.device, memory_format=torch.channels_last)
        es = a.element_size()
        L.check(self.lib.tsg_cat2_rows(a.data_ptr(), b.data_ptr(), out.data_ptr(), B * H * W, Ca * es, Cb * es, L.stream_ptr(a)),
                "tsg_cat2_rows")
        return out

    def chanscale_fwd(self, x, s, layout, N, Cc, HW, add_identity):
        y = torch.empty_like(x)
        L.check(self.lib.tsg_chanscale_fwd(x.data_ptr(), s.data_ptr(), y.data_ptr(), L.dtype_code(x), layout,
                                           N, Cc, HW, int(add_identity), L.stream_ptr(x)), "tsg_chanscale_fwd")
        return y

    def chanscale_bwd(self, dy, x, s, layout, N, Cc, HW, add_identity):
        wsb = self.lib.tsg_gap_ws_bytes(layout, N, Cc, HW)
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=x.device)
        dx = torch.empty_like(x)
        ds = torch.empty((N, Cc), dtype=x.dtype, device=x.device)
        L.check(self.lib.tsg_chanscale_bwd(dy.data_ptr(), x.data_ptr(), s.data_ptr(), dx.data_ptr(), ds.data_ptr(),
                                           L.dtype_code(x), layout, N, Cc, HW, int(add_identity), ws.data_ptr(),
                                           ws.numel(), L.stream_ptr(x)), "tsg_chanscale_bwd")
        return dx, ds

    def chanscale_split_supported(self, x, layout, Cc):
        """the two-phase backward of a gate (chanscale_bwd_ds / chanscale_bwd_dx): NHWC, whole 16-byte channel vectors"""
        return layout == L.NHWC and Cc % (8 if x.dtype == torch.bfloat16 else 4) == 0 and x.data_ptr() % 16 == 0

    def chanscale_bwd_ds(self, dy, x, layout, N, Cc, HW):
        """ds [N, C] = sum over pixels of dy x (the gate's gradient) without writing dx"""
        wsb = self.lib.tsg_gap_ws_bytes(layout, N, Cc, HW)
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=x.device)
        ds = torch.empty((N, Cc), dtype=x.dtype, device=x.device)
        L.check(self.lib.tsg_chanscale_bwd_ds(dy.data_ptr(), x.data_ptr(), ds.data_ptr(), L.dtype_code(x), layout, N, Cc, HW,
                                              ws.data_ptr(), ws.numel(), L.stream_ptr(x)), "tsg_chanscale_bwd_ds")
        return ds

    def chanscale_bwd_dx(self, dy, s, gadd, layout, N, Cc, HW, add_identity):
        """dx = dy s (+ dy) + gadd[n, c] (gadd: [N, C] of dy's dtype)"""
        dx = torch.empty_like(dy)
        L.check(self.lib.tsg_chanscale_bwd_dx(dy.data_ptr(), s.data_ptr(), gadd.data_ptr(), dx.data_ptr(), L.dtype_code(dy), layout,
                                              N, Cc, HW, int(add_identity), L.stream_ptr(dy)), "tsg_chanscale_bwd_dx")
        return dx

    # ---- max pool (channels_last) ------------------------------------------------
    def maxpool_fwd(self, x, K_, S_, P_):
        """x channels_last-dense [N,C,IH,IW] -> (y channels_last, argmax uint8 [N,OH,OW,C])"""
        N, Cc, IH, IW = x.shape
        OH, OW = (IH + 2 * P_ - K_) // S_ + 1, (IW + 2 * P_ - K_) // S_ + 1
        y = torch.empty((N, Cc, OH, OW), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        idx = torch.empty((N, OH, OW, Cc), dtype=torch.uint8, device=x.device)
        L.check(self.lib.tsg_maxpool_nhwc_fwd(x.data_ptr(), y.data_ptr(), idx.data_ptr(), L.dtype_code(x), N, Cc,
                                              IH, IW, OH, OW, K_, S_, P_, L.stream_ptr(x)), "tsg_maxpool_nhwc_fwd")
        return y, idx

    def maxpool_bwd(self, dy, idx, in_shape, K_, S_, P_):
        N, Cc, IH, IW = in_shape
        OH, OW = dy.shape[2], dy.shape[3]
        dx = torch.empty(in_shape, dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
        L.check(self.lib.tsg_maxpool_nhwc_bwd(dy.data_ptr(), idx.data_ptr(), dx.data_ptr(), L.dtype_code(dy), N, Cc,
                                              IH, IW, OH, OW, K_, S_, P_, L.stream_ptr(dy)), "tsg_maxpool_nhwc_bwd")
        return dx

    # ---- BN + ReLU + MaxPool2d(3, 2, 1) of the ResNet stem, fused ------------------
    def bn_relu_pool_fwd(self, x, fp):
        """x channels_last-dense [N,C,IH,IW], fp = forward pack -> (y channels_last [N,C,OH,OW], argmax uint8 [N,OH,OW,C])"""
        N, Cc, IH, IW = x.shape
        OH, OW = (IH - 1) // 2 + 1, (IW - 1) // 2 + 1
        y = torch.empty((N, Cc, OH, OW), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        idx = torch.empty((N, OH, OW, Cc), dtype=torch.uint8, device=x.device)
        L.check(self.lib.tsg_bn_relu_pool_fwd(x.data_ptr(), y.data_ptr(), idx.data_ptr(), L.dtype_code(x), N, Cc, IH, IW,
                                              OH, OW, fp.data_ptr(), L.stream_ptr(x)), "tsg_bn_relu_pool_fwd")
        return y, idx

    def bn_relu_pool_bwd_reduce(self, dpool, idx, x, fp):
        """-> (partial fp32 [S,2,C] = {sum dy', sum dy'(x-mean)}, S) with dy' gathered from dpool through idx"""
        N, Cc, IH, IW = x.shape
        OH, OW = dpool.shape[2], dpool.shape[3]
        dt = L.dtype_code(x)
        S = self._count(("bn_pool", dt, N, Cc, IH, IW),
                        lambda: self.lib.tsg_bn_relu_pool_bwd_num_partials(dt, N, Cc, IH, IW),
                        "tsg_bn_relu_pool_bwd_num_partials")
        partial = torch.empty((S, 2, Cc), dtype=torch.float32, device=x.device)
        L.check(self.lib.tsg_bn_relu_pool_bwd_reduce(dpool.data_ptr(), idx.data_ptr(), x.data_ptr(), L.dtype_code(x), N, Cc,
                                                     IH, IW, OH, OW, fp.data_ptr(), partial.data_ptr(), L.stream_ptr(x)),
                "tsg_bn_relu_pool_bwd_reduce")
        return partial, S

    def bn_relu_pool_bwd_apply(self, dpool, idx, x, bp):
        N, Cc, IH, IW = x.shape
        OH, OW = dpool.shape[2], dpool.shape[3]
        dx = torch.empty_like(x)
        L.check(self.lib.tsg_bn_relu_pool_bwd_apply(dpool.data_ptr(), idx.data_ptr(), x.data_ptr(), dx.data_ptr(),
                                                    L.dtype_code(x), N, Cc, IH, IW, OH, OW, bp.data_ptr(), L.stream_ptr(x)),
                "tsg_bn_relu_pool_bwd_apply")
        return dx

    # ---- stem convolution ----------------------------------------------------
    def stem_conv_supported(self, x, weight, stride, padding, dilation, groups):
        if x.dim() != 4 or weight.dim() != 4 or x.dtype != torch.bfloat16 or not x.is_contiguous():
            return False
        return bool(self.lib.tsg_stem_conv_supported(L.dtype_code(x), x.shape[1], weight.shape[0], weight.shape[2],
                                                     weight.shape[3], stride, padding, dilation, groups,
                                                     x.shape[2], x.shape[3]))

    def _scratch(self, name, nbytes, dev):
        """A scratch buffer per (kernel family, device, STREAM), grown on demand.  Per stream because launches on different
        streams may overlap (the weight gradients of one backward pass can be split between the side stream of
        convwrw.wrw_on_side_stream and the compute stream its fallbacks use; one shared buffer would be written by two
        kernels at once).  Launches of one stream run in order, so they share."""
        st = torch.cuda.current_stream(dev)
        key = (name, dev.index, st.cuda_stream)
        pool = self.__dict__.setdefault("_scratch_bufs", {})
        ws = pool.get(key)
        if ws is None or ws.numel() < nbytes:
            if ws is not None:
                # a captured hipGraph may have recorded the predecessor's address (round 6: a graph captured on a side stream
                # whose scratch a LATER capture on that stream outgrew replayed into freed memory): predecessors are kept
                self.__dict__.setdefault("_scratch_retired", []).append(ws)
            ws = pool[key] = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
            ws.record_stream(st)                  # allocated under whatever stream torch considers current: a grown
            #                                       buffer's predecessor must outlive the kernel still using it
        return ws

    def presize_scratch(self, streams, dev):
        """Give every stream in `streams` scratch buffers as large as the largest any stream has per kernel family, so that no
        buffer grows (= moves) between two graph captures on one stream (bench.SegmentedStep)."""
        pool = self.__dict__.setdefault("_scratch_bufs", {})
        need = {}
        for (name, di, _sid), ws in list(pool.items()):
            if di == dev.index:
                need[name] = max(need.get(name, 0), ws.numel())
        for st in streams:
            with torch.cuda.stream(st):
                for name, n in need.items():
                    self._scratch(name, n, dev)

    def _stem_ws(self, dev):
        return self._scratch("stem", self.lib.tsg_stem_conv_ws_bytes(), dev)

    def stem_conv_fwd(self, x, weight):
        """x [B,3,H,W] bf16 contiguous, weight fp32 [64,3,7,7] -> y [B,64,OH,OW] bf16 channels_last"""
        _require_contiguous(x, weight)
        B, _, H, W = x.shape
        y = torch.empty((B, 64, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=x.dtype, device=x.device,
                        memory_format=torch.channels_last)
        ws = self._stem_ws(x.device)
        L.check(self.lib.tsg_stem_conv_fwd(x.data_ptr(), weight.data_ptr(), y.data_ptr(), B, H, W, ws.data_ptr(),
                                           ws.numel(), L.stream_ptr(x)), "tsg_stem_conv_fwd")
        return y

    def stem_conv_fwd_stats(self, x, weight):
        """stem_conv_fwd + the BatchNorm statistics of its output: -> (y, partial fp32 [S,2,64] = {sum y, sum y^2})"""
        _require_contiguous(x, weight)
        B, _, H, W = x.shape
        y = torch.empty((B, 64, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=x.dtype, device=x.device,
                        memory_format=torch.channels_last)
        S = self._count(("stem_stats", B, H, W), lambda: self.lib.tsg_stem_conv_stats_partials(B, H, W),
                        "tsg_stem_conv_stats_partials")
        partial = torch.empty((S, 2, 64), dtype=torch.float32, device=x.device)
        ws = self._stem_ws(x.device)
        L.check(self.lib.tsg_stem_conv_fwd_stats(x.data_ptr(), weight.data_ptr(), y.data_ptr(), partial.data_ptr(), B, H,
                                                 W, ws.data_ptr(), ws.numel(), L.stream_ptr(x)), "tsg_stem_conv_fwd_stats")
        return y, partial

    def stem_conv_wrw(self, x, dy):
        """x as in stem_conv_fwd, dy [B,64,OH,OW] bf16 channels_last -> dw fp32 [64,3,7,7]"""
        _require_contiguous(x)
        if not dy.is_contiguous(memory_format=torch.channels_last):
            raise ValueError("stem_conv_wrw expects a channels_last gradient")
        B, _, H, W = x.shape
        dw = torch.empty((64, 3, 7, 7), dtype=torch.float32, device=x.device)
        ws = self._stem_ws(x.device)
        L.check(self.lib.tsg_stem_conv_wrw(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, H, W, ws.data_ptr(),
                                           ws.numel(), L.stream_ptr(x)), "tsg_stem_conv_wrw")
        return dw

    def stem_conv_wrw_bn(self, x, da, xc, bp):
        """stem_conv_wrw of dy = BN+ReLU backward(da; xc, bp) without writing dy: x the image (as stem_conv_fwd), da / xc
        [B,64,OH,OW] bf16 channels_last (gradient w.r.t. relu(bn(xc)) / the stem output), bp fp32 [5,64] backward pack"""
        _require_contiguous(x, bp)
        for t in (da, xc):
            if not t.is_contiguous(memory_format=torch.channels_last) or t.dtype != torch.bfloat16 or t.shape != da.shape:
                raise ValueError("stem_conv_wrw_bn expects bf16 channels_last da / xc of one shape")
        if bp.dtype != torch.float32 or tuple(bp.shape) != (5, 64):
            raise ValueError("stem_conv_wrw_bn expects the [5, 64] fp32 backward pack")
        B, _, H, W = x.shape
        dw = torch.empty((64, 3, 7, 7), dtype=torch.float32, device=x.device)
        ws = self._stem_ws(x.device)
        L.check(self.lib.tsg_stem_conv_wrw_bn(x.data_ptr(), da.data_ptr(), xc.data_ptr(), bp.data_ptr(), dw.data_ptr(), B, H,
                                              W, ws.data_ptr(), ws.numel(), L.stream_ptr(x)), "tsg_stem_conv_wrw_bn")
        return dw

    # ---- the recomputing ResNet stem (round 6): conv1 -> bn1 -> relu -> maxpool without the stem activation ----
    def stem_conv_stats(self, x, weight):
        """-> partial fp32 [S,2,64] = {sum y, sum y^2} of y = stem_conv_fwd(x, weight), y not written"""
        _require_contiguous(x, weight)
        B, _, H, W = x.shape
        S = self._count(("stem_stats", B, H, W), lambda: self.lib.tsg_stem_conv_stats_partials(B, H, W),
                        "tsg_stem_conv_stats_partials")
        partial = torch.empty((S, 2, 64), dtype=torch.float32, device=x.device)
        ws = self._stem_ws(x.device)
        L.check(self.lib.tsg_stem_conv_stats(x.data_ptr(), weight.data_ptr(), partial.data_ptr(), B, H, W, ws.data_ptr(),
                                             ws.numel(), L.stream_ptr(x)), "tsg_stem_conv_stats")
        return partial

    def stem_conv_bn_relu_pool_fwd(self, x, weight, fp):
        """maxpool_3x3/2/1(relu(bn(stem_conv(x)))) -> (ypool [B,64,PH,PW] bf16 channels_last, argmax uint8 [B,PH,PW,64])"""
        _require_contiguous(x, weight, fp)
        B, _, H, W = x.shape
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        PH, PW = (OH - 1) // 2 + 1, (OW - 1) // 2 + 1
        y = torch.empty((B, 64, PH, PW), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
        idx = torch.empty((B, PH, PW, 64), dtype=torch.uint8, device=x.device)
        ws = self._stem_ws(x.device)
        L.check(self.lib.tsg_stem_conv_bn_relu_pool_fwd(x.data_ptr(), weight.data_ptr(), fp.data_ptr(), y.data_ptr(),
                                                        idx.data_ptr(), B, H, W, ws.data_ptr(), ws.numel(), L.stream_ptr(x)),
                "tsg_stem_conv_bn_relu_pool_fwd")
        return y, idx

    def stem_conv_bn_relu_pool_bwd_reduce(self, x, weight, dpool, idx, fp):
        """-> (partial fp32 [S,2,64] = {sum dy', sum dy'(y-mean)}, S) with y recomputed and dy' gathered from dpool through idx"""
        _require_contiguous(x, weight, fp)
        if not dpool.is_contiguous(memory_format=torch.channels_last) or dpool.dtype != torch.bfloat16:
            raise ValueError("stem_conv_bn_relu_pool_bwd_reduce expects a bf16 channels_last pooled gradient")
        B, _, H, W = x.shape
        S = self._count(("stem_pool", B, H, W), lambda: self.lib.tsg_stem_pool_bwd_num_partials(B, H, W),
                        "tsg_stem_pool_bwd_num_partials")
        partial = torch.empty((S, 2, 64), dtype=torch.float32, device=x.device)
        ws = self._stem_ws(x.device)
        L.check(self.lib.tsg_stem_conv_bn_relu_pool_bwd_reduce(x.data_ptr(), weight.data_ptr(), dpool.data_ptr(), idx.data_ptr(),
                                                               fp.data_ptr(), partial.data_ptr(), B, H, W, ws.data_ptr(),
                                                               ws.numel(), L.stream_ptr(x)),
                "tsg_stem_conv_bn_relu_pool_bwd_reduce")
        return partial, S

    def stem_conv_wrw_bn_pool(self, x, weight, dpool, idx, bp, xc=None):
        """dw fp32 [64,3,7,7] of the stem with dy = BN+ReLU+pool backward(dpool; idx, y, bp) staged on the fly; y = xc (the
        stored stem output, bf16 channels_last) when given, re-evaluated from x otherwise"""
        _require_contiguous(x, weight, bp)
        if not dpool.is_contiguous(memory_format=torch.channels_last) or dpool.dtype != torch.bfloat16:
            raise ValueError("stem_conv_wrw_bn_pool expects a bf16 channels_last pooled gradient")
        if bp.dtype != torch.float32 or tuple(bp.shape) != (5, 64):
            raise ValueError("stem_conv_wrw_bn_pool expects the [5, 64] fp32 backward pack")
        B, _, H, W = x.shape
        if xc is not None and (xc.dtype != torch.bfloat16 or not xc.is_contiguous(memory_format=torch.channels_last)
                               or tuple(xc.shape) != (B, 64, (H - 1) // 2 + 1, (W - 1) // 2 + 1)):
            raise ValueError("stem_conv_wrw_bn_pool expects xc = the bf16 channels_last stem output")
        dw = torch.empty((64, 3, 7, 7), dtype=torch.float32, device=x.device)
        ws = self._stem_ws(x.device)
        L.check(self.lib.tsg_stem_conv_wrw_bn_pool(x.data_ptr(), weight.data_ptr(), xc.data_ptr() if xc is not None else None,
                                                   dpool.data_ptr(), idx.data_ptr(), bp.data_ptr(), dw.data_ptr(), B, H, W,
                                                   ws.data_ptr(), ws.numel(), L.stream_ptr(x)), "tsg_stem_conv_wrw_bn_pool")
        return dw

    # ---- OHEM / focal / upsample ------------------------------------------
    def ohem_fwd(self, logits, labels, ignore_label, thresh, min_kept, weight):
        """logits [B,C,H,W] contiguous, labels [B,H,W] -> (loss[1], nll[P], lse[P], sel[8] int32)"""
        _require_contiguous(logits, labels, weight)
        B, Cc = logits.shape[0], logits.shape[1]
        HW = logits.numel() // (B * Cc)
        P = B * HW
        dev = logits.device
        plan = L.OhemPlan()
        L.check(self.lib.tsg_ohem_make_plan(B, Cc, HW, float(thresh), C.byref(plan)), "tsg_ohem_make_plan")
        ws = torch.empty(plan.ws_bytes, dtype=torch.uint8, device=dev)
        nll = torch.empty(P, dtype=torch.float32, device=dev)
        lse = torch.empty(P, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        sel = torch.empty(8, dtype=torch.int32, device=dev)
        L.check(self.lib.tsg_ohem_fwd(logits.data_ptr(), L.dtype_code(logits), labels.data_ptr(),
                                      _label_code(labels), B, Cc, HW, int(ignore_label), float(thresh),
                                      int(min_kept), L.ptr(weight), nll.data_ptr(), lse.data_ptr(),
                                      loss.data_ptr(), sel.data_ptr(), ws.data_ptr(), plan.ws_bytes,
                                      L.stream_ptr(logits)), "tsg_ohem_fwd")
        return loss, nll, lse, sel

    def ohem_bwd(self, logits, labels, ignore_label, weight, nll, lse, sel, gscale):
        _require_contiguous(logits, labels, weight)
        B, Cc = logits.shape[0], logits.shape[1]
        HW = logits.numel() // (B * Cc)
        dlogits = torch.empty_like(logits)
        L.check(self.lib.tsg_ohem_bwd(logits.data_ptr(), L.dtype_code(logits), labels.data_ptr(),
                                      _label_code(labels), B, Cc, HW, int(ignore_label), L.ptr(weight),
                                      nll.data_ptr(), lse.data_ptr(), sel.data_ptr(), gscale.data_ptr(),
                                      dlogits.data_ptr(), None, L.stream_ptr(logits)), "tsg_ohem_bwd")
        return dlogits

    def ohem_up_supported(self, z, OH, OW, thresh):
        return bool(self.lib.tsg_ohem_up_supported(z.shape[1], z.shape[2], z.shape[3], int(OH), int(OW), float(thresh)))

    def ohem_up_fwd(self, z, labels, OH, OW, ignore_label, thresh, min_kept, weight):
        """z [B,C,IH,IW] contiguous low-res logits, labels [B,OH,OW] -> as ohem_fwd"""
        _require_contiguous(z, labels, weight)
        B, Cc, IH, IW = z.shape
        P = B * OH * OW
        dev = z.device
        plan = L.OhemPlan()
        L.check(self.lib.tsg_ohem_make_plan(B, Cc, OH * OW, float(thresh), C.byref(plan)), "tsg_ohem_make_plan")
        ws = torch.empty(plan.ws_bytes, dtype=torch.uint8, device=dev)
        nll = torch.empty(P, dtype=torch.float32, device=dev)
        lse = torch.empty(P, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        sel = torch.empty(8, dtype=torch.int32, device=dev)
        L.check(self.lib.tsg_ohem_up_fwd(z.data_ptr(), L.dtype_code(z), labels.data_ptr(), _label_code(labels),
                                         B, Cc, IH, IW, int(OH), int(OW), int(ignore_label), float(thresh),
                                         int(min_kept), L.ptr(weight), nll.data_ptr(), lse.data_ptr(),
                                         loss.data_ptr(), sel.data_ptr(), ws.data_ptr(), plan.ws_bytes,
                                         L.stream_ptr(z)), "tsg_ohem_up_fwd")
        return loss, nll, lse, sel

    def ohem_up_bwd(self, z, labels, OH, OW, ignore_label, weight, nll, lse, sel, gscale):
        _require_contiguous(z, labels, weight)
        B, Cc, IH, IW = z.shape
        dz = torch.empty_like(z)
        wsb = self.lib.tsg_ohem_up_bwd_ws_bytes(B, Cc, IH, int(OW))
        ws = torch.empty(wsb, dtype=torch.uint8, device=z.device)
        L.check(self.lib.tsg_ohem_up_bwd(z.data_ptr(), L.dtype_code(z), labels.data_ptr(), _label_code(labels),
                                         B, Cc, IH, IW, int(OH), int(OW), int(ignore_label), L.ptr(weight),
                                         nll.data_ptr(), lse.data_ptr(), sel.data_ptr(), gscale.data_ptr(),
                                         dz.data_ptr(), ws.data_ptr(), wsb, L.stream_ptr(z)), "tsg_ohem_up_bwd")
        return dz

    def ohem_target_prob(self, nll, labels, C_, ignore_label):
        """mask_prob (loss_opr.py:81-83) from the forward's nll, with the kernels' own exp: fp32 [P]"""
        _require_contiguous(nll, labels)
        out = torch.empty_like(nll)
        L.check(self.lib.tsg_ohem_target_prob(nll.data_ptr(), labels.data_ptr(), _label_code(labels), nll.numel(),
                                              int(C_), int(ignore_label), out.data_ptr(), L.stream_ptr(nll)),
                "tsg_ohem_target_prob")
        return out

    def kth_value(self, v, k):
        n = v.numel()
        wsb = self.lib.tsg_kth_ws_bytes(n)
        ws = torch.empty(wsb, dtype=torch.uint8, device=v.device)
        out = torch.empty(1, dtype=torch.float32, device=v.device)
        L.check(self.lib.tsg_kth_value(v.data_ptr(), n, int(k), out.data_ptr(), ws.data_ptr(), wsb,
                                       L.stream_ptr(v)), "tsg_kth_value")
        return out

    def focal_fwd(self, pred, target, ignore_label, gamma, alpha):
        _require_contiguous(pred, target)
        P = pred.numel()
        wsb = self.lib.tsg_focal_ws_bytes(P)
        ws = torch.empty(wsb, dtype=torch.uint8, device=pred.device)
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        L.check(self.lib.tsg_focal_fwd(pred.data_ptr(), L.dtype_code(pred), target.data_ptr(),
                                       _label_code(target), P, int(ignore_label), float(gamma),
                                       float(alpha), loss.data_ptr(), ws.data_ptr(), wsb,
                                       L.stream_ptr(pred)), "tsg_focal_fwd")
        return loss

    def focal_bwd(self, pred, target, ignore_label, gamma, alpha, gscale):
        dpred = torch.empty_like(pred)
        L.check(self.lib.tsg_focal_bwd(pred.data_ptr(), L.dtype_code(pred), target.data_ptr(),
                                       _label_code(target), pred.numel(), int(ignore_label), float(gamma),
                                       float(alpha), gscale.data_ptr(), dpred.data_ptr(),
                                       L.stream_ptr(pred)), "tsg_focal_bwd")
        return dpred

    def upsample_fwd(self, x, add, OH, OW):
        """x [N,C,IH,IW] contiguous -> [N,C,OH,OW]"""
        _require_contiguous(x, add)
        N, Cc, IH, IW = x.shape
        y = torch.empty((N, Cc, OH, OW), dtype=x.dtype, device=x.device)
        L.check(self.lib.tsg_upsample_bilinear_ac_fwd(x.data_ptr(), L.ptr(add), y.data_ptr(),
                                                      L.dtype_code(x), N * Cc, IH, IW, OH, OW,
                                                      L.stream_ptr(x)), "tsg_upsample_bilinear_ac_fwd")
        return y

    def upsample_presum_fwd(self, x, x2, OH, OW):
        """up(x + x2) for two same-shaped tensors, both NCHW-contiguous or both channels_last-dense"""
        N, Cc, IH, IW = x.shape
        cl = torch.channels_last
        same_layout = (x.is_contiguous() and x2.is_contiguous()) or \
            (x.is_contiguous(memory_format=cl) and x2.is_contiguous(memory_format=cl))     # batch 1 / C 1: strides of
        if x2.shape != x.shape or x2.dtype != x.dtype or not same_layout:                   # size-1 dims are free
            raise L.TsgError("upsample_presum_fwd: the two addends must share shape, dtype and a dense layout")
        if x.is_contiguous():
            y = torch.empty((N, Cc, OH, OW), dtype=x.dtype, device=x.device)
            L.check(self.lib.tsg_upsample_bilinear_ac_presum_fwd(x.data_ptr(), x2.data_ptr(), y.data_ptr(),
                                                                 L.dtype_code(x), N * Cc, IH, IW, OH, OW,
                                                                 L.stream_ptr(x)), "tsg_upsample_bilinear_ac_presum_fwd")
            return y
        if not x.is_contiguous(memory_format=torch.channels_last):
            raise L.TsgError("upsample_presum_fwd: dense NCHW or channels_last tensors only")
        y = torch.empty((N, Cc, OH, OW), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        L.check(self.lib.tsg_upsample_bilinear_ac_nhwc_presum_fwd(x.data_ptr(), x2.data_ptr(), y.data_ptr(),
                                                                  L.dtype_code(x), N, Cc, IH, IW, OH, OW,
                                                                  L.stream_ptr(x)), "tsg_upsample_bilinear_ac_nhwc_presum_fwd")
        return y

    def upsample_bwd(self, dy, IH, IW):
        _require_contiguous(dy)
        N, Cc, OH, OW = dy.shape
        dx = torch.empty((N, Cc, IH, IW), dtype=dy.dtype, device=dy.device)
        L.check(self.lib.tsg_upsample_bilinear_ac_bwd(dy.data_ptr(), dx.data_ptr(), L.dtype_code(dy),
                                                      N * Cc, IH, IW, OH, OW, L.stream_ptr(dy)),
                "tsg_upsample_bilinear_ac_bwd")
        return dx

    def upsample_fwd_nhwc(self, x, add, OH, OW):
        """x [N,C,IH,IW] channels_last-dense -> [N,C,OH,OW] channels_last"""
        N, Cc, IH, IW = x.shape
        y = torch.empty((N, Cc, OH, OW), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        L.check(self.lib.tsg_upsample_bilinear_ac_nhwc_fwd(x.data_ptr(), L.ptr(add), y.data_ptr(),
                                                           L.dtype_code(x), N, Cc, IH, IW, OH, OW,
                                                           L.stream_ptr(x)), "tsg_upsample_bilinear_ac_nhwc_fwd")
        return y

    def upsample_bwd_nhwc(self, dy, IH, IW):
        N, Cc, OH, OW = dy.shape
        dx = torch.empty((N, Cc, IH, IW), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
        L.check(self.lib.tsg_upsample_bilinear_ac_nhwc_bwd(dy.data_ptr(), dx.data_ptr(), L.dtype_code(dy),
                                                           N, Cc, IH, IW, OH, OW, L.stream_ptr(dy)),
                "tsg_upsample_bilinear_ac_nhwc_bwd")
        return dx

    def upsample_nearest(self, x, OH, OW):
        shp = x.shape
        IH, IW = shp[-2], shp[-1]
        NC = x.numel() // (IH * IW)
        y = torch.empty(tuple(shp[:-2]) + (OH, OW), dtype=x.dtype, device=x.device)
        L.check(self.lib.tsg_upsample_nearest_fwd(x.data_ptr(), y.data_ptr(), x.element_size(), NC,
                                                  IH, IW, OH, OW, L.stream_ptr(x)),
                "tsg_upsample_nearest_fwd")
        return y


    # ---- PSA attention -------------------------------------------------------
    def psa_fwd(self, X, A):
        """X [B,Cx,K], A [B,K,N] contiguous, same dtype -> (out [B,Cx,N], lse fp32 [B,N])"""
        _require_contiguous(X, A)
        B, Cx, Kd = X.shape
        N = A.shape[2]
        dt = L.dtype_code(X)
        wsb = self.lib.tsg_psa_ws_bytes(dt, 0, B, Cx, Kd, N)
        ws = torch.empty(wsb, dtype=torch.uint8, device=X.device)
        out = torch.empty((B, Cx, N), dtype=X.dtype, device=X.device)
        lse = torch.empty((B, N), dtype=torch.float32, device=X.device)
        L.check(self.lib.tsg_psa_fwd(X.data_ptr(), A.data_ptr(), out.data_ptr(), lse.data_ptr(), dt, B, Cx,
                                     Kd, N, ws.data_ptr(), wsb, L.stream_ptr(X)), "tsg_psa_fwd")
        return out, lse

    def psa_bwd(self, X, A, out, dout, lse):
        _require_contiguous(X, A, out, dout)
        B, Cx, Kd = X.shape
        N = A.shape[2]
        dt = L.dtype_code(X)
        wsb = self.lib.tsg_psa_ws_bytes(dt, 1, B, Cx, Kd, N)
        ws = torch.empty(wsb, dtype=torch.uint8, device=X.device)
        dX = torch.empty_like(X)
        dA = torch.empty_like(A)
        L.check(self.lib.tsg_psa_bwd(X.data_ptr(), A.data_ptr(), out.data_ptr(), dout.data_ptr(),
                                     lse.data_ptr(), dX.data_ptr(), dA.data_ptr(), dt, B, Cx, Kd, N,
                                     ws.data_ptr(), wsb, L.stream_ptr(X)), "tsg_psa_bwd")
        return dX, dA

    # ---- classifier convolution of a head (csrc/clshead.hip) ---------------------
    # ---- 1x1 convolution of a globally pooled map (csrc/vecconv.hip)
    def conv1x1_vec_supported(self, x, weight):
        return (x.dim() == 4 and x.shape[2] == 1 and x.shape[3] == 1 and x.dtype == torch.bfloat16 and weight.dim() == 4
                and weight.dtype == torch.float32 and tuple(weight.shape[2:]) == (1, 1) and weight.shape[1] == x.shape[1]
                and bool(self.lib.tsg_conv1x1_vec_supported(x.shape[0], x.shape[1], weight.shape[0])))

    @staticmethod
    def _rows(t):
        """[N, C, 1, 1] as N rows of C contiguous elements: the tensor itself when its memory already is that (contiguous or
        channels_last strides, or a broadcast-free view), otherwise a copy"""
        if t.stride(1) == 1 and t.stride(0) == t.shape[1]:
            return t
        return t.reshape(t.shape[0], t.shape[1]).contiguous()

    def conv1x1_vec_fwd(self, x, weight):
        """x [B,Cin,1,1] bf16, weight fp32 [Cout,Cin,1,1] -> y [B,Cout,1,1] bf16"""
        B, Cin = x.shape[0], x.shape[1]
        Cout = weight.shape[0]
        x, w = self._rows(x), self._rows(weight)
        y = torch.empty((B, Cout, 1, 1), dtype=torch.bfloat16, device=x.device)
        L.check(self.lib.tsg_conv1x1_vec_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), B, Cin, Cout, L.stream_ptr(x)),
                "tsg_conv1x1_vec_fwd")
        return y

    def conv1x1_vec_bnact_fwd(self, x, weight, bnmode, act, gamma=None, beta=None, running_mean=None, running_var=None,
                              num_batches_tracked=None, eps=1e-5, momentum=0.1):
        """The pooled layer conv1x1 -> [BatchNorm over the batch] -> [ReLU | sigmoid] in one launch: x [B,Cin,1,1] bf16 ->
        (out [B,Cout,1,1] bf16, yc = the convolution's output [B,Cout] bf16 or None, stats fp32 [4,Cout] or None)"""
        B, Cin = x.shape[0], x.shape[1]
        Cout = weight.shape[0]
        x, w = self._rows(x), self._rows(weight)
        out = torch.empty((B, Cout, 1, 1), dtype=torch.bfloat16, device=x.device)
        yc = torch.empty((B, Cout), dtype=torch.bfloat16, device=x.device) if bnmode else None
        stats = torch.empty((4, Cout), dtype=torch.float32, device=x.device) if bnmode else None
        L.check(self.lib.tsg_conv1x1_vec_bnact_fwd(x.data_ptr(), w.data_ptr(), out.data_ptr(), L.ptr(yc), L.ptr(stats),
                                                   L.ptr(gamma), L.ptr(beta), L.ptr(running_mean), L.ptr(running_var),
                                                   L.ptr(num_batches_tracked), float(eps), float(momentum), int(bnmode),
                                                   int(act), B, Cin, Cout, L.stream_ptr(x)), "tsg_conv1x1_vec_bnact_fwd")
        return out, yc, stats

    def conv1x1_vec_bnact_bwd(self, dout, out, yc, stats, x, weight, bnmode, act, need_dx=True):
        """-> (dx bf16 [B,Cin,1,1] or None, dw fp32 like weight, dgamma, dbeta fp32 [Cout] or None)"""
        B, Cin = x.shape[0], x.shape[1]
        Cout = weight.shape[0]
        dout, out, x, w = self._rows(dout), self._rows(out), self._rows(x), self._rows(weight)
        dx = torch.empty((B, Cin, 1, 1), dtype=torch.bfloat16, device=x.device) if need_dx else None
        dw = torch.empty((Cout, Cin, 1, 1), dtype=torch.float32, device=x.device)
        dgamma = torch.empty(Cout, dtype=torch.float32, device=x.device) if bnmode else None
        dbeta = torch.empty(Cout, dtype=torch.float32, device=x.device) if bnmode else None
        L.check(self.lib.tsg_conv1x1_vec_bnact_bwd(dout.data_ptr(), out.data_ptr(), L.ptr(yc), L.ptr(stats), x.data_ptr(),
                                                   w.data_ptr(), L.ptr(dx), dw.data_ptr(), L.ptr(dgamma), L.ptr(dbeta),
                                                   int(bnmode), int(act), B, Cin, Cout, L.stream_ptr(x)),
                "tsg_conv1x1_vec_bnact_bwd")
        return dx, dw, dgamma, dbeta

    def conv1x1_vec_bwd(self, dy, x, weight, need_dx=True):
        """dy [B,Cout,1,1] bf16, x [B,Cin,1,1] bf16, weight fp32 -> (dx bf16 [B,Cin,1,1] or None, dw fp32 like weight)"""
        B, Cin = x.shape[0], x.shape[1]
        Cout = weight.shape[0]
        dy, x, w = self._rows(dy), self._rows(x), self._rows(weight)
        dx = torch.empty((B, Cin, 1, 1), dtype=torch.bfloat16, device=x.device) if need_dx else None
        dw = torch.empty((Cout, Cin, 1, 1), dtype=torch.float32, device=x.device)
        L.check(self.lib.tsg_conv1x1_vec_bwd(dy.data_ptr(), x.data_ptr(), w.data_ptr(), L.ptr(dx), dw.data_ptr(), B, Cin, Cout,
                                             L.stream_ptr(x)), "tsg_conv1x1_vec_bwd")
        return dx, dw

    def cls_head_supported(self, x, weight):
        return (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and weight.dim() == 4 and weight.shape[2:] == (1, 1)
                and weight.dtype == torch.float32 and x.shape[1] == weight.shape[1]
                and bool(self.lib.tsg_cls_head_supported(L.BF16, x.shape[1], weight.shape[0], x.shape[2] * x.shape[3])))

    def cls_head_fwd(self, x, weight, bias):
        """x bf16 channels_last [B,C,H,W], weight fp32 [N,C,1,1] -> z bf16 NCHW-contiguous [B,N,H,W] (planar logits)"""
        if not x.is_contiguous(memory_format=torch.channels_last):
            raise ValueError("cls_head_fwd expects a channels_last activation")
        _require_contiguous(weight, bias)
        B, Cc, H, W = x.shape
        N = weight.shape[0]
        z = torch.empty((B, N, H, W), dtype=torch.bfloat16, device=x.device)
        L.check(self.lib.tsg_cls_head_fwd(x.data_ptr(), weight.data_ptr(), L.ptr(bias), z.data_ptr(), B, H * W, Cc, N,
                                          L.stream_ptr(x)), "tsg_cls_head_fwd")
        return z

    def cls_head_bwd(self, dz, x, weight, need_dx=True, need_db=True):
        """dz bf16 NCHW-contiguous [B,N,H,W] -> (dx bf16 channels_last like x | None, dw fp32 like weight, dbias fp32 [N] | None)"""
        _require_contiguous(dz, weight)
        B, Cc, H, W = x.shape
        N = weight.shape[0]
        dx = None
        if need_dx:
            dx = torch.empty_like(x)
            L.check(self.lib.tsg_cls_head_dgrad(dz.data_ptr(), weight.data_ptr(), dx.data_ptr(), B, H * W, Cc, N,
                                                L.stream_ptr(dz)), "tsg_cls_head_dgrad")
        dw = torch.empty_like(weight)
        db = torch.empty(N, dtype=torch.float32, device=x.device) if need_db else None
        wsb = self.lib.tsg_cls_head_wgrad_ws_bytes(B, Cc, N)
        ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
        L.check(self.lib.tsg_cls_head_wgrad(dz.data_ptr(), x.data_ptr(), dw.data_ptr(), L.ptr(db), B, H * W, Cc, N,
                                            ws.data_ptr(), wsb, L.stream_ptr(dz)), "tsg_cls_head_wgrad")
        return dx, dw, db

    # ---- reference-accuracy fp32 convolution (parity path; csrc/convf32.hip) ------
    @staticmethod
    def _strides4(t):
        return (C.c_int64 * 4)(*[int(v) for v in t.stride()])

    def conv2d_f32_exact_fwd(self, x, w, stride, padding, dilation):
        """y = conv2d(x, w) with exact products and fp64 accumulation; x [B,Cin,H,W] / w [Cout,Cin,KH,KW] fp32 in any
        (dense, non-overlapping) layout; y takes x's memory format."""
        B, Cin, H, W = x.shape
        Cout, _, KH, KW = w.shape
        OH = (H + 2 * padding[0] - dilation[0] * (KH - 1) - 1) // stride[0] + 1
        OW = (W + 2 * padding[1] - dilation[1] * (KW - 1) - 1) // stride[1] + 1
        cl = x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
        y = torch.empty((B, Cout, OH, OW), dtype=torch.float32, device=x.device,
                        memory_format=torch.channels_last if cl else torch.contiguous_format)
        L.check(self.lib.tsg_conv2d_f32_exact_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), B, Cin, H, W, Cout, KH, KW,
                                                  stride[0], stride[1], padding[0], padding[1], dilation[0], dilation[1],
                                                  self._strides4(x), self._strides4(w), self._strides4(y),
                                                  L.stream_ptr(x)), "tsg_conv2d_f32_exact_fwd")
        return y

    def conv2d_f32_exact_dgrad(self, dy, w, x_like, stride, padding, dilation):
        B, Cin, H, W = x_like.shape
        Cout, _, KH, KW = w.shape
        dx = torch.empty_like(x_like)
        L.check(self.lib.tsg_conv2d_f32_exact_dgrad(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), B, Cin, H, W, Cout, KH, KW,
                                                    stride[0], stride[1], padding[0], padding[1], dilation[0], dilation[1],
                                                    self._strides4(dx), self._strides4(w), self._strides4(dy),
                                                    L.stream_ptr(dy)), "tsg_conv2d_f32_exact_dgrad")
        return dx

    def conv2d_f32_exact_wgrad(self, x, dy, w_like, stride, padding, dilation):
        B, Cin, H, W = x.shape
        Cout, _, KH, KW = w_like.shape
        dw = torch.empty_like(w_like)
        geo = (Cin, H, W, Cout, KH, KW, stride[0], stride[1], padding[0], padding[1], dilation[0], dilation[1])
        wsb = self.lib.tsg_conv2d_f32_exact_wgrad_ws_bytes(B, *geo)
        ws = torch.empty(max(wsb, 8), dtype=torch.uint8, device=x.device)
        L.check(self.lib.tsg_conv2d_f32_exact_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, *geo,
                                                    self._strides4(x), self._strides4(dw), self._strides4(dy),
                                                    ws.data_ptr(), wsb, L.stream_ptr(x)), "tsg_conv2d_f32_exact_wgrad")
        return dw

    def sgd_step(self, param, grad, buf, lr, momentum, weight_decay, grad_scale, first):
        L.check(self.lib.tsg_sgd_step(param.data_ptr(), grad.data_ptr(), buf.data_ptr(), param.numel(),
                                      float(lr), float(momentum), float(weight_decay), float(grad_scale),
                                      int(first), L.stream_ptr(param)), "tsg_sgd_step")


    def sgd_step_dev(self, param, grad, buf, lr_dev, lr_mult, momentum, weight_decay, grad_scale=1.0):
        L.check(self.lib.tsg_sgd_step_dev(param.data_ptr(), grad.data_ptr(), buf.data_ptr(), param.numel(),
                                          lr_dev.data_ptr(), float(lr_mult), float(momentum), float(weight_decay),
                                          float(grad_scale), L.stream_ptr(param)), "tsg_sgd_step_dev")

    # ---- 3x3 weight gradient ----------------------------------------------------
    def conv3x3_c64_supported(self, x, weight, stride, padding, dilation, groups):
        if x.dim() != 4 or x.dtype != torch.bfloat16 or not x.is_contiguous(memory_format=torch.channels_last):
            return False
        return bool(self.lib.tsg_conv3x3_c64_supported(L.dtype_code(x), x.shape[1], weight.shape[0], weight.shape[2],
                                                       weight.shape[3], stride, padding, dilation, groups))

    def conv3x3_c64_bnsums_supported(self, B, H, W, stride):
        """can the 64 -> 64 data gradient of this input size emit the BatchNorm backward sums in its epilogue (`bsum=`)?"""
        fn = self.lib.tsg_conv3x3_c64_dgrad_bnsums_partials if stride == 1 else self.lib.tsg_conv3x3_c64_s2_dgrad_partials
        key = ("c64_bsum", stride, B, H, W)
        v = self._npart.get(key)
        if v is None:
            v = self._npart[key] = fn(B, H, W)
        return v > 0

    @staticmethod
    def _check_bsum(bsum, like_shape):
        bx, fp = bsum
        if (tuple(bx.shape) != tuple(like_shape) or bx.dtype != torch.bfloat16
                or not bx.is_contiguous(memory_format=torch.channels_last)
                or fp.dtype != torch.float32 or not fp.is_contiguous() or fp.dim() != 2 or fp.shape[0] < 3
                or fp.shape[1] != like_shape[1]):
            raise ValueError("bsum = (x of the BatchNorm: bf16 channels_last, the gradient's shape; its forward pack fp32 [>=3, C])")
        return bx, fp

    def conv3x3_c64_fwd(self, x, wb, with_stats=False, stride=1, in_ab=None, addend=None, bsum=None):
        """x [B,64,H,W] bf16 channels_last, wb bf16 [64,64,3,3] channels_last, 3x3 / stride 1 or 2 / padding 1
        -> y (channels_last) or (y, partial [S,2,64]).  in_ab: fp32 [>=2, 64] whose rows 0 / 1 are the a / b of a BN forward
        pack: the convolution reads relu(a x + b) (normalise-on-load).
        bsum = (bn_x, fp) (stride 1, the launch being a DATA gradient: x = dy, wb = the rotated filter): the backward sums
        of the BatchNorm -> ReLU in front of the convolution in the epilogue -> (dx, partial [S,2,64]) with the layout of
        bn_bwd_reduce's partial."""
        if bsum is not None:
            if stride != 1 or with_stats or in_ab is not None or addend is not None:
                raise ValueError("conv3x3_c64_fwd: bsum goes with the plain stride-1 launch only")
            bx, fp = self._check_bsum(bsum, x.shape)
            B, _, H, W = x.shape
            S = self._count(("c64_bsum", 1, B, H, W), lambda: self.lib.tsg_conv3x3_c64_dgrad_bnsums_partials(B, H, W),
                            "tsg_conv3x3_c64_dgrad_bnsums_partials")
            y = torch.empty_like(x)
            partial = torch.empty((S, 2, 64), dtype=torch.float32, device=x.device)
            L.check(self.lib.tsg_conv3x3_c64_dgrad_bnsums(x.data_ptr(), wb.data_ptr(), y.data_ptr(), bx.data_ptr(),
                                                          fp.data_ptr(), partial.data_ptr(), B, H, W, L.stream_ptr(x)),
                    "tsg_conv3x3_c64_dgrad_bnsums")
            return y, partial
        if in_ab is not None and (in_ab.dtype != torch.float32 or not in_ab.is_contiguous() or in_ab.shape[-1] != 64):
            raise ValueError("conv3x3_c64_fwd: in_ab must be a contiguous fp32 [>=2, 64] pack")
        if not x.is_contiguous(memory_format=torch.channels_last) or not wb.is_contiguous(memory_format=torch.channels_last):
            raise ValueError("conv3x3_c64_fwd expects channels_last operands")
        B, _, H, W = x.shape
        lib = self.lib
        if stride == 1:
            y = torch.empty_like(x)
            fn, cnt, what = lib.tsg_conv3x3_c64_fwd, lib.tsg_conv3x3_c64_stats_partials, "tsg_conv3x3_c64_fwd"
        else:
            y = torch.empty((B, 64, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=x.dtype, device=x.device,
                            memory_format=torch.channels_last)
            fn, cnt, what = lib.tsg_conv3x3_c64_s2_fwd, lib.tsg_conv3x3_c64_s2_stats_partials, "tsg_conv3x3_c64_s2_fwd"
        partial = None
        if with_stats:
            S = self._count(("c64_stats", stride, B, H, W), lambda: cnt(B, H, W), what)
            partial = torch.empty((S, 2, 64), dtype=torch.float32, device=x.device)
        if addend is not None:
            if stride != 1 or with_stats or addend.shape != y.shape or addend.dtype != y.dtype \
                    or not addend.is_contiguous(memory_format=torch.channels_last):
                raise ValueError("conv3x3_c64_fwd: addend needs stride 1, no statistics, a bf16 channels_last tensor of y's shape")
            L.check(fn(x.data_ptr(), wb.data_ptr(), y.data_ptr(), None, L.ptr(in_ab), addend.data_ptr(), B, H, W,
                       L.stream_ptr(x)), what)
            return y
        if stride == 1:
            L.check(fn(x.data_ptr(), wb.data_ptr(), y.data_ptr(), L.ptr(partial), L.ptr(in_ab), None, B, H, W,
                       L.stream_ptr(x)), what)
        else:
            L.check(fn(x.data_ptr(), wb.data_ptr(), y.data_ptr(), L.ptr(partial), L.ptr(in_ab), B, H, W, L.stream_ptr(x)), what)
        return (y, partial) if with_stats else y

    def conv3x3_gen_supported(self, x, weight, stride, padding, dilation, groups):
        """the general stride-1 3x3 forward kernel (csrc/conv3g.hip): bf16 channels_last, Cin % 16 == 0, Cout % 64 == 0"""
        if x.dim() != 4 or x.dtype != torch.bfloat16 or not x.is_contiguous(memory_format=torch.channels_last):
            return False
        return bool(self.lib.tsg_conv3x3_gen_supported(L.dtype_code(x), x.shape[1], weight.shape[0], weight.shape[2],
                                                       weight.shape[3], stride, padding, dilation, groups))

    def conv3x3_gen_tile(self, B, H, W, Cin, Cout):
        """output channels per block (64 / 128) for this problem: the prepared filter is laid out for it"""
        return self._count(("g3_tile", B, H, W, Cin, Cout), lambda: self.lib.tsg_conv3x3_gen_tile(B, H, W, Cin, Cout),
                           "tsg_conv3x3_gen_tile")

    def conv3x3_gen_variant(self, B, H, W, Cin, Cout, with_in_ab=False):
        """0: 8-row pixel tiles (conv3g_fwd_k), 1: 16-row tiles with all staging by LDS-DMA (conv3h_fwd_k)"""
        v = self.lib.tsg_conv3x3_gen_variant(B, H, W, Cin, Cout, self.conv3x3_gen_tile(B, H, W, Cin, Cout), int(bool(with_in_ab)))
        if v < 0:
            L.check(v, "tsg_conv3x3_gen_variant")
        return v

    def conv3x3_gen_prep_filter(self, weight, mode, like, bn=None):
        """weight [O,I,3,3] channels_last (fp32 master or bf16) -> (wf, BN): the bf16 filter in MFMA fragment order for
        the convolution that will read `like` ([B,C,H,W]): mode 0 for conv(x, w), mode 1 for the data gradient
        conv(dy, rot180(w)^T).  bn: the tile width to lay the filter out for (default: what tsg_conv3x3_gen_fwd will use;
        32 for tsg_conv3x3_s2_dgrad)"""
        if weight.dim() != 4 or tuple(weight.shape[2:]) != (3, 3) or not weight.is_contiguous(memory_format=torch.channels_last):
            raise ValueError("conv3x3_gen_prep_filter expects a channels_last [O, I, 3, 3] weight")
        O, I = weight.shape[0], weight.shape[1]
        Cin, Cout = (O, I) if mode else (I, O)
        B, C, H, W = like.shape
        if C != Cin:
            raise ValueError("conv3x3_gen_prep_filter: the input does not have the filter's channel count")
        if bn is None:
            bn = self.conv3x3_gen_tile(B, H, W, Cin, Cout)
        if _SHADOW_PREP and weight.dtype == torch.float32 and weight.is_leaf and weight.requires_grad:
            from .shadow import bank                  # a parameter: its prepared images are refreshed once per optimizer step
            return bank.get_gen(weight, int(mode), bn), bn
        out = torch.empty(9 * O * I, dtype=torch.bfloat16, device=weight.device)
        L.check(self.lib.tsg_conv3x3_gen_prep_filter(weight.data_ptr(), L.dtype_code(weight), out.data_ptr(), O, I, int(mode),
                                                     bn, L.stream_ptr(weight)), "tsg_conv3x3_gen_prep_filter")
        return out, bn

    def conv3x3_gen_fwd(self, x, wf, Cout, with_stats=False, in_ab=None, addend=None):
        """x [B,Cin,H,W] bf16 channels_last, wf = conv3x3_gen_prep_filter(..., like=x) -> y [B,Cout,H,W] channels_last, or
        (y, partial [S,2,Cout]).  in_ab: fp32 [>=2, Cin] BN forward pack: the convolution reads relu(a x + b).
        addend: bf16 channels_last [B,Cout,H,W], y = bf16(bf16(conv) + addend)."""
        if not x.is_contiguous(memory_format=torch.channels_last) or x.dtype != torch.bfloat16:
            raise ValueError("conv3x3_gen_fwd expects a bf16 channels_last input")
        B, Cin, H, W = x.shape
        wf, bn = wf
        if wf.numel() != 9 * Cin * Cout or wf.dtype != torch.bfloat16 or bn != self.conv3x3_gen_tile(B, H, W, Cin, Cout):
            raise ValueError("conv3x3_gen_fwd: the prepared filter does not belong to this convolution")
        if in_ab is not None and (in_ab.dtype != torch.float32 or not in_ab.is_contiguous() or in_ab.shape[-1] != Cin):
            raise ValueError("conv3x3_gen_fwd: in_ab must be a contiguous fp32 [>=2, Cin] pack")
        y = torch.empty((B, Cout, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        if addend is not None and (with_stats or addend.shape != y.shape or addend.dtype != y.dtype
                                   or not addend.is_contiguous(memory_format=torch.channels_last)):
            raise ValueError("conv3x3_gen_fwd: addend must be a bf16 channels_last tensor of the output's shape (no statistics)")
        partial = None
        if with_stats:
            S = self._count(("g3_stats", B, H, W, Cin, Cout, bn),
                            lambda: self.lib.tsg_conv3x3_gen_stats_partials(B, H, W, Cin, Cout, bn),
                            "tsg_conv3x3_gen_stats_partials")
            partial = torch.empty((S, 2, Cout), dtype=torch.float32, device=x.device)
        L.check(self.lib.tsg_conv3x3_gen_fwd(x.data_ptr(), wf.data_ptr(), y.data_ptr(), L.ptr(partial), L.ptr(in_ab),
                                             L.ptr(addend), B, H, W, Cin, Cout, bn, L.stream_ptr(x)), "tsg_conv3x3_gen_fwd")
        return (y, partial) if with_stats else y

    def conv3x3_s2_dgrad_supported(self, Cin, Cout):
        return bool(self.lib.tsg_conv3x3_s2_dgrad_supported(L.BF16, Cin, Cout))

    def conv3x3_s2_dgrad(self, dy, weight, in_hw, addend=None, addend_sub=None):
        """dy [B,Cout,OH,OW] bf16 channels_last, weight [Cout,Cin,3,3] channels_last (fp32 master or bf16) of a 3x3 / stride 2
        / padding 1 convolution whose input was [B,Cin,H,W] = in_hw -> dx bf16 channels_last (+ addend, same shape;
        or + addend_sub [B,Cin,OH,OW], the gradient of x[:, :, ::2, ::2], at the even pixels)"""
        if not dy.is_contiguous(memory_format=torch.channels_last) or dy.dtype != torch.bfloat16:
            raise ValueError("conv3x3_s2_dgrad expects a bf16 channels_last dy")
        B, Cout = dy.shape[0], dy.shape[1]
        Cin = weight.shape[1]
        H, W = in_hw
        if weight.shape[0] != Cout or (H - 1) // 2 + 1 != dy.shape[2] or (W - 1) // 2 + 1 != dy.shape[3]:
            raise ValueError("conv3x3_s2_dgrad: dy does not belong to that weight / an input of that size")
        if addend is not None and (tuple(addend.shape) != (B, Cin, H, W) or addend.dtype != torch.bfloat16
                                   or not addend.is_contiguous(memory_format=torch.channels_last)):
            raise ValueError("conv3x3_s2_dgrad: addend must be a bf16 channels_last tensor of dx's shape")
        if addend_sub is not None and (addend is not None or tuple(addend_sub.shape) != (B, Cin, dy.shape[2], dy.shape[3])
                                       or addend_sub.dtype != torch.bfloat16
                                       or not addend_sub.is_contiguous(memory_format=torch.channels_last)):
            raise ValueError("conv3x3_s2_dgrad: addend_sub must be a bf16 channels_last [B, Cin, OH, OW] tensor (and the only addend)")
        wf, _ = self.conv3x3_gen_prep_filter(weight, 1, dy, bn=32)
        dx = torch.empty((B, Cin, H, W), dtype=torch.bfloat16, device=dy.device, memory_format=torch.channels_last)
        if addend_sub is not None:
            L.check(self.lib.tsg_conv3x3_s2_dgrad_subadd(dy.data_ptr(), wf.data_ptr(), dx.data_ptr(), addend_sub.data_ptr(), B, H,
                                                         W, Cin, Cout, L.stream_ptr(dy)), "tsg_conv3x3_s2_dgrad_subadd")
            return dx
        L.check(self.lib.tsg_conv3x3_s2_dgrad(dy.data_ptr(), wf.data_ptr(), dx.data_ptr(), L.ptr(addend), B, H, W, Cin, Cout,
                                              L.stream_ptr(dy)), "tsg_conv3x3_s2_dgrad")
        return dx

    def conv3x3_c64_s2_dgrad(self, dy, wt, in_hw, bsum=None):
        """dy [B,64,OH,OW] bf16 channels_last, wt = conv3x3_weight_rot180_t(w) -> dx [B,64,H,W] of the stride-2 convolution.
        bsum = (bn_x, fp): also the backward sums of the BatchNorm -> ReLU in front of the convolution -> (dx, partial [S,2,64])"""
        if not dy.is_contiguous(memory_format=torch.channels_last) or not wt.is_contiguous(memory_format=torch.channels_last):
            raise ValueError("conv3x3_c64_s2_dgrad expects channels_last operands")
        B = dy.shape[0]
        H, W = in_hw
        if (H - 1) // 2 + 1 != dy.shape[2] or (W - 1) // 2 + 1 != dy.shape[3]:
            raise ValueError("conv3x3_c64_s2_dgrad: dy does not belong to an input of that size")
        dx = torch.empty((B, 64, H, W), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
        if bsum is not None:
            bx, fp = self._check_bsum(bsum, dx.shape)
            S = self._count(("c64_bsum", 2, B, H, W), lambda: self.lib.tsg_conv3x3_c64_s2_dgrad_partials(B, H, W),
                            "tsg_conv3x3_c64_s2_dgrad_partials")
            partial = torch.empty((S, 2, 64), dtype=torch.float32, device=dy.device)
            L.check(self.lib.tsg_conv3x3_c64_s2_dgrad_bnsums(dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), bx.data_ptr(),
                                                             fp.data_ptr(), partial.data_ptr(), B, H, W, L.stream_ptr(dy)),
                    "tsg_conv3x3_c64_s2_dgrad_bnsums")
            return dx, partial
        L.check(self.lib.tsg_conv3x3_c64_s2_dgrad(dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), B, H, W, L.stream_ptr(dy)),
                "tsg_conv3x3_c64_s2_dgrad")
        return dx

    def conv3x3_wrw_supported(self, x, weight, stride, padding, dilation, groups):
        if x.dim() != 4 or x.dtype != torch.bfloat16 or not x.is_contiguous(memory_format=torch.channels_last):
            return False
        return bool(self.lib.tsg_conv3x3_wrw_gen_supported(L.dtype_code(x), x.shape[1], weight.shape[0], weight.shape[2],
                                                           weight.shape[3], stride, padding, dilation, groups))

    def conv3x3_wrw(self, x, dy, variant=None, stride=1, in_ab=None, out=None):
        """x [B,Cin,Hin,Win], dy [B,Cout,OH,OW] bf16 channels_last (Cin, Cout multiples of 64; 3x3, padding 1, stride 1 or
        2) -> dw fp32 [Cout,Cin,3,3] channels_last.  The pair-tiled kernel ("gen") computes every shape; 64 -> 64 / stride 1 can
        also take the single-pair kernels (variant "tr", or "v1" = the transposed-staging kernel; TSG_CONV_WRW_IMPL)."""
        if variant is None:
            # 64 -> 64 / stride 1: the pair-tiled kernel (buffer-load fetch) is 4-7 % faster than the single-pair "tr" kernel
            # at 16 x 64 x 256^2 (94-95 vs 99-101 us, profiles/r04_wrw_double_buffer_null.txt); TSG_CONV_WRW_IMPL=tr|v1 selects those
            variant = os.environ.get("TSG_CONV_WRW_IMPL", "gen")
        if in_ab is not None:                 # normalise-on-load: x is the input of the BN + ReLU in front of the convolution
            if in_ab.dtype != torch.float32 or not in_ab.is_contiguous() or in_ab.shape[-1] != x.shape[1]:
                raise ValueError("conv3x3_wrw: in_ab must be a contiguous fp32 [>=2, Cin] pack")
            if variant == "v1":
                variant = "tr"
        for t in (x, dy):
            if not t.is_contiguous(memory_format=torch.channels_last) or t.dtype != torch.bfloat16:
                raise ValueError("conv3x3_wrw expects bf16 channels_last tensors")
        B, Cin, H, W = x.shape
        Cout = dy.shape[1]
        if tuple(dy.shape) != (B, Cout, (H - 1) // stride + 1, (W - 1) // stride + 1):
            raise ValueError("conv3x3_wrw: dy does not have the output shape of a 3x3 / padding 1 / stride %d convolution of x" % stride)
        dw = out if out is not None else \
            torch.empty((Cout, Cin, 3, 3), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        if out is not None and (tuple(out.shape) != (Cout, Cin, 3, 3) or out.dtype != torch.float32
                                or not out.is_contiguous(memory_format=torch.channels_last)):
            raise ValueError("conv3x3_wrw: out must be an fp32 channels_last [Cout, Cin, 3, 3] tensor")
        if Cin == 64 and Cout == 64 and stride == 1 and variant != "gen":
            fn = self.lib.tsg_conv3x3_wrw_tr if variant == "tr" else self.lib.tsg_conv3x3_wrw
            ws = self._scratch("c3", self.lib.tsg_conv3x3_wrw_ws_bytes(), x.device)
            if in_ab is not None:
                L.check(self.lib.tsg_conv3x3_wrw_tr_norm(x.data_ptr(), in_ab.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, H, W,
                                                         ws.data_ptr(), ws.numel(), L.stream_ptr(x)), "tsg_conv3x3_wrw_tr_norm")
            else:
                L.check(fn(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, H, W, ws.data_ptr(), ws.numel(), L.stream_ptr(x)),
                        "tsg_conv3x3_wrw")
            return dw
        wsb = self.lib.tsg_conv3x3_wrw_gen_ws_bytes(B, H, W, Cin, Cout, stride)
        if wsb == 0:
            raise L.TsgError("conv3x3_wrw: unsupported shape %s -> %d channels, stride %d" % (tuple(x.shape), Cout, stride))
        ws = self._scratch("c3g", wsb, x.device)                  # per stream, grown to the largest layer (<= 38 MB)
        if in_ab is not None:
            L.check(self.lib.tsg_conv3x3_wrw_gen_norm(x.data_ptr(), in_ab.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, H, W, Cin,
                                                      Cout, int(stride), ws.data_ptr(), ws.numel(), L.stream_ptr(x)),
                    "tsg_conv3x3_wrw_gen_norm")
        else:
            L.check(self.lib.tsg_conv3x3_wrw_gen(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, H, W, Cin, Cout, int(stride),
                                                 ws.data_ptr(), ws.numel(), L.stream_ptr(x)), "tsg_conv3x3_wrw_gen")
        return dw

    def conv3x3_weight_rot180_t(self, w):
        """w [O,I,3,3] fp32 / bf16 channels_last -> bf16 [I,O,3,3] channels_last with the taps rotated by 180 degrees"""
        if w.dim() != 4 or tuple(w.shape[2:]) != (3, 3) or not w.is_contiguous(memory_format=torch.channels_last):
            raise L.TsgError("conv3x3_weight_rot180_t takes a channels_last [O, I, 3, 3] filter")
        O, I = w.shape[0], w.shape[1]
        out = torch.empty((I, O, 3, 3), dtype=torch.bfloat16, device=w.device, memory_format=torch.channels_last)
        L.check(self.lib.tsg_conv3x3_weight_rot180_t(w.data_ptr(), L.dtype_code(w), out.data_ptr(), O, I, L.stream_ptr(w)),
                "tsg_conv3x3_weight_rot180_t")
        return out

    # ---- training pre-processing ---------------------------------------------------
    def augment_crop(self, imgs, gts, geom, crop_hw, mean, std, pad_label=255, label_dtype=torch.int64, pad_pixel=-1.0,
                     inv_scale=None):
        """imgs[i] uint8 [H,W,3], gts[i] uint8 [H,W] (or gts=None) on the GPU; geom int32 numpy [n,7] = H, W, SH, SW, flip,
        crop_y, crop_x -> (float32 [n,3,CH,CW], labels [n,CH,CW] or None).  pad_pixel < 0: the normalised image is padded
        with 0 (TrainPre); >= 0: the raw image is padded with that value (evaluator).  inv_scale float64 [n,2] = the
        (fy, fx) factors cv2.resize was called with, None = derived from the sizes."""
        import numpy as np
        n = len(imgs)
        CH, CW = int(crop_hw[0]), int(crop_hw[1])
        dev = imgs[0].device
        for t in list(imgs) + list(gts or []):
            if t.dtype != torch.uint8 or not t.is_contiguous() or not t.is_cuda:
                raise L.TsgError("augment_crop takes contiguous uint8 tensors on the GPU")
        out = torch.empty((n, 3, CH, CW), dtype=torch.float32, device=dev)
        lab = torch.empty((n, CH, CW), dtype=label_dtype, device=dev) if gts is not None else None
        cap = self.lib.tsg_augment_max_samples()
        geom = np.ascontiguousarray(geom, dtype=np.int32).reshape(n, 7)
        m = np.ascontiguousarray(mean, dtype=np.float32)
        s = np.ascontiguousarray(std, dtype=np.float32)
        if inv_scale is not None:
            inv_scale = np.ascontiguousarray(inv_scale, dtype=np.float64).reshape(n, 2)
        for i0 in range(0, n, cap):
            k = min(cap, n - i0)
            pi = (C.c_void_p * k)(*[imgs[i0 + j].data_ptr() for j in range(k)])
            pg = (C.c_void_p * k)(*[gts[i0 + j].data_ptr() for j in range(k)]) if gts is not None else None
            L.check(self.lib.tsg_augment_crop(pi, pg, geom[i0:i0 + k].ctypes.data,
                                              inv_scale[i0:i0 + k].ctypes.data if inv_scale is not None else None,
                                              k, CH, CW, m.ctypes.data, s.ctypes.data,
                                              float(pad_pixel), int(pad_label), out[i0:].data_ptr(),
                                              lab[i0:].data_ptr() if lab is not None else None,
                                              _label_code(lab) if lab is not None else L.I64,
                                              L.stream_ptr(out)), "tsg_augment_crop")
        return out, lab

    def edge_labels(self, gts, geom, crop_hw, ignore_label=255, threshold=5, aperture=7, dilate_size=7, pad_label=255,
                    label_dtype=torch.int64):
        """DFN's border labels (dfn dataloader.py:24-29) for n samples: gts[i] uint8 [H,W] on the GPU, geom int32 [n,7] as
        for augment_crop -> [n, CH, CW] with values {0, 1, pad_label}."""
        import numpy as np
        n = len(gts)
        CH, CW = int(crop_hw[0]), int(crop_hw[1])
        dev = gts[0].device
        geom = np.ascontiguousarray(geom, dtype=np.int32).reshape(n, 7)
        out = torch.empty((n, CH, CW), dtype=label_dtype, device=dev)
        for i in range(n):
            if gts[i].dtype != torch.uint8 or not gts[i].is_contiguous() or not gts[i].is_cuda:
                raise L.TsgError("edge_labels takes contiguous uint8 tensors on the GPU")
            wsb = self.lib.tsg_edge_labels_ws_bytes(int(geom[i, 2]), int(geom[i, 3]))
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            L.check(self.lib.tsg_edge_labels(gts[i].data_ptr(), geom[i].ctypes.data, None, CH, CW, int(ignore_label),
                                             int(threshold), int(threshold), int(aperture), int(dilate_size), int(pad_label),
                                             out[i].data_ptr(), _label_code(out), ws.data_ptr(), wsb, L.stream_ptr(out)),
                    "tsg_edge_labels")
        return out

    def resize_bilinear_hp(self, x, OH, OW, out=None, accumulate=False):
        """x [..., IH, IW] f32 / bf16 contiguous -> fp32 [..., OH, OW], half-pixel centres (cv2 INTER_LINEAR on float data)"""
        _require_contiguous(x)
        IH, IW = x.shape[-2], x.shape[-1]
        NC = x.numel() // (IH * IW)
        if out is None:
            out = torch.empty(tuple(x.shape[:-2]) + (OH, OW), dtype=torch.float32, device=x.device)
            accumulate = False
        L.check(self.lib.tsg_resize_bilinear_hp(x.data_ptr(), out.data_ptr(), L.dtype_code(x), NC, IH, IW, int(OH), int(OW),
                                                int(accumulate), L.stream_ptr(x)), "tsg_resize_bilinear_hp")
        return out

    # ---- evaluation metric ----------------------------------------------------
    def confusion_map(self, pred, gt, n_cl, out=None):
        """pred, gt: class-index maps (int64 or uint8, same numel) -> int64 [n_cl*n_cl + 3], accumulated into `out`"""
        _require_contiguous(pred, gt)
        if pred.numel() != gt.numel():
            raise ValueError("pred and gt must have the same number of pixels")
        if out is None:
            out = torch.zeros(n_cl * n_cl + 3, dtype=torch.int64, device=gt.device)
        L.check(self.lib.tsg_confusion_map(pred.data_ptr(), _label_code(pred), gt.data_ptr(), _label_code(gt),
                                           gt.numel(), n_cl, out.data_ptr(), L.stream_ptr(gt)), "tsg_confusion_map")
        return out

    def confusion_logits(self, logits, gt, n_cl, out=None):
        """logits [B,C,H,W] (f32 / bf16, contiguous), gt [B,H,W] -> as confusion_map with pred = argmax_C"""
        _require_contiguous(logits, gt)
        B, Cc = logits.shape[0], logits.shape[1]
        HW = logits.numel() // max(B * Cc, 1)
        if gt.numel() != B * HW:
            raise ValueError("gt must have one label per pixel of logits")
        if out is None:
            out = torch.zeros(n_cl * n_cl + 3, dtype=torch.int64, device=gt.device)
        L.check(self.lib.tsg_confusion_logits(logits.data_ptr(), L.dtype_code(logits), gt.data_ptr(), _label_code(gt),
                                              B, Cc, HW, n_cl, out.data_ptr(), L.stream_ptr(gt)), "tsg_confusion_logits")
        return out

    SGD_MAX_SEGS = 128
    SGD_MAX_GROUPS = 24          # TSG_SGD_MAX_GROUPS (include/tsg_hip.h)

    def sgd_multi_blockmap(self, numel, device):
        """Static block -> (tensor, chunk) table for sgd_multi_step_dev, as a device int32 tensor."""
        import numpy as np
        n = np.ascontiguousarray(numel, dtype=np.int64)
        nb = self.lib.tsg_sgd_multi_blockmap(n.ctypes.data, len(n), None, 0)
        if nb < 0:
            L.check(int(nb), "tsg_sgd_multi_blockmap")
        host = np.empty((nb, 2), dtype=np.int32)
        L.check(int(min(0, self.lib.tsg_sgd_multi_blockmap(n.ctypes.data, len(n), host.ctypes.data, nb))),
                "tsg_sgd_multi_blockmap")
        return torch.from_numpy(host).to(device)

    def sgd_multi_step_dev(self, ptrs, numel, group, lr_dev, momentum, weight_decay, blockmap, grad_scale=1.0):
        """ptrs: uint64 numpy [3, nseg] (param, grad, momentum buffer addresses); numel int64 / group int32 numpy
        [nseg]; momentum / weight_decay float32 numpy [ngroups]; blockmap from sgd_multi_blockmap."""
        L.check(self.lib.tsg_sgd_multi_step_dev(ptrs[0].ctypes.data, ptrs[1].ctypes.data, ptrs[2].ctypes.data,
                                                numel.ctypes.data, group.ctypes.data, len(numel), lr_dev.data_ptr(),
                                                momentum.ctypes.data, weight_decay.ctypes.data, len(momentum),
                                                blockmap.data_ptr(), blockmap.shape[0], float(grad_scale),
                                                L.stream_ptr(lr_dev)), "tsg_sgd_multi_step_dev")

    def multi_copy(self, srcs, dsts, blockmap=None, scale=1.0):
        """dsts[i].copy_(srcs[i]) * scale for lists of dense fp32 tensors with pairwise equal element order
        (<= SGD_MAX_SEGS per launch); returns the block map so a static list can reuse it."""
        import numpy as np
        numel = np.array([t.numel() for t in srcs], dtype=np.int64)
        if blockmap is None:
            blockmap = self.sgd_multi_blockmap(numel, srcs[0].device)
        ptrs = np.array([[t.data_ptr() for t in srcs], [t.data_ptr() for t in dsts]], dtype=np.uint64)
        L.check(self.lib.tsg_multi_copy_f32(ptrs[0].ctypes.data, ptrs[1].ctypes.data, numel.ctypes.data, len(srcs),
                                            blockmap.data_ptr(), blockmap.shape[0], float(scale),
                                            L.stream_ptr(srcs[0])), "tsg_multi_copy_f32")
        return blockmap


_provider = None


def provider():
    """The kernel provider of the product path: HipKernels, or an exception."""
    global _provider
    if _provider is None:
        _provider = HipKernels()
    return _provider


def _set_provider_for_tests(p):
    """tests/ only: swap in a stand-in provider to exercise host logic on CPU."""
    global _provider
    old = _provider
    _provider = p
    return old


def _nbytes(t):
    return 0 if t is None else t.numel() * t.element_size()


def _bsum_bytes(kw):
    """the BatchNorm input a data gradient reads for the backward sums of its epilogue (bsum=(bn_x, fp))"""
    b = (kw or {}).get("bsum")
    return _nbytes(b[0]) if b is not None else 0


# algorithmic bytes per launch (each operand read once + each result written once, the
# convention of the reference's own tools/benchmark/compute_memory.py:49-72); DESIGN.md §4
_ALGO_BYTES = {
    "bn_stats": lambda a, r: _nbytes(a[0]),
    "bn_apply_fwd": lambda a, r: 2 * _nbytes(a[0]) + _nbytes(a[1]),
    "bn_bwd_reduce": lambda a, r: 2 * _nbytes(a[0]) + _nbytes(a[2]),
    "bn_bwd_apply": lambda a, r: 3 * _nbytes(a[0]) + _nbytes(a[2]) + (_nbytes(a[0]) if a[9] else 0),
    "bn_apply_fwd_bits": lambda a, r: 2 * _nbytes(a[0]) + _nbytes(a[1]) + _nbytes(a[0]) // 16,
    "bn_bwd_reduce_bits": lambda a, r: 2 * _nbytes(a[0]) + _nbytes(a[2]),
    "bn_bwd_apply_bits": lambda a, r: 3 * _nbytes(a[0]) + _nbytes(a[2]) + (_nbytes(a[0]) if a[8] else 0),
    "ohem_fwd": lambda a, r: _nbytes(a[0]) + _nbytes(a[1]) + 8 * a[1].numel(),
    "ohem_bwd": lambda a, r: 2 * _nbytes(a[0]) + _nbytes(a[1]) + 8 * a[1].numel(),
    "ohem_up_fwd": lambda a, r: _nbytes(a[0]) + _nbytes(a[1]) + 8 * a[1].numel(),
    "ohem_up_bwd": lambda a, r: 2 * _nbytes(a[0]) + _nbytes(a[1]) + 8 * a[1].numel(),
    "upsample_fwd": lambda a, r: _nbytes(a[0]) + _nbytes(a[1]) + _nbytes(r),
    "upsample_presum_fwd": lambda a, r: _nbytes(a[0]) + _nbytes(a[1]) + _nbytes(r),
    "upsample_bwd": lambda a, r: _nbytes(a[0]) + _nbytes(r),
    "upsample_fwd_nhwc": lambda a, r: _nbytes(a[0]) + _nbytes(a[1]) + _nbytes(r),
    "upsample_bwd_nhwc": lambda a, r: _nbytes(a[0]) + _nbytes(r),
    "bn_apply_fwd_mixed": lambda a, r: 2 * _nbytes(a[0]),
    "bn_bwd_reduce_mixed": lambda a, r: 2 * _nbytes(a[0]),
    "bn_bwd_apply_mixed": lambda a, r: 3 * _nbytes(a[0]),
    "chanscale_fwd": lambda a, r: 2 * _nbytes(a[0]),
    "chanscale_bwd": lambda a, r: 3 * _nbytes(a[0]),
    "chanscale_bwd_ds": lambda a, r: 2 * _nbytes(a[0]),
    "chanscale_bwd_dx": lambda a, r: 2 * _nbytes(a[0]),
    "maxpool_fwd": lambda a, r: _nbytes(a[0]) + _nbytes(r[0]) + _nbytes(r[1]),
    "maxpool_bwd": lambda a, r: _nbytes(a[0]) + _nbytes(a[1]) + _nbytes(r),
    "gap_fwd": lambda a, r: _nbytes(a[0]),
    "stem_conv_fwd": lambda a, r: _nbytes(a[0]) + _nbytes(r),
    "stem_conv_fwd_stats": lambda a, r: _nbytes(a[0]) + _nbytes(r),
    "bn_relu_pool_fwd": lambda a, r: _nbytes(a[0]) + _nbytes(r[0]) + _nbytes(r[1]),
    "bn_relu_pool_bwd_reduce": lambda a, r: _nbytes(a[0]) + _nbytes(a[1]) + _nbytes(a[2]),
    "bn_relu_pool_bwd_apply": lambda a, r: _nbytes(a[0]) + _nbytes(a[1]) + 2 * _nbytes(a[2]),
    "stem_conv_wrw": lambda a, r: _nbytes(a[0]) + _nbytes(a[1]),
    "stem_conv_wrw_bn": lambda a, r: _nbytes(a[0]) + _nbytes(a[1]) + _nbytes(a[2]),
    # the recomputing stem (round 6): the image, the pooled side arrays; y is never in memory
    "stem_conv_stats": lambda a, r: _nbytes(a[0]),
    "stem_conv_bn_relu_pool_fwd": lambda a, r: _nbytes(a[0]) + _nbytes(r[0]) + _nbytes(r[1]),
    "stem_conv_bn_relu_pool_bwd_reduce": lambda a, r: _nbytes(a[0]) + _nbytes(a[2]) + _nbytes(a[3]),
    "stem_conv_wrw_bn_pool": lambda a, r, kw=None: (_nbytes(a[0]) + _nbytes(a[2]) + _nbytes(a[3])
                                                    + (_nbytes((kw or {}).get("xc")) if (kw or {}).get("xc") is not None else 0)),
    "conv3x3_wrw": lambda a, r: _nbytes(a[0]) + _nbytes(a[1]),
    "conv3x3_c64_fwd": lambda a, r, kw=None: _nbytes(a[0]) + _nbytes(r) + _bsum_bytes(kw),
    "conv3x3_gen_fwd": lambda a, r: _nbytes(a[0]) + _nbytes(a[1][0]) + _nbytes(r[0] if isinstance(r, tuple) else r),
    "conv3x3_c64_s2_dgrad": lambda a, r, kw=None: _nbytes(a[0]) + _nbytes(r) + _bsum_bytes(kw),
    "conv3x3_s2_dgrad": lambda a, r: _nbytes(a[0]) + _nbytes(r),
    "gap_bwd": lambda a, r: _nbytes(r),
}


# algorithmic flops per launch of the MFMA kernels (2 * MACs of the convolution they compute)
_ALGO_FLOPS = {
    "conv3x3_wrw": lambda a, r: 2 * 9 * a[1].numel() * a[0].shape[1],          # dy elements x C_in x 9 taps
    "stem_conv_fwd": lambda a, r: 2 * 147 * r.numel(),
    "conv3x3_c64_fwd": lambda a, r: 2 * 9 * 64 * r.numel(),
    "conv3x3_gen_fwd": lambda a, r: 2 * 9 * a[0].shape[1] * (r[0] if isinstance(r, tuple) else r).numel(),
    "conv3x3_c64_s2_dgrad": lambda a, r: 2 * 9 * 64 * a[0].numel(),
    "conv3x3_s2_dgrad": lambda a, r: 2 * 9 * a[1].shape[1] * a[0].numel(),
    "stem_conv_fwd_stats": lambda a, r: 2 * 147 * r.numel(),
    "stem_conv_wrw": lambda a, r: 2 * 147 * a[1].numel(),
    "stem_conv_wrw_bn": lambda a, r: 2 * 147 * a[1].numel(),
    # y elements = 64 x OH x OW per image; the weight gradient evaluates the convolution AND its gradient product
    "stem_conv_stats": lambda a, r: 2 * 147 * _stem_y_numel(a[0]),
    "stem_conv_bn_relu_pool_fwd": lambda a, r: 2 * 147 * _stem_y_numel(a[0]),
    "stem_conv_bn_relu_pool_bwd_reduce": lambda a, r: 2 * 147 * _stem_y_numel(a[0]),
    "stem_conv_wrw_bn_pool": lambda a, r: 2 * 2 * 147 * _stem_y_numel(a[0]),
}


def _stem_y_numel(x):
    B, _, H, W = x.shape
    return B * 64 * ((H - 1) // 2 + 1) * ((W - 1) // 2 + 1)
MFMA_PEAK_TFLOPS = 2500.0        # dense bf16, MI355X_MICROARCH.md


class CallCounter:
    """Counts the calls of every public method of a provider (= the C-ABI entry points the host logic reached), so a
    test or bench.py can assert WHICH kernels a step took — e.g. that the fused head, the summing up-sampling and our
    convolutions ran for an UNCHANGED reference network.py, not a silent eager fallback.  `stop()` restores the provider."""

    def __init__(self, prov):
        self.prov = prov
        self.counts = {}
        self._names = [n for n in dir(type(prov)) if not n.startswith("_") and callable(getattr(type(prov), n))]
        for n in self._names:
            self._wrap(n)

    def _wrap(self, name):
        fn = getattr(self.prov, name)
        counts = self.counts

        def counted(*args, **kw):
            counts[name] = counts.get(name, 0) + 1
            return fn(*args, **kw)

        setattr(self.prov, name, counted)

    def stop(self):
        for n in self._names:
            try:
                delattr(self.prov, n)
            except AttributeError:
                pass
        return dict(self.counts)


class KernelTimer:
    """Brackets every launch of the streaming kernels with HIP events recorded on
    the stream the kernel is enqueued on (torch's current stream) and accumulates
    algorithmic bytes, so bench.py can report achieved GB/s per kernel live."""

    def __init__(self, prov, names=None):
        self.prov = prov
        self.names = list(names or _ALGO_BYTES.keys())
        self.enabled = True          # bench.py brackets only every few timed steps: two event records per launch cost GPU time
        self.records = {n: [] for n in self.names}
        self.marks = []              # mark_step(): records per label at the end of every instrumented step
        self._orig = {}
        for n in self.names:
            self._wrap(n)

    def _wrap(self, name):
        fn = getattr(self.prov, name)
        self._orig[name] = fn
        rec = self.records[name]
        cost = _ALGO_BYTES[name]
        flops = _ALGO_FLOPS.get(name)

        def timed(*args, **kw):
            if not self.enabled:
                return fn(*args, **kw)
            s = torch.cuda.Event(enable_timing=True)
            e = torch.cuda.Event(enable_timing=True)
            s.record()
            out = fn(*args, **kw)
            e.record()
            res = out if name in ("maxpool_fwd", "bn_relu_pool_fwd") else (out[0] if isinstance(out, tuple) else out)
            by = cost(args, res, kw) if cost.__code__.co_argcount == 3 else cost(args, res)
            rec.append((s, e, by, flops(args, res) if flops else 0))
            return out

        setattr(self.prov, name, timed)

    def mark_step(self):
        """End of one instrumented step.  With several marked steps of equal launch counts, every launch is represented by
        the MINIMUM of its brackets over the steps: a bracket is start event, host-side call, kernel, end event — in a
        host-bound instrumented step the GPU waits inside it for the host, and one host hiccup (a collection, an allocation)
        used to put 10 ms on one stem launch and make `stem` the line's dominant family at 0.03 of its roof (round 6)."""
        self.marks.append({n: len(r) for n, r in self.records.items()})

    def _measured(self):
        """label -> [(ms, algorithmic bytes, flops)] of ONE step"""
        out = {}
        steps = len(self.marks)
        for n, recs in self.records.items():
            if not recs:
                continue
            el = [(r[0].elapsed_time(r[1]), r[2], r[3]) for r in recs]
            per = len(recs) // steps if steps > 1 else 0
            if steps > 1 and per > 0 and all(m.get(n, 0) == per * (i + 1) for i, m in enumerate(self.marks)):
                out[n] = [min((el[k * per + j] for k in range(steps)), key=lambda t: t[0]) for j in range(per)]
            elif steps > 1:                              # launch counts differ from step to step: the last step as it is
                lo = self.marks[-2].get(n, 0)
                out[n] = el[lo:] or el
            else:
                out[n] = el
        return out

    def stop(self):
        for n, fn in self._orig.items():
            try:
                delattr(self.prov, n)
            except AttributeError:
                setattr(self.prov, n, fn)
        torch.cuda.synchronize()
        self.stats = {}
        self.eff = self._measured()
        for n, recs in self.eff.items():
            if recs:
                ms = sum(r[0] for r in recs)
                by = sum(r[1] for r in recs)
                fl = sum(r[2] for r in recs)
                self.stats[n] = {"launches": len(recs), "total_ms": round(ms, 3), "avg_us": round(ms * 1e3 / len(recs), 2),
                                 "algo_MB_per_launch": round(by / len(recs) / 1e6, 3),
                                 "GBps": round(by / (ms * 1e-3) / 1e9, 1) if ms > 0 else None}
                if fl and ms > 0:
                    self.stats[n]["algo_GFLOP_per_launch"] = round(fl / len(recs) / 1e9, 3)
                    self.stats[n]["TFLOPs"] = round(fl / (ms * 1e-3) / 1e12, 1)

    def summary(self):
        return self.stats

    # kernel families of the bench line's `roofline` (the four bn_* labels are ONE family — SyncBN — and
    # the line must name the family with the largest total time, not the largest label)
    FAMILIES = {
        "syncbn": ("bn_stats", "bn_apply_fwd", "bn_bwd_reduce", "bn_bwd_apply", "bn_apply_fwd_bits", "bn_bwd_reduce_bits",
                   "bn_bwd_apply_bits", "bn_apply_fwd_mixed", "bn_bwd_reduce_mixed",
                   "bn_bwd_apply_mixed", "bn_relu_pool_fwd", "bn_relu_pool_bwd_reduce", "bn_relu_pool_bwd_apply"),
        "conv3x3_wrw": ("conv3x3_wrw",),
        "conv3x3_fwd_dgrad": ("conv3x3_gen_fwd", "conv3x3_c64_fwd", "conv3x3_c64_s2_dgrad", "conv3x3_s2_dgrad"),
        "fused_heads": ("ohem_up_fwd", "ohem_up_bwd", "ohem_fwd", "ohem_bwd"),
        "stem": ("stem_conv_fwd", "stem_conv_fwd_stats", "stem_conv_wrw", "stem_conv_wrw_bn", "stem_conv_stats",
                 "stem_conv_bn_relu_pool_fwd", "stem_conv_bn_relu_pool_bwd_reduce", "stem_conv_wrw_bn_pool"),
        "upsample": ("upsample_fwd", "upsample_presum_fwd", "upsample_bwd", "upsample_fwd_nhwc", "upsample_bwd_nhwc"),
    }

    def family_stats(self):
        """Per family: launches, total ms, algorithmic GB/s (and TFLOP/s for the matrix kernels), summed over its labels."""
        fam = {}
        for name, members in self.FAMILIES.items():
            recs = [r for m in members for r in self.eff.get(m, ())]
            if not recs:
                continue
            ms = sum(r[0] for r in recs)
            by = sum(r[1] for r in recs)
            fl = sum(r[2] for r in recs)
            if ms <= 0:
                continue
            st = {"launches": len(recs), "total_ms": round(ms, 3), "avg_us": round(ms * 1e3 / len(recs), 2),
                  "algo_MB_per_launch": round(by / len(recs) / 1e6, 3), "GBps": round(by / (ms * 1e-3) / 1e9, 1),
                  "members": [m for m in members if self.eff.get(m)]}
            if fl:
                st["algo_GFLOP_per_launch"] = round(fl / len(recs) / 1e9, 3)
                st["TFLOPs"] = round(fl / (ms * 1e-3) / 1e12, 1)
            fam[name] = st
        return fam

    def dominant(self):
        """Name of the instrumented kernel label with the largest total time (None if nothing ran)."""
        return max(self.stats, key=lambda n: self.stats[n]["total_ms"]) if self.stats else None

    def dominant_family(self):
        fam = self.family_stats()
        return max(fam, key=lambda n: fam[n]["total_ms"]) if fam else None

    def _roof_entry(self, name, st, peak_gbs, tj):
        traffic = busy = None
        if tj is not None:
            # PMC bytes per launch (separate rocprofv3 --pmc passes), launch-weighted over the family's labels
            tr = [(tj.get(m), self.stats[m]["launches"]) for m in st.get("members", [name]) if m in self.stats]
            if tr and all(t is not None for t, _ in tr):
                traffic = int(sum(t * n for t, n in tr) / sum(n for _, n in tr))
            bz = [((tj.get("_mfma_busy_frac") or {}).get(m), self.stats[m]["total_ms"]) for m in st.get("members", [name])
                  if m in self.stats]
            if bz and all(b is not None for b, _ in bz):
                busy = round(sum(b * w for b, w in bz) / sum(w for _, w in bz), 4)
        out = {"bound": "hbm", "kernel": name, "launches": st["launches"], "total_ms": st["total_ms"],
               "achieved": st["GBps"], "peak": peak_gbs, "unit": "GB/s",
               "frac": round(st["GBps"] / peak_gbs, 4), "traffic": traffic,
               "algo_bytes_per_launch": int(st["algo_MB_per_launch"] * 1e6), "avg_launch_us": st["avg_us"]}
        if "TFLOPs" in st and st["TFLOPs"] / MFMA_PEAK_TFLOPS > out["frac"]:
            # a matrix kernel: the roof it is closer to is the MFMA one (the HBM figure stays alongside)
            out.update({"bound": "mfma", "achieved": st["TFLOPs"], "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(st["TFLOPs"] / MFMA_PEAK_TFLOPS, 4),
                        "algo_flops_per_launch": int(st["algo_GFLOP_per_launch"] * 1e9),
                        "hbm_GBps": st["GBps"], "hbm_frac": round(st["GBps"] / peak_gbs, 4)})
            if busy is not None:
                out["mfma_busy_frac"] = busy             # SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CYCLES), same source
        return out

    def roofline(self, peak_gbs, profiles_dir=None, by_family=False):
        """The bench line's `roofline` object.  by_family: the dominant kernel FAMILY (SyncBN = all bn_* passes, ...)
        with the three largest families listed under `top_families`; otherwise the dominant single label."""
        tj = source = None
        if profiles_dir:
            import json
            import os
            f = os.path.join(profiles_dir, "traffic.json")
            if os.path.exists(f):
                tj = json.load(open(f))
                # NOT a counter of this run: rocprofv3 cannot be attached from inside the process it profiles
                source = "profiles/traffic.json@" + str(tj.get("_round", "r02")) + \
                    " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this bench command, tools/pmc_traffic.sh)"
        if by_family:
            fam = self.family_stats()
            order = sorted(fam, key=lambda n: -fam[n]["total_ms"])
            out = self._roof_entry(order[0], fam[order[0]], peak_gbs, tj)
            out["family_members"] = fam[order[0]]["members"]
            out["traffic_source"] = source
            out["top_families"] = [self._roof_entry(n, fam[n], peak_gbs, tj) for n in order[:3]]
            return out
        name = self.dominant()
        out = self._roof_entry(name, dict(self.stats[name], members=[name]), peak_gbs, tj)
        out["traffic_source"] = source
        return out
