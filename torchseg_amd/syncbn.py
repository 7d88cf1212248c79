"""SyncBatchNorm host logic: the autograd function and the nn.Module that the
reference obtains from `apex.parallel` (model/*/train.py:24-25,54-55).

Arithmetic: furnace/legacy/sync_bn/syncbn.py:32-52,86-98 (sum / square-sum ->
mean, biased var for normalisation, unbiased var for the running estimate) and
syncbn_kernel.cu:92-138,160-174 for the backward.  Exchange steps (one per
direction, as §8(e) of SURVEY.md lists):
  forward : all-reduce(SUM) of [sum x | sum x^2 | count_hi | count_lo]  (2C+2 fp32)
  backward: all-reduce(SUM) of [sum dy' | sum dy' (x - mean)]           (2C fp32)
With world_size 1 both collectives are skipped.

Extension over the reference surface (used by our furnace/seg_opr and
base_model, invisible to an unchanged network.py): `forward(x, residual=None,
relu=False)` fuses the residual add and the ReLU that follow the BN into the
normalise kernel, and their backward into the BN backward kernels.
"""
import os

import torch
import torch.distributed as dist
from torch.nn.modules.batchnorm import _BatchNorm

from . import kernels as K

# Set by the DDP wrapper when the model runs channels_last: an NCHW activation
# (only conv stems produce one) then leaves the BN already converted to
# channels_last through the mixed-layout kernels.
PREFER_CHANNELS_LAST_OUTPUT = False


def _world(group):
    """Ranks that share batch statistics.  TSG_FORCE_COLLECTIVES=1 makes a 1-rank
    process group take the multi-rank code path (collapse -> all-reduce -> finalize),
    which is how the N > 1 path is exercised on a single-GPU box."""
    if dist.is_available() and dist.is_initialized():
        w = dist.get_world_size(group)
        if w == 1 and os.environ.get("TSG_FORCE_COLLECTIVES", "0") == "1":
            return 2
        return w
    return 1


def _exchange(msg, group):
    """All-reduce(SUM) of a statistics message across the ranks of `group`: on HIP tensors one tsg_comm call on the
    compute stream (RCCL, or the one-shot xGMI mailbox kernel), on CPU tensors torch.distributed (gloo tests)."""
    from . import comm
    c = comm.get(group, like=msg)
    if c is not None:
        c.small_all_reduce(msg)
    else:
        dist.all_reduce(msg, op=dist.ReduceOp.SUM, group=group)


# TSG_BN_FP32_GATHER=1|0 (default 1, round 6): fp32 tensors (the parity mode: north_star's "fp32 loss / logits within 1e-4 of
# the reference CPU path") exchange their statistics by ALL-GATHER of hi / lo word pairs instead of the fp32 all-reduce.
# The fp64 accumulation of csrc/bn.hip (RedAcc<float, .>) used to end at the rank boundary: bn_collapse rounded a rank's sums
# to the fp32 message, and E[x^2] - mean^2 of the batch-2 global-context BatchNorm cancelled again (3-6e-4 of the logits,
# ADVICE r4 / VERDICT r5 weak 3).  Sending hi + lo through an all-REDUCE would not help (the collective adds in fp32); gathered,
# every rank folds the 2 x world rows in fp64 itself (bn_finalize / bn_bwd_coeffs sum partial rows in fp64 anyway), in rank
# order, so the result is also bit-identical on all ranks.  Message: 4C + 2 floats per rank forward, 4C backward; bf16
# tensors (the benched path) keep the 2C + 2 all-reduce.
_FP32_GATHER = os.environ.get("TSG_BN_FP32_GATHER", "1") != "0"


def _all_gather(msg, group, world):
    """[world, len(msg)] = every rank's message, rank order (tsg_comm on HIP tensors, torch.distributed otherwise)."""
    from . import comm
    out = torch.empty((world, msg.numel()), dtype=msg.dtype, device=msg.device)
    c = comm.get(group, like=msg)
    if c is not None and c.world == world:
        c.all_gather(msg, out.view(-1))
    elif c is not None and c.world == 1:                   # TSG_FORCE_COLLECTIVES on a 1-rank group: "world" is 2; the
        c.all_gather(msg, out[0])                          # collective still runs, the absent rank contributes zeros
        out[1:].zero_()
    elif dist.is_initialized() and dist.get_world_size(group) == world:
        dist.all_gather(list(out.unbind(0)), msg, group=group)
    else:                                                  # TSG_FORCE_COLLECTIVES on a 1-rank group: "world" is 2
        out.copy_(msg.unsqueeze(0).expand_as(out))
        out[1:].zero_()
    return out


def _hilo_rows(partial, S):
    """The fp64 fold of a partial's rows as one hi and one lo fp32 row: [2, 2, C]."""
    d = partial[:S].to(torch.float64).sum(0)
    hi = d.to(torch.float32)
    return torch.stack([hi, (d - hi.to(torch.float64)).to(torch.float32)])


def _gather_hilo(partial, S, C, group, world, count=None):
    """-> (rows [2 world, 2, C] whose fp64 column sums are the global sums, count_dev [2] or None)"""
    words = [_hilo_rows(partial, S).reshape(-1)]
    if count is not None:
        words.append(torch.tensor([float(count // 4096), float(count % 4096)], dtype=torch.float32, device=partial.device))
    got = _all_gather(torch.cat(words), group, world)
    rows = got[:, :4 * C].reshape(2 * world, 2, C).contiguous()
    return rows, (got[:, 4 * C:].sum(0) if count is not None else None)       # the count words stay exact under fp32 sums


def _dense(x):
    """Return (x_dense, layout tuple): copies only when x is neither NCHW- nor NHWC-dense."""
    lay = K.bn_layout(x)
    if lay is None:
        if x.dim() == 4 and x.stride(1) == 1:
            x = x.contiguous(memory_format=torch.channels_last)
        else:
            x = x.contiguous()
        lay = K.bn_layout(x)
    return x, lay


def _like(t, ref):
    """t with ref's dtype and strides (no copy when it already matches)."""
    if t.dtype != ref.dtype:
        t = t.to(ref.dtype)
    if t.stride() != ref.stride():
        out = torch.empty_like(ref)
        out.copy_(t)
        t = out
    return t


def _batch_statistics(kp, x, layout, N, C, HW, mod, gamma, beta, group, world, hint=None):
    """stats -> (all-reduce) -> finalize.  Returns (invstd, fwd_pack, count_dev).  `hint`: per-block sums [S,2,C] the
    producer of x collected in its epilogue (stemconv.attach_bn_partial); the statistics pass over x is then skipped."""
    n_local = N * HW
    if world == 1 and n_local <= 1:
        raise ValueError("Expected more than 1 value per channel when training, got input size {}"
                         .format(tuple(x.shape)))
    if hint is not None:
        partial, S = hint, hint.shape[0]
    else:
        partial, S = kp.bn_stats(x, layout, N, C, HW)
    rm = mod.running_mean if mod.track_running_stats else None
    rv = mod.running_var if mod.track_running_stats else None
    nbt = mod.num_batches_tracked if mod.track_running_stats else None
    momentum = 0.0 if mod.momentum is None else float(mod.momentum)
    if world > 1 and _FP32_GATHER and x.dtype == torch.float32:
        rows, count_dev = _gather_hilo(partial, S, C, group, world, count=n_local)
        _, invstd, fp = kp.bn_finalize(rows, 2 * world, C, 0.0, count_dev, float(mod.eps), momentum,
                                       gamma, beta, rm, rv, nbt)
        return invstd, fp, count_dev
    if world > 1:
        msg = torch.empty(2 * C + 2, dtype=torch.float32, device=x.device)
        kp.bn_collapse(partial, S, C, msg, count=n_local)      # sums and the exactly summable count in one launch
        _exchange(msg, group)
        count_dev = msg[2 * C:]
        _, invstd, fp = kp.bn_finalize(msg, 1, C, 0.0, count_dev, float(mod.eps), momentum,
                                       gamma, beta, rm, rv, nbt)
        return invstd, fp, count_dev
    _, invstd, fp = kp.bn_finalize(partial, S, C, float(n_local), None, float(mod.eps), momentum,
                                   gamma, beta, rm, rv, nbt)
    return invstd, fp, None


def _backward_pack(kp, partial, S, C, n_local, invstd, fp, count_dev, use_batch_stats, group, world, device,
                   parity=False):
    """(all-reduce of the backward sums) -> dgamma, dbeta (local) and the bwd pack (global).  `parity`: fp32 tensors
    exchange hi / lo rows by all-gather (_FP32_GATHER above)."""
    if use_batch_stats and world > 1 and parity and _FP32_GATHER:
        dgamma, dbeta, _ = kp.bn_bwd_coeffs(partial, S, C, 1.0, None, True, invstd, fp, True, False)
        rows, _ = _gather_hilo(partial, S, C, group, world)
        _, _, bp = kp.bn_bwd_coeffs(rows, 2 * world, C, 0.0, count_dev, True, invstd, fp, False, True)
        return dgamma, dbeta, bp
    if use_batch_stats and world > 1:
        sums = torch.empty(2 * C, dtype=torch.float32, device=device)
        kp.bn_collapse(partial, S, C, sums)
        dgamma, dbeta, _ = kp.bn_bwd_coeffs(sums, 1, C, 1.0, None, True, invstd, fp, True, False)
        _exchange(sums, group)
        _, _, bp = kp.bn_bwd_coeffs(sums, 1, C, 0.0, count_dev, True, invstd, fp, False, True)
        return dgamma, dbeta, bp
    return kp.bn_bwd_coeffs(partial, S, C, float(n_local), None, use_batch_stats, invstd, fp, True, True)


# TSG_BN_MASKBITS=1|0 (default 1, round 6): the block tail BN -> (+identity) -> ReLU keeps its ReLU mask as one bit per
# element (tsg_bn_apply_fwd_maskbits) and its two backward kernels read that instead of the stored output: 2 of the 8
# tensor passes of the layer's backward gone, same values.
_MASKBITS = os.environ.get("TSG_BN_MASKBITS", "1") != "0"


class _SyncBNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, weight, bias, mod, relu, use_batch_stats, group, hint=None):
        kp = K.provider()
        x, (layout, N, C, HW) = _dense(x)
        if residual is not None:
            residual = _like(residual, x)
        world = _world(group) if use_batch_stats else 1
        count_dev = None
        gamma = weight.float() if weight is not None else None
        beta = bias.float() if bias is not None else None
        if use_batch_stats:
            invstd, fp, count_dev = _batch_statistics(kp, x, layout, N, C, HW, mod, gamma, beta, group, world, hint)
        else:
            mean = mod.running_mean.float()
            invstd = torch.rsqrt(mod.running_var.float() + mod.eps)
            fp = kp.bn_affine(mean, invstd, gamma, beta)
        mixed = (PREFER_CHANNELS_LAST_OUTPUT and residual is None and layout == K.L.NCHW
                 and kp.bn_mixed_supported(x))
        need_y = relu and residual is not None
        bits = None
        if mixed:
            y = kp.bn_apply_fwd_mixed(x, N, C, HW, fp, relu)
        elif (need_y and _MASKBITS and hasattr(kp, "bn_apply_fwd_bits") and kp.bn_maskbits_supported(x, layout, C, HW)
              and residual.data_ptr() % 16 == 0):
            y, bits = kp.bn_apply_fwd_bits(x, residual, layout, N, C, HW, fp)
        else:
            y = kp.bn_apply_fwd(x, residual, layout, N, C, HW, fp, relu)
        ctx.save_for_backward(x, bits if bits is not None else (y if need_y else None), weight, bias, invstd, fp, count_dev)
        ctx.cfg = (layout, N, C, HW, relu, use_batch_stats, group, world, residual is not None, mixed)
        ctx.bits = bits is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        kp = K.provider()
        x, y, weight, bias, invstd, fp, count_dev = ctx.saved_tensors
        layout, N, C, HW, relu, use_batch_stats, group, world, has_res, mixed = ctx.cfg
        if mixed:
            if dy.dtype != x.dtype:
                dy = dy.to(x.dtype)
            dy = dy.contiguous(memory_format=torch.channels_last)
            partial, S = kp.bn_bwd_reduce_mixed(dy, x, N, C, HW, fp, relu)
        else:
            dy = _like(dy, x)
            if ctx.bits and dy.data_ptr() % 16:
                dy = dy.clone(memory_format=torch.preserve_format)           # (a view at an odd offset: the bit-mask kernels take aligned tensors only)
            if ctx.bits:                                          # y holds the one-bit-per-element ReLU mask
                partial, S = kp.bn_bwd_reduce_bits(dy, x, y, layout, N, C, HW, fp)
            else:
                partial, S = kp.bn_bwd_reduce(dy, x, y, layout, N, C, HW, fp, relu)
        dgamma, dbeta, bp = _backward_pack(kp, partial, S, C, N * HW, invstd, fp, count_dev,
                                           use_batch_stats, group, world, x.device, parity=x.dtype == torch.float32)
        if mixed:
            dx, dres = kp.bn_bwd_apply_mixed(dy, x, N, C, HW, bp, relu), None
        elif ctx.bits:
            dx, dres = kp.bn_bwd_apply_bits(dy, x, y, layout, N, C, HW, bp, has_res)
        else:
            dx, dres = kp.bn_bwd_apply(dy, x, y, layout, N, C, HW, bp, relu, has_res)
        if weight is None:
            dgamma = dbeta = None
        else:
            dgamma = dgamma.to(weight.dtype)
            dbeta = dbeta.to(bias.dtype) if bias is not None else None
        return dx, dres, dgamma, dbeta, None, None, None, None, None


class _BnReluPoolFn(torch.autograd.Function):
    """maxpool_3x3/2/1(relu(bn(x))) of the ResNet stem in one pass per direction (csrc/bnpool.hip): neither the
    normalised activation nor its gradient is written.  x channels_last-dense."""

    @staticmethod
    def forward(ctx, x, weight, bias, mod, use_batch_stats, group, hint):
        kp = K.provider()
        layout, N, C, HW = K.bn_layout(x)
        world = _world(group) if use_batch_stats else 1
        count_dev = None
        gamma = weight.float() if weight is not None else None
        beta = bias.float() if bias is not None else None
        if use_batch_stats:
            invstd, fp, count_dev = _batch_statistics(kp, x, layout, N, C, HW, mod, gamma, beta, group, world, hint)
        else:
            mean = mod.running_mean.float()
            invstd = torch.rsqrt(mod.running_var.float() + mod.eps)
            fp = kp.bn_affine(mean, invstd, gamma, beta)
        y, idx = kp.bn_relu_pool_fwd(x, fp)
        ctx.save_for_backward(x, idx, weight, bias, invstd, fp, count_dev)
        ctx.cfg = (N, C, HW, use_batch_stats, group, world)
        return y

    @staticmethod
    def backward(ctx, dpool):
        kp = K.provider()
        x, idx, weight, bias, invstd, fp, count_dev = ctx.saved_tensors
        N, C, HW, use_batch_stats, group, world = ctx.cfg
        if dpool.dtype != x.dtype:
            dpool = dpool.to(x.dtype)
        dpool = dpool.contiguous(memory_format=torch.channels_last)
        partial, S = kp.bn_relu_pool_bwd_reduce(dpool, idx, x, fp)
        dgamma, dbeta, bp = _backward_pack(kp, partial, S, C, N * HW, invstd, fp, count_dev,
                                           use_batch_stats, group, world, x.device, parity=x.dtype == torch.float32)
        dx = kp.bn_relu_pool_bwd_apply(dpool, idx, x, bp)
        if weight is None:
            dgamma = dbeta = None
        else:
            dgamma = dgamma.to(weight.dtype)
            dbeta = dbeta.to(bias.dtype) if bias is not None else None
        return dx, dgamma, dbeta, None, None, None, None


class _StemConvBnReluPoolFn(torch.autograd.Function):
    """maxpool_3x3/2/1(relu(bn(conv7x7/2(img)))) — the whole ResNet stem (resnet.py:96-100,131-133) as ONE node, so that the
    gradient of the 537 MB stem activation never exists: the weight gradient gathers it from the pooled gradient, applies the
    BatchNorm backward and stages it tile by tile (tsg_stem_conv_wrw_bn_pool).  `full`: the activation itself is not stored either — statistics pass, fused forward and backward sums re-evaluate
    the convolution too (csrc/stemconv.hip, round 6; measured slower: the stem convolution is issue-bound, 140 us per
    evaluation).  img bf16 [B,3,H,W] contiguous, no gradient."""

    @staticmethod
    def forward(ctx, img, w_stem, weight, bias, mod, use_batch_stats, group, full):
        kp = K.provider()
        B, _, H, W = img.shape
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        N, C, HW = B, 64, OH * OW
        world = _world(group) if use_batch_stats else 1
        count_dev = None
        gamma = weight.float() if weight is not None else None
        beta = bias.float() if bias is not None else None
        xc = None
        if use_batch_stats:
            if full:
                partial = kp.stem_conv_stats(img, w_stem)
            else:
                xc, partial = kp.stem_conv_fwd_stats(img, w_stem)
            invstd, fp, count_dev = _batch_statistics(kp, img, 1, N, C, HW, mod, gamma, beta, group, world, partial)
        else:
            mean = mod.running_mean.float()
            invstd = torch.rsqrt(mod.running_var.float() + mod.eps)
            fp = kp.bn_affine(mean, invstd, gamma, beta)
            if not full:
                xc = kp.stem_conv_fwd(img, w_stem)
        if full:
            y, idx = kp.stem_conv_bn_relu_pool_fwd(img, w_stem, fp)
        else:
            y, idx = kp.bn_relu_pool_fwd(xc, fp)
        ctx.save_for_backward(img, w_stem, idx, weight, bias, invstd, fp, count_dev, xc)
        ctx.cfg = (N, C, HW, use_batch_stats, group, world)
        ctx.wparam = w_stem
        return y

    @staticmethod
    def backward(ctx, dpool):
        kp = K.provider()
        img, w_stem, idx, weight, bias, invstd, fp, count_dev, xc = ctx.saved_tensors
        N, C, HW, use_batch_stats, group, world = ctx.cfg
        if dpool.dtype != torch.bfloat16:
            dpool = dpool.to(torch.bfloat16)
        dpool = dpool.contiguous(memory_format=torch.channels_last)
        if xc is None:
            partial, S = kp.stem_conv_bn_relu_pool_bwd_reduce(img, w_stem, dpool, idx, fp)
        else:
            partial, S = kp.bn_relu_pool_bwd_reduce(dpool, idx, xc, fp)
        dgamma, dbeta, bp = _backward_pack(kp, partial, S, C, N * HW, invstd, fp, count_dev,
                                           use_batch_stats, group, world, img.device, parity=False)
        dw = None
        if ctx.needs_input_grad[1]:
            from .convwrw import wrw_on_side_stream          # nothing but the optimizer waits for a stem's weight gradient
            # y re-evaluated from the image (xc=None) unless TSG_STEM_WRW_READS_Y=1: reading the 537 MB back measured slower
            # than the 22 MFMAs per tile that remake it (profiles/r06_stem_without_its_gradient.txt)
            yy = xc if _STEM_WRW_READS_Y else None
            dw = wrw_on_side_stream(lambda: kp.stem_conv_wrw_bn_pool(img, w_stem, dpool, idx, bp, xc=yy), ctx.wparam,
                                    img, dpool, idx, bp, yy)
        if weight is None:
            dgamma = dbeta = None
        else:
            dgamma = dgamma.to(weight.dtype)
            dbeta = dbeta.to(bias.dtype) if bias is not None else None
        return None, dw, dgamma, dbeta, None, None, None, None


# TSG_STEM_RECOMPUTE=2|1|0 (default 2): the ResNet stem as the one node above.  2: the stem activation is stored (forward and
# backward sums read it) but its GRADIENT never is; 1: nothing of the stem is stored (every pass re-evaluates the
# convolution: 1.05 ms against 0.94 ms, kept as evidence); 0: the three modules with their own nodes.
_STEM_RECOMPUTE = int(os.environ.get("TSG_STEM_RECOMPUTE", "2"))
_STEM_WRW_READS_Y = os.environ.get("TSG_STEM_WRW_READS_Y", "0") == "1"


def stem_bn_relu_maxpool(conv, bn, img, pool):
    """`pool(relu(bn(conv(img))))` (furnace/base_model/resnet.py:96-100,131-133) as the recomputing node when `conv` is our
    3 -> 64 7x7/2 StemConv2d on the bf16 path, `bn` our SyncBatchNorm(64), `pool` MaxPool2d(3, 2, 1) and `img` an image that
    needs no gradient; None otherwise (the caller then runs the modules one by one)."""
    from .stemconv import StemConv2d, _as_bf16_image, _wants_bf16

    def one(v):
        return v[0] if isinstance(v, (tuple, list)) and len(set(v)) == 1 else v
    if not (_STEM_RECOMPUTE > 0 and _FUSE_STEM_POOL and isinstance(conv, StemConv2d) and isinstance(bn, SyncBatchNorm)
            and isinstance(pool, torch.nn.MaxPool2d) and isinstance(img, torch.Tensor) and img.is_cuda and img.dim() == 4
            and not img.requires_grad and conv.bias is None and conv.weight.dtype == torch.float32
            and conv.padding_mode == "zeros" and bn.num_features == 64 and bn.momentum is not None
            and _wants_bf16(img)
            and not (conv._forward_hooks or conv._forward_pre_hooks or bn._forward_hooks or bn._forward_pre_hooks
                     or pool._forward_hooks or pool._forward_pre_hooks)):
        return None
    if (one(pool.kernel_size), one(pool.stride), one(pool.padding), one(pool.dilation)) != (3, 2, 1, 1) \
            or pool.ceil_mode or pool.return_indices:
        return None
    kp = K.provider()
    if not hasattr(kp, "stem_conv_bn_relu_pool_fwd"):
        return None
    xb = _as_bf16_image(img)
    w_stem = conv.weight if conv.weight.is_contiguous() else conv.weight.contiguous()
    if not kp.stem_conv_supported(xb, w_stem, conv.stride[0], conv.padding[0], conv.dilation[0], conv.groups):
        return None
    if ((xb.shape[2] - 1) // 2 + 1) < 1 or ((xb.shape[3] - 1) // 2 + 1) < 1:
        return None
    use_batch_stats = bn.training or not bn.track_running_stats
    with torch.autocast("cuda", enabled=False):
        return _StemConvBnReluPoolFn.apply(xb, w_stem, bn.weight, bn.bias, bn, use_batch_stats, bn.process_group,
                                           _STEM_RECOMPUTE == 1)


_FUSE_STEM_POOL = os.environ.get("TSG_FUSE_STEM_POOL", "1") != "0"


def bn_relu_maxpool(bn, x, pool):
    """`pool(relu(bn(x)))` (furnace/base_model/resnet.py:98-100,131-133).  One fused pass per direction when `bn` is
    our SyncBatchNorm, `pool` is MaxPool2d(3, 2, 1) and x is a channels_last HIP activation; None otherwise (the
    caller then runs the three modules)."""
    def one(v):
        return v[0] if isinstance(v, (tuple, list)) and len(set(v)) == 1 else v
    if not (_FUSE_STEM_POOL and isinstance(bn, SyncBatchNorm) and isinstance(pool, torch.nn.MaxPool2d)
            and isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4
            and x.dtype in (torch.float32, torch.bfloat16)):
        return None
    if (one(pool.kernel_size), one(pool.stride), one(pool.padding), one(pool.dilation)) != (3, 2, 1, 1) \
            or pool.ceil_mode or pool.return_indices or bn.momentum is None:
        return None
    vec = 8 if x.dtype == torch.bfloat16 else 4
    if x.shape[1] % vec or x.is_contiguous() or not x.is_contiguous(memory_format=torch.channels_last):
        return None
    bn._check_input_dim(x)
    use_batch_stats = bn.training or not bn.track_running_stats
    from .stemconv import take_bn_partial
    hint = take_bn_partial(x) if use_batch_stats else None
    return _BnReluPoolFn.apply(x, bn.weight, bn.bias, bn, use_batch_stats, bn.process_group, hint)


class SyncBatchNorm(_BatchNorm):
    """Drop-in for apex.parallel.SyncBatchNorm / torch.nn.BatchNorm{1,2,3}d.

    Constructor as the reference calls it: norm_layer(planes, eps=, momentum=)
    (furnace/base_model/resnet.py:24,28), norm_layer(planes, eps=)
    (seg_oprs.py:34), norm_layer(channels) (seg_oprs.py:85).  `.eps`,
    `.momentum`, `.weight`, `.bias` stay writable (utils/init_func.py:16-21) and
    the buffers keep torch's names so ImageNet checkpoints load.
    """

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True,
                 track_running_stats=True, process_group=None, channel_last=False,
                 fuse_relu=False):
        super().__init__(num_features, eps=eps, momentum=momentum, affine=affine,
                         track_running_stats=track_running_stats)
        self.process_group = process_group
        self.channel_last = channel_last
        self.fuse_relu = fuse_relu

    def _check_input_dim(self, input):
        if input.dim() < 2:
            raise ValueError("expected at least 2D input (got {}D input)".format(input.dim()))
        if input.shape[1] != self.num_features:
            raise ValueError("expected {} channels, got {}".format(self.num_features, input.shape[1]))

    def forward(self, input, residual=None, relu=None):
        self._check_input_dim(input)
        if self.momentum is None:
            raise NotImplementedError("cumulative moving average (momentum=None) is not supported")
        relu = self.fuse_relu if relu is None else bool(relu)
        use_batch_stats = self.training or not self.track_running_stats
        hint = None
        if use_batch_stats and hasattr(input, "_tsg_bn_partial"):
            from .stemconv import take_bn_partial
            hint = take_bn_partial(input)
        return _SyncBNFn.apply(input, residual, self.weight, self.bias, self, relu,
                               use_batch_stats, self.process_group, hint)


def convert_syncbn_model(module, process_group=None, channel_last=False):
    """apex.parallel.convert_syncbn_model: swap torch BatchNorm layers for ours."""
    mod = module
    if isinstance(module, torch.nn.modules.batchnorm._BatchNorm) and not isinstance(module, SyncBatchNorm):
        mod = SyncBatchNorm(module.num_features, module.eps, module.momentum, module.affine,
                            module.track_running_stats, process_group, channel_last)
        if module.affine:
            mod.weight = module.weight
            mod.bias = module.bias
        if module.track_running_stats:
            mod.running_mean = module.running_mean
            mod.running_var = module.running_var
            mod.num_batches_tracked = module.num_batches_tracked
    for name, child in module.named_children():
        mod.add_module(name, convert_syncbn_model(child, process_group, channel_last))
    return mod
