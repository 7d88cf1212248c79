#!/usr/bin/env python
"""Headline benchmark: training images/sec of BiSeNet-R18 on 1024x1024 synthetic
Cityscapes-shaped crops (BASELINE.json `metric`, configs[1]): bf16 activations,
per-GPU batch 16, SyncBN + OHEM, SGD with the reference's 14 parameter groups.

    python bench.py [--gpus N --steps K --warmup W] [--config bisenet|pspnet|dfn|psanet]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

--config selects the BASELINE.json configuration at its PER-RANK shape (default bisenet = configs[1], the one the
metric is quoted on): pspnet = configs[2] PSPNet-R50_v1c 2 x 720^2, 150 classes (713 is not a legal crop);
dfn = configs[3] DFN-R101_v1c 2 x 1024^2, 4 CE + 4 focal heads; psanet = configs[4] PSANet-R101_v1c 2 x 480^2 (473 is
not legal).  Same JSON schema for every config.

A step = zero_grad -> loss = model(imgs, gts) -> backward (incl. bucketed RCCL
gradient all-reduce) -> SGD step, i.e. the body of the reference's loop
(model/bisenet/cityscapes.bisenet.R18/train.py:115-142) without its tqdm/.item()
display.  Inputs are resident in HBM before the timed region.  --scaling weak
(default): every rank keeps batch 16.  --scaling strong: the reference's own setting, a
GLOBAL batch of 16 split over the ranks (dataloader.py:51-53 batch_size // world_size,
min_kept scaled with it, train.py:48-49).  Rank 0 prints ONE JSON line.
"""
import argparse
import contextlib
import json
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.nn as nn

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NUM_CLASSES = 19
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable copy)


CONFIGS = {
    # name: (per-rank batch, crop, classes, BASELINE.json configs index, reference experiment directory)
    "bisenet": dict(batch=16, size=1024, classes=19, idx=1, model="BiSeNet-R18", ref="model/bisenet/cityscapes.bisenet.R18"),
    "pspnet": dict(batch=2, size=720, classes=150, idx=2, model="PSPNet-R50_v1c", ref="model/pspnet/ade.pspnet.R50_v1c"),
    "dfn": dict(batch=2, size=1024, classes=19, idx=3, model="DFN-R101_v1c", ref="model/dfn/cityscapes.dfn.R101_v1c"),
    "psanet": dict(batch=2, size=480, classes=150, idx=4, model="PSANet-R101_v1c", ref="model/psanet/ade.psanet.R101_v1c"),
}


def _optimizer(groups, lr, wd, fused_sgd):
    if fused_sgd:
        from torchseg_amd.optim import FusedSGD                        # same update, one HIP launch for all tensors
        return FusedSGD(groups, lr=lr, momentum=0.9, weight_decay=wd)
    return torch.optim.SGD(groups, lr=lr, momentum=0.9, weight_decay=wd)


_REF_NET = {}


def reference_network_module(config):
    """The reference's UNCHANGED network.py of a BASELINE config, imported against our furnace/ (tools/stage_reference.py:
    from the checkout where it exists, on the GPU box from the archive the build container packed under oracle/_ref/).
    One experiment per process (the module is called `network` whatever the family)."""
    if config not in _REF_NET:
        if _REF_NET:
            raise RuntimeError("one reference experiment per process")
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import stage_reference
        family, exp = CONFIGS[config]["ref"].split("/")[1:]
        _REF_NET[config] = stage_reference.import_experiment(family, exp)[0]
    return _REF_NET[config]


def build_model(device, batch, size, criterion_cls, norm_layer, seed=12345, fused_sgd=False, config="bisenet",
                focal_cls=None, dropout=True, network="native"):
    """Model + optimizer of one BASELINE config exactly as the reference's train.py builds them (bisenet train.py:48-89;
    pspnet / psanet train.py:48-80; dfn train.py:48-78).  `criterion_cls` is the OHEM criterion class for bisenet (ours
    on the GPU, the oracle's on the CPU); the other families use nn.CrossEntropyLoss as the reference does, DFN's
    border heads `focal_cls` (ours / the oracle's SigmoidFocalLoss)."""
    from torchseg_amd.workloads import ensure_furnace_on_path
    ensure_furnace_on_path()
    from utils.init_func import group_weight, init_weight
    torch.manual_seed(seed)
    groups = []
    ref_net = reference_network_module(config) if network == "reference" else None
    if config == "bisenet":
        if ref_net is not None:
            BiSeNet = ref_net.BiSeNet                                  # network.py:18 as it is
        else:
            from torchseg_amd.workloads.bisenet import BiSeNet
        min_kept = int(batch * size * size // 16)                      # train.py:48-49
        criterion = criterion_cls(ignore_label=255, thresh=0.7, min_kept=min_kept, use_weight=False)
        model = BiSeNet(NUM_CLASSES, is_training=True, criterion=criterion, pretrained_model=None,
                        norm_layer=norm_layer)
        init_weight(model.business_layer, nn.init.kaiming_normal_, norm_layer, 1e-5, 0.1,
                    mode='fan_in', nonlinearity='relu')                # train.py:61-63
        base_lr, wd = 1e-2, 5e-4
        groups = group_weight(groups, model.context_path, norm_layer, base_lr)
        for part in (model.spatial_path, model.global_context, model.arms, model.refines, model.heads, model.ffm):
            groups = group_weight(groups, part, norm_layer, base_lr * 10)  # train.py:70-84
    else:
        if config == "dfn":
            if ref_net is not None:
                DFN = ref_net.DFN
            else:
                from torchseg_amd.workloads.dfn import DFN
            model = DFN(19, nn.CrossEntropyLoss(reduction='mean', ignore_index=255), focal_cls(255, 2.0, 0.25), 0.1,
                        None, norm_layer)                              # dfn train.py:48-58, config.py:76
            base_lr, wd = 7e-4, 1e-4                                   # dfn config.py:79-82
        else:
            crit = nn.CrossEntropyLoss(reduction='mean', ignore_index=-1)
            if ref_net is not None:                                    # both families call their class PSPNet (network.py:17)
                model = ref_net.PSPNet(150, crit, None, norm_layer)
            else:
                from torchseg_amd.workloads.pspnet import PSANet, PSPNet
                cls, depth = (PSPNet, 50) if config == "pspnet" else (PSANet, 101)
                model = cls(150, crit, None, norm_layer, depth=depth)
            base_lr, wd = 1e-2, 1e-4                                   # pspnet / psanet config.py:76-79 / 80-83
        if not dropout:
            for m in model.modules():
                if isinstance(m, nn.Dropout2d):
                    m.p = 0.0                                          # parity runs: CPU and GPU RNG streams differ
        init_weight(model.business_layer, nn.init.kaiming_normal_, norm_layer, 1e-5, 0.1,
                    mode='fan_in', nonlinearity='relu')
        groups = group_weight(groups, model.backbone, norm_layer, base_lr)
        for part in model.business_layer:
            groups = group_weight(groups, part, norm_layer, base_lr * 10)
    model.to(device)
    return model, _optimizer(groups, base_lr, wd, fused_sgd), base_lr


def synthetic_batch(device, batch, size, seed=0, config="bisenet", label_dtype=torch.int64):
    """x ~ N(0,1), labels uniform over the classes with rows 0-7 ignored.  `label_dtype` uint8 is what the
    GPU loader (torchseg_amd.data) emits for the 19-class / ignore-255 families: the criteria read 1 B instead of
    8 B per pixel and head (the CPU oracle indexes with int64)."""
    g = torch.Generator(device=device).manual_seed(seed)
    classes = CONFIGS[config]["classes"]
    imgs = torch.randn(batch, 3, size, size, generator=g, device=device)
    gts = torch.randint(0, classes, (batch, size, size), generator=g, device=device)
    if config in ("pspnet", "psanet"):
        gts[:, :8] = -1                                                # ADE: ignore_index -1 (train.py:48-49)
        return imgs, gts
    gts[:, :8] = 255
    gts = gts.to(label_dtype)
    if config == "dfn":                                                # border labels in {0, 1, 255} (dfn dataloader.py:24-29)
        edge = torch.randint(0, 2, (batch, size, size), generator=g, device=device)
        edge[:, :, :8] = 255
        return imgs, gts, edge.to(label_dtype)
    return imgs, gts


def set_lr(opt, lr_policy, it):
    lr = lr_policy.get_lr(it)
    for i, gparam in enumerate(opt.param_groups):
        gparam['lr'] = lr if i < 2 else lr * 10                        # train.py:133-139
    if hasattr(opt, "refresh_lr"):
        opt.refresh_lr()                                               # device-side lr for graph replay


@contextlib.contextmanager
def serial_kernels(on=True):
    """Steps whose kernels are bracketed with HIP events run with the weight gradients on the COMPUTE stream: beside the
    SyncBatchNorm passes on their side stream (convwrw.wrw_on_side_stream, the default) every bracketed kernel's elapsed time
    would contain the kernel it shares the chip with, and the per-kernel roofline figures would be those of a time-shared
    GPU (round 5: SyncBN 0.60 -> 0.47 of the HBM peak, weight gradients 0.28 -> 0.15 of the MFMA peak, their sum > the step)."""
    from torchseg_amd import convwrw
    from torchseg_amd.workloads import bisenet as _wb
    old, old_heads, old_sp = convwrw._WRW_STREAM, _wb._FORK_HEADS, _wb._FORK_SPATIAL
    if on:
        convwrw._WRW_STREAM = False
        _wb._FORK_HEADS = False                        # (round 6) the auxiliary heads' and the detail branch's side streams likewise
        _wb._FORK_SPATIAL = False
    try:
        yield
    finally:
        convwrw._WRW_STREAM = old
        _wb._FORK_HEADS = old_heads
        _wb._FORK_SPATIAL = old_sp


def step_body(model, opt, batch, world, with_optimizer=True):
    from utils.pyt_utils import all_reduce_tensor
    opt.zero_grad()
    loss = model(*batch)
    if world > 1 or dist.is_initialized():
        all_reduce_tensor(loss, world_size=world)                      # train.py:129-131
    loss.backward()
    if with_optimizer:
        opt.step()
    return loss


def train_step(model, opt, batch, lr_policy, it, world):
    set_lr(opt, lr_policy, it)
    return step_body(model, opt, batch, world)


class GraphedStep(object):
    """zero_grad -> forward -> backward (incl. collectives) captured once into a hipGraph and replayed, which removes
    ~650 host-side launches per step.  With opt_inside (FusedSGD: its learning rates live in a device vector,
    `refresh_lr`) the optimizer step is part of the graph too; otherwise it runs eagerly after each replay, so the
    reference's per-iteration lr schedule (train.py:133-139) needs no special handling.

    Warm-up, capture, replays and the eager optimizer MUST all run on one stream (`GraphedStep.stream`; bench.py wraps
    the whole graphed section in it).  Measured on ROCm 7.2 at the bench shape (tools/debug_graph4.py):
    whenever the eager warm-up ran on a stream other than the one the replays and the optimizer later use -- the
    side-stream warm-up of PyTorch's CUDA-graphs recipe included, even with only the last warm-up step elsewhere -- the
    SECOND replay produces non-finite output in the MIOpen 1x1 convolution of the global-context branch, then NaN
    gradients everywhere; with one stream throughout (the default stream or any other) the replayed trajectory equals
    the eager one.  State left behind by the eager warm-up in the convolution library is bound to the stream it ran on."""

    stream = None

    @classmethod
    def capture_stream(cls):
        if cls.stream is None:
            cls.stream = torch.cuda.Stream()
        return cls.stream

    def __init__(self, model, opt, batch, world, opt_inside=False):
        self.graph = torch.cuda.CUDAGraph()
        self.opt = opt
        self.opt_inside = bool(opt_inside)
        opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph, stream=self.capture_stream()):
            self.loss = step_body(model, opt, batch, world, with_optimizer=self.opt_inside)

    def __call__(self):
        self.graph.replay()
        if not self.opt_inside:
            self.opt.step()
        return self.loss


class SegmentedStep(object):
    """The same step as FIFTEEN linear hipGraphs instead of one, launched on three streams so that what eager launches overlap
    overlaps under replay too (a captured graph with parallel BRANCHES is replayed node by node by the runtime and is slower than
    the single-stream graph; linear graphs on different streams overlap like eager launches):
        s0: stem + layer1 | context tail, first stage | second stage | fusion + main head (fwd+loss+bwd) + fusion bwd | layer2..4 bwd | layer1 bwd | stem bwd | optimizer
        s1:               | detail branch (SpatialPath) ........... | aux head 1 .............................. | detail-branch backward ......................... |
        s2:                                         | aux head 0 ................................................... | heads' weight grads | layer2..4 weight grads | layer1 weight grads |
    The autograd graph is cut at the heads' and the fusion module's inputs (detached leaves whose gradients are handed to
    `torch.autograd.backward` of the segment in front): every kernel and every operand is the one the one-graph step runs, so
    losses and gradients are the same bit for bit.  Our own BiSeNet builder only (context_head / context_tail_first / _second / heads)."""

    @staticmethod
    def applies(model, world):
        net = getattr(model, "module", model)
        return (world == 1 and not dist.is_initialized() and hasattr(net, "context_tail") and hasattr(net, "heads")
                and len(getattr(net, "heads", ())) == 3 and getattr(net, "is_training", False)
                and not getattr(model, "fuse_chain", False))

    def __init__(self, model, opt, batch):
        from torchseg_amd.workloads import bisenet as wb
        net = model.module
        data, label = batch
        dev = data.device
        self.opt = opt
        s0 = self.s0 = GraphedStep.capture_stream()
        s1, s2 = self.s1, self.s2 = wb._side_stream(dev, 1), wb._side_stream(dev, 2)
        G = torch.cuda.CUDAGraph
        self.g = {k: G() for k in ("a1", "sp", "a2", "a2b", "ffm", "h0", "h1", "hm", "bffm", "wh", "bsp", "bctx_a", "wa", "opt_a", "bctx_b", "wb", "bctx_c", "opt")}
        g = self.g
        p0, p1, p2 = (torch.cuda.graph_pool_handle() for _ in range(3))             # one memory pool per stream's graphs
        from torchseg_amd import convwrw
        cdt = getattr(model, "compute_dtype", torch.bfloat16)
        ac = lambda: torch.autocast("cuda", dtype=cdt)
        leaf = lambda t: t.detach().requires_grad_(True)
        from torchseg_amd import kernels as K
        K.provider().presize_scratch((s1, s2), dev)      # no scratch buffer may move between two captures on one stream
        torch.cuda.synchronize()
        opt.zero_grad(set_to_none=True)
        cp = net.context_path
        self.early_opt = os.environ.get("TSG_SEG_EARLY_OPT", "0") == "1" and hasattr(opt, "prepare")
        late_ids = {id(p) for m in ([m for n, m in cp.named_children() if n not in ("layer2", "layer3", "layer4")]
                                    + [net.spatial_path]) for p in m.parameters()}
        params = [p for grp in opt.param_groups for p in grp["params"]]
        early, late = [p for p in params if id(p) not in late_ids], [p for p in params if id(p) in late_ids]
        if self.early_opt:                               # (see below) block maps / shadow tables: not under capture
            opt.prepare(only=early)
            opt.prepare(only=late)
            torch.cuda.synchronize()
        with torch.cuda.graph(g["a1"], pool=p0, stream=s0):
            opt.zero_grad()
            with ac():
                x1 = cp._stem(data)
                x1_l = leaf(x1)                          # cuts of the context backward: stem | layer1 | layer2 .. (below)
                c2 = cp.layer1(x1_l)
                c2_l = leaf(c2)
        s1.wait_stream(s0)
        with torch.cuda.graph(g["sp"], pool=p1, stream=s1):
            with ac():
                sp = net.spatial_path(data)
        # TSG_SEG_EARLY_HEADS=1|0 (default 1): an auxiliary head starts as soon as ITS feature map exists — head 0 on s2 behind
        # the first attention-refinement stage (beside the second stage and the fusion module), head 1 on s1 behind the second
        # (beside the fusion module) — instead of all three behind the fusion module: the fused head kernels are VALU-bound
        # and three of them side by side contend for the same unit, next to matrix-core / HBM-bound kernels they do not
        # (12.01 -> 11.97 ms in six interleaved pairs, profiles/r06_segmented_early_heads_early_optimizer_ab.txt: - 0.3 %, bit-equal trajectory)
        self.early_heads = os.environ.get("TSG_SEG_EARLY_HEADS", "1") != "0"
        losses = [None, None, None]

        # TSG_SEG_DEFER_HEADS=1|0: the 3x3 weight gradients of the three heads are not launched inside the heads' graphs (the
        # main head's sits on s0's critical path, the auxiliary heads' hold back the join in front of the context backward)
        # but listed (convwrw._DEFER) and replayed as a graph of their own (wh) on s2 beside the first part of the context
        # backward, where s2 has nothing else to do
        self.defer_heads = os.environ.get("TSG_SEG_DEFER_HEADS", "1") != "0"
        d_heads = []

        @contextlib.contextmanager
        def deferring():
            if not self.defer_heads:
                yield
                return
            convwrw._DEFER = []
            try:
                yield
            finally:
                d_heads.extend(convwrw._DEFER)
                convwrw._DEFER = None

        def aux_head(i, side, key, pool, fm_leaf):
            side.wait_stream(s0)
            with deferring(), torch.cuda.graph(g[key], pool=pool, stream=side):
                with ac():
                    losses[i] = net.criterion(net.heads[i](fm_leaf), label)
                losses[i].backward()

        if self.early_heads:
            with torch.cuda.graph(g["a2"], pool=p0, stream=s0):
                with ac():
                    f16, c3, c4 = net.context_tail_first(c2_l)
            l16 = leaf(f16)
            aux_head(0, s2, "h0", p2, l16)
            with torch.cuda.graph(g["a2b"], pool=p0, stream=s0):
                with ac():
                    f8 = net.context_tail_second(f16, c3, c4)
            l8 = leaf(f8)
            s1.wait_stream(s0)
            s0.wait_stream(s1)
            aux_head(1, s1, "h1", p1, l8)
        else:
            with torch.cuda.graph(g["a2"], pool=p0, stream=s0):
                with ac():
                    f16, f8 = net.context_tail(c2_l)
            s0.wait_stream(s1)
        # fusion module, main head and the fusion module's backward: consecutive on s0 with no other stream waiting in
        # between — ONE graph (a graph boundary on the critical path is 7-30 us with no kernel on any queue,
        # profiles/r06_segmented_replay_timeline.txt); TSG_SEG_MERGE=0 keeps the three apart
        self.merge = os.environ.get("TSG_SEG_MERGE", "1") != "0"
        self.one = self.merge and self.early_heads       # (without the early heads the auxiliary heads wait for the fusion module)

        def fusion_forward():
            sp_l, f8_ffm = leaf(sp), leaf(f8)
            with ac():
                fused = net.ffm(sp_l, f8_ffm)
            leaves = [l16, l8, leaf(fused)] if self.early_heads else [leaf(f16), leaf(f8), leaf(fused)]
            return sp_l, f8_ffm, fused, leaves

        def main_head(fused, leaves):
            with ac():
                losses[2] = net.criterion(net.heads[-1](leaves[2]), label)
            losses[2].backward()
            if self.merge:
                torch.autograd.backward([fused], [leaves[2].grad])

        if self.one:
            with deferring(), torch.cuda.graph(g["hm"], pool=p0, stream=s0):
                sp_l, f8_ffm, fused, leaves = fusion_forward()
                main_head(fused, leaves)
        else:
            with torch.cuda.graph(g["ffm"], pool=p0, stream=s0):
                sp_l, f8_ffm, fused, leaves = fusion_forward()
            if not self.early_heads:
                aux_head(0, s1, "h0", p1, leaves[0])
                aux_head(1, s2, "h1", p2, leaves[1])
            with deferring(), torch.cuda.graph(g["hm"], pool=p0, stream=s0):
                main_head(fused, leaves)
        if not self.merge:
            with torch.cuda.graph(g["bffm"], pool=p0, stream=s0):
                torch.autograd.backward([fused], [leaves[2].grad])
        s0.wait_stream(s1)
        s0.wait_stream(s2)
        s1.wait_stream(s0)
        self.wh = bool(d_heads)
        if self.wh:
            s2.wait_stream(s0)
            with torch.cuda.graph(g["wh"], pool=p2, stream=s2):
                for fn, _ops, _buf in d_heads:
                    fn()
        with torch.cuda.graph(g["bsp"], pool=p1, stream=s1):
            torch.autograd.backward([sp], [sp_l.grad])
        # Context backward in three parts: layer2 .. layer4 + attention refinement | layer1 | stem.  The 3x3 weight gradients of
        # the first two parts are not launched in place: convwrw hands autograd their result tensors and lists the launches,
        # which become graphs of their own (wa, wb) replayed on s2 beside the NEXT part — matrix-core-bound weight gradients
        # beside the HBM-bound passes over the large maps of layer1 and the stem.
        def deferred_backward(key, roots, grads):
            convwrw._DEFER = []
            try:
                with torch.cuda.graph(g[key], pool=p0, stream=s0):
                    torch.autograd.backward(roots, grads)
                return convwrw._DEFER
            finally:
                convwrw._DEFER = None

        def launches(key, lst):
            s2.wait_stream(s0)
            with torch.cuda.graph(g[key], pool=p2, stream=s2):
                for fn, _ops, _buf in lst:
                    fn()

        d_a = deferred_backward("bctx_a", [f16, f8, f8], [leaves[0].grad, leaves[1].grad, f8_ffm.grad])
        launches("wa", d_a)
        # TSG_SEG_EARLY_OPT=1 (default 0): the optimizer in two parts — behind wa every gradient but those of the context
        # path's stem and layer1 and of the detail branch is final, and nothing that still runs reads those parameters or
        # their bf16 shadows, so their update (97 % of the elements: the two HBM-bound launches that close the step alone)
        # can go on s2 beside the layer1 / stem backward.  Bit-equal trajectory, and measured NEUTRAL (12.02 vs 12.01 ms in
        # three interleaved pairs, profiles/r06_segmented_early_heads_early_optimizer_ab.txt): the update's 0.5 GB beside the HBM-bound layer1 passes slows
        # those by what it saves at the end.  Left opt-in.
        if self.early_opt:
            with torch.cuda.graph(g["opt_a"], pool=p2, stream=s2):
                opt.step(only=early)
        d_b = deferred_backward("bctx_b", [c2], [c2_l.grad])
        launches("wb", d_b)
        with torch.cuda.graph(g["bctx_c"], pool=p0, stream=s0):
            torch.autograd.backward([x1], [x1_l.grad])
        s0.wait_stream(s1)
        s0.wait_stream(s2)
        with torch.cuda.graph(g["opt"], pool=p0, stream=s0):
            self.loss = losses[2].detach() + losses[0].detach() + losses[1].detach()
            if self.early_opt:
                opt.step(only=late)
            else:
                opt.step()
        self.keep = [x1, x1_l, c2, c2_l, sp, f16, f8, sp_l, f8_ffm, fused, leaves, losses, d_a, d_b, d_heads]   # d_a / d_b / d_heads: operands read by wa / wb / wh

    def __call__(self):
        g, s0, s1, s2 = self.g, self.s0, self.s1, self.s2
        cs = torch.cuda.stream
        with cs(s0):
            g["a1"].replay()
        s1.wait_stream(s0)
        with cs(s1):
            g["sp"].replay()
        if self.early_heads:
            with cs(s0):
                g["a2"].replay()
            s2.wait_stream(s0)
            with cs(s2):
                g["h0"].replay()
            with cs(s0):
                g["a2b"].replay()
            s0.wait_stream(s1)                           # the detail branch (before head 1 is queued behind it on s1)
            s1.wait_stream(s0)
            with cs(s1):
                g["h1"].replay()
            if not self.one:
                with cs(s0):
                    g["ffm"].replay()
        else:
            with cs(s0):
                g["a2"].replay()
                s0.wait_stream(s1)
                g["ffm"].replay()
            s1.wait_stream(s0)
            s2.wait_stream(s0)
            with cs(s1):
                g["h0"].replay()
            with cs(s2):
                g["h1"].replay()
        with cs(s0):
            g["hm"].replay()
            if not self.merge:
                g["bffm"].replay()
            s0.wait_stream(s1)                           # aux head 0's gradient (and s1 is free for the detail branch's backward)
            s0.wait_stream(s2)
        s1.wait_stream(s0)
        if self.wh:
            s2.wait_stream(s0)
            with cs(s2):
                g["wh"].replay()
        with cs(s1):
            g["bsp"].replay()
        with cs(s0):
            g["bctx_a"].replay()
        s2.wait_stream(s0)
        with cs(s2):
            g["wa"].replay()
            if self.early_opt:
                g["opt_a"].replay()
        with cs(s0):
            g["bctx_b"].replay()
        s2.wait_stream(s0)
        with cs(s2):
            g["wb"].replay()
        with cs(s0):
            g["bctx_c"].replay()
            s0.wait_stream(s1)
            s0.wait_stream(s2)
            g["opt"].replay()
        return self.loss


def _cpu_leg(size, batch, cores, budget_s, max_steps, config="bisenet", warm=True):
    """img/s of the oracle network (1 untimed warm-up step, then steps until `budget_s` or `max_steps`)."""
    from oracle.focal_ref import SigmoidFocalLoss as OracleFocal
    from oracle.ohem_ref import ProbOhemCrossEntropy2d as OracleOhem
    from engine.lr_policy import PolyLR
    dev = torch.device("cpu")
    model, opt, base_lr = build_model(dev, batch, size, OracleOhem, nn.BatchNorm2d, config=config, focal_cls=OracleFocal)
    model.train()
    data = synthetic_batch(dev, batch, size, config=config)
    pol = PolyLR(base_lr, 0.9, 80000)
    if warm:
        train_step(model, opt, data, pol, 0, 1)                        # warm-up (thread pool, allocator)
    t0 = time.perf_counter()
    done = 0
    while done < max_steps and (done < 1 or time.perf_counter() - t0 < budget_s):
        train_step(model, opt, data, pol, done + 1, 1)
        done += 1
    dt = time.perf_counter() - t0
    return round(batch * done / dt, 3), done


def cpu_baseline_family(config):
    """cpu_baseline for --config pspnet|dfn|psanet: the same oracle network (nn.BatchNorm2d, nn.CrossEntropyLoss, the
    loss_opr.py focal restatement) at the bench shape: ONE timed step, no warm-up step (a step of these networks is
    20-90 s of host time, against which thread-pool start-up is noise)."""
    from torchseg_amd.workloads import ensure_furnace_on_path
    ensure_furnace_on_path()
    cfg = CONFIGS[config]
    ncpu = os.cpu_count() or 1
    cores = min(ncpu, 64)
    torch.set_num_threads(cores)
    v, n = _cpu_leg(cfg["size"], cfg["batch"], cores, 0.0, 1, config, warm=False)
    return {"value": v, "unit": "img/s", "cores": cores, "host_cpus": ncpu, "kind": "port",
            "sample": f"{n} step (no warm-up step at this size) of batch {cfg['batch']} at {cfg['size']}x{cfg['size']}, fp32, "
                      f"torch CPU, oracle {cfg['model']} (nn.BatchNorm2d + nn.CrossEntropyLoss"
                      + (" + loss_opr.py focal restatement" if config == "dfn" else "") + " + torch.optim.SGD)"}


def ohem_kth_branch_probe(device, batch, size, reps=5):
    """One BiSeNet head (fused upsample -> OHEM, the path the step takes) on TRAINED-LIKE logits in the spirit of SURVEY
    8(d): 8 * onehot(label') + N(0,1) at 1/8 resolution, generator seed 1.  For the k-th-value branch of
    loss_opr.py:84-90 fewer than min_kept = P/16 pixels may have p_target <= 0.7, so the labels are constant over 512 x
    512 regions (the bilinear up-sampling blurs the logits along region borders: ~3 % of the pixels) and label' re-draws
    1 % of the low-resolution pixels (SURVEY's 10 % at full resolution would leave > P/16 hard pixels and the threshold
    branch).  Random-init logits (the timed step) give p_target ~ 1/19 << 0.7, so every step takes the threshold branch and the
    k-th-value machinery (radix select over 16.8 M probabilities: sel_refine, ohem_pass_c) exits early; the reference
    would run its full torch.sort there all the same (loss_opr.py:84-90).  This record puts the cost of the k-th branch
    in front of the driver: forward / forward+backward time per head in both regimes, and what the selection did."""
    from torchseg_amd.losses import ohem_cross_entropy
    from torchseg_amd.upsample import DeferredUpsample
    g = torch.Generator(device=device).manual_seed(1)
    low = size // 8
    cells = max(low // 64, 1)
    lab_low = torch.randint(0, NUM_CLASSES, (batch, cells, cells), generator=g, device=device)
    lab_low = lab_low.repeat_interleave(low // cells, 1).repeat_interleave(low // cells, 2)
    target = lab_low.repeat_interleave(8, 1).repeat_interleave(8, 2)
    redraw = torch.rand(batch, low, low, generator=g, device=device) < 0.01
    lab2 = torch.where(redraw, torch.randint(0, NUM_CLASSES, (batch, low, low), generator=g, device=device), lab_low)
    z_tr = (8.0 * torch.nn.functional.one_hot(lab2, NUM_CLASSES).permute(0, 3, 1, 2).float()
            + torch.randn(batch, NUM_CLASSES, low, low, generator=g, device=device)).to(torch.bfloat16)
    z_rand = torch.randn(batch, NUM_CLASSES, low, low, generator=g, device=device).to(torch.bfloat16)
    target[:, :8] = 255
    target = target.to(torch.uint8)
    k = batch * size * size // 16
    out = {"shape": f"{batch}x{NUM_CLASSES}x{low}^2 -> {size}^2 bf16, uint8 labels, min_kept {k}, thresh 0.7"}

    def timed(fn):
        """median over `reps` of (events around 2 calls) / 2: one host hiccup (a collection of the models the side runs
        just freed cost one repetition 7 ms in round 5) must not become the figure"""
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            fn()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 2)
        ts.sort()
        return round(ts[len(ts) // 2] * 1e3, 1)

    for name, z0 in (("trained_like", z_tr), ("random_init", z_rand)):
        z = z0.clone().requires_grad_(True)

        def fwd():
            return ohem_cross_entropy(DeferredUpsample(z, (size, size)), target, 255, 0.7, k, return_selection=True)

        def fb():
            z.grad = None
            fwd()[0].backward()
        with torch.no_grad():
            t_f = timed(fwd)
        t_fb = timed(fb)
        loss, sel = fwd()
        sel = sel.cpu()
        thr = float(sel[0:1].view(torch.float32).item())
        out[name] = {"fwd_us": t_f, "fwd_bwd_us": t_fb, "threshold": round(thr, 6), "kept": int(sel[1]),
                     "valid": int(sel[2]), "kth_branch_taken": bool(thr > 0.7 + 1e-7), "loss": round(float(loss), 5)}
    out["selection_tail_us"] = round(out["trained_like"]["fwd_us"] - out["random_init"]["fwd_us"], 1)
    return out


def ohem_kth_branch_oracle(device, batch, size):
    """The oracle's selection (oracle.ohem_ref = loss_opr.py:68-98 on the host) on the SAME trained-like head input the
    probe above times: threshold / kept / loss to print beside the device's.  ~15 s of host time at 16 x 19 x 1024^2."""
    import torch.nn.functional as F
    from oracle import ohem_ref
    g = torch.Generator(device=device).manual_seed(1)
    low = size // 8
    cells = max(low // 64, 1)
    lab_low = torch.randint(0, NUM_CLASSES, (batch, cells, cells), generator=g, device=device)
    lab_low = lab_low.repeat_interleave(low // cells, 1).repeat_interleave(low // cells, 2)
    target = lab_low.repeat_interleave(8, 1).repeat_interleave(8, 2)
    redraw = torch.rand(batch, low, low, generator=g, device=device) < 0.01
    lab2 = torch.where(redraw, torch.randint(0, NUM_CLASSES, (batch, low, low), generator=g, device=device), lab_low)
    z_tr = (8.0 * F.one_hot(lab2, NUM_CLASSES).permute(0, 3, 1, 2).float()
            + torch.randn(batch, NUM_CLASSES, low, low, generator=g, device=device)).to(torch.bfloat16)
    target[:, :8] = 255
    with torch.no_grad():
        logits = F.interpolate(z_tr.float().cpu(), size=(size, size), mode="bilinear", align_corners=True)
        loss, info = ohem_ref.ohem_cross_entropy(logits, target.cpu(), 255, 0.7, batch * size * size // 16, None,
                                                 return_info=True)
    return {"threshold": round(float(info["threshold"]), 6), "kept": int(info["n_kept"]), "valid": int(info["num_valid"]),
            "kth_branch_taken": bool(info["branch"] == 1), "loss": round(float(loss), 5)}


def psa_probe(device, reps=10, with_oracle=True):
    """SURVEY 8 row a9 in front of the driver: one collect (or distribute) attention of PSANet-R101 at BASELINE configs[4]'s
    per-rank size — out = X @ softmax(A, dim=1), X [2, 512, 3600], A [2, 3600, 3600], bf16 (psanet network.py:119-137) —
    forward and backward (dX, dA) through the C-ABI, HIP-event timed on the launch stream; flops 2 B Cx K N forward, twice
    that backward; `frac` against the 2.5 PF dense bf16 MFMA peak.  With the oracle importable the forward is also checked
    against oracle.psa_ref on the host (the same bf16-rounded operands, fp32 softmax + bmm)."""
    from torchseg_amd import kernels as K
    kp = K.provider()
    B, Cx, Lk = 2, 512, 3600
    g = torch.Generator(device=device).manual_seed(2)
    X = torch.relu(torch.randn(B, Cx, Lk, device=device, generator=g)).to(torch.bfloat16)
    A = torch.randn(B, Lk, Lk, device=device, generator=g).to(torch.bfloat16)
    dout = torch.randn(B, Cx, Lk, device=device, generator=g).to(torch.bfloat16)
    out, lse = kp.psa_fwd(X, A)

    def timed(fn):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e3

    t_f = timed(lambda: kp.psa_fwd(X, A))
    t_b = timed(lambda: kp.psa_bwd(X, A, out, dout, lse))
    flop = 2.0 * B * Cx * Lk * Lk
    rec = {"shape": f"X [{B}, {Cx}, {Lk}] x softmax(A [{B}, {Lk}, {Lk}], dim=1), bf16", "fwd_us": round(t_f, 1),
           "bwd_us": round(t_b, 1), "fwd_TFLOPs": round(flop / t_f / 1e6, 1), "bwd_TFLOPs": round(2 * flop / t_b / 1e6, 1),
           "peak_TFLOPs": 2500.0, "fwd_frac": round(flop / t_f / 1e6 / 2500.0, 4),
           "bwd_frac": round(2 * flop / t_b / 1e6 / 2500.0, 4)}
    if with_oracle:
        try:
            from oracle import psa_ref
            want = psa_ref.psa_attention(X[:1].float().cpu(), A[:1].float().cpu())
            got = out[:1].float().cpu()
            rec["fwd_max_err_vs_oracle"] = float(((got - want).abs().max() / want.abs().max()).item())
        except ImportError:
            pass
    return rec


def _cpu_reference_leg(size, batch, budget_s, max_steps, cores):
    """tools/cpu_reference.py in a process of its own (the reference's module names are the ones our furnace/ answers to
    here): the reference's OWN network.py / seg_oprs.py / resnet.py / loss_opr.py on this host's cores.  None when neither
    the checkout nor the staged archive is there, or when the run fails."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "tools", "cpu_reference.py"), "--size", str(size), "--batch", str(batch),
           "--budget", str(budget_s), "--max-steps", str(max_steps), "--threads", str(cores)]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")      # a CPU process: it must not hold the GPU
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            return {"failed": (r.stderr or r.stdout)[-300:]}
        return json.loads(line[-1])
    except (OSError, subprocess.TimeoutExpired, ValueError) as e:
        return {"failed": "%s: %s" % (type(e).__name__, e)}


def cpu_baseline(headline=True):
    """SURVEY 8(d) "CPU reference timing" on this host's cores, bounded to ~30-40 s of CPU work.

    kind "reference" (round 6): the reference code ITSELF — the unchanged model/bisenet/cityscapes.bisenet.R18/network.py on
    the reference's own seg_oprs.py / resnet.py / init_func.py / loss_opr.py (the `~valid_mask` edit in memory) with
    nn.BatchNorm2d and torch.optim.SGD — run by tools/cpu_reference.py from the files tools/stage_reference.py staged
    (oracle/_ref, git-ignored; /root/reference does not exist on the GPU box).  `value` is the headline 1024 x 1024 crop with
    the batch reduced to 2 (batch 1 is not a legal training batch: the global-context BN sees [B,128,1,1]); BASELINE
    configs[0]'s own 2 x 512 x 512 shape is reported next to it, and the oracle PORT (the same architecture re-typed on our
    furnace surface + the loss_opr.py restatement: what rounds 1-5 reported) beside both.  Where the staged files are missing
    the port is the value and `kind` says so."""
    from torchseg_amd.workloads import ensure_furnace_on_path
    ensure_furnace_on_path()
    ncpu = os.cpu_count() or 1
    cores = min(ncpu, 64)                      # torch's CPU conv stops scaling (and oversubscribes) beyond this
    torch.set_num_threads(cores)
    what = "BiSeNet-R18 (nn.BatchNorm2d + ProbOhemCrossEntropy2d + torch.optim.SGD), fp32, torch CPU"
    ref512 = _cpu_reference_leg(512, 2, 5.0, 5, cores)
    ref1024 = _cpu_reference_leg(1024, 2, 12.0, 3, cores) if headline else None
    v512, n512 = _cpu_leg(512, 2, cores, 4.0, 4)
    port = {"value": v512, "unit": "img/s", "kind": "port",
            "sample": f"{n512} steps (after 1 warm-up) of batch 2 at 512x512, oracle port of {what}"}
    if headline:
        v1024, n1024 = _cpu_leg(1024, 2, cores, 8.0, 2)
        port = {"value": v1024, "unit": "img/s", "kind": "port",
                "sample": f"{n1024} steps (after 1 warm-up) of batch 2 at 1024x1024, oracle port of {what}",
                "configs0": {"value": v512, "unit": "img/s", "sample": port["sample"]}}
    main = ref1024 if headline else ref512
    if not main or "value" not in main:
        out = dict(port)
        out.update({"cores": cores, "host_cpus": ncpu,
                    "reference_leg": main or {"failed": "not run"}})
        return out
    out = {"value": main["value"], "unit": "img/s", "cores": cores, "host_cpus": ncpu, "kind": "reference",
           "sample": f"{main['steps']} steps (after 1 warm-up) of batch 2 at {main['size']}x{main['size']}"
                     + (" (the headline crop of BASELINE configs[1]; batch reduced from 16 to bound the sample)" if headline else
                        " (BASELINE configs[0] shape)")
                     + f": the reference's UNCHANGED network.py + furnace/seg_opr/seg_oprs.py + base_model/resnet.py + "
                       f"seg_opr/loss_opr.py (~valid_mask edit in memory), {what}; files from {main['source']}",
           "port": port}
    if headline and ref512 and "value" in ref512:
        out["configs0"] = {"value": ref512["value"], "unit": "img/s", "kind": "reference",
                           "sample": f"{ref512['steps']} steps (after 1 warm-up) of batch 2 at 512x512 (BASELINE configs[0] shape)"}
    return out


def forced_collectives_run(args):
    """config.forced_collectives: the N > 1 CODE PATH on one rank — TSG_FORCE_COLLECTIVES=1 makes a
    1-rank RCCL group take every exchange step of SURVEY 8(e): SyncBN's collapse -> all-reduce -> finalize per layer and
    direction (70 exchanges per step), the gradient buckets gathered and all-reduced through tsg_comm, the loss all-reduce —
    in a process of its own (a process group cannot be added to a process that has already trained without one), eager
    (the reducer's hooks launch the buckets from inside the backward pass), same shape, same seed.  Zero wire time: what it
    prices is the host / launch side of the path every rank of a 2/4/8-GPU run starts from."""
    import subprocess
    # 12 warm-up steps: an eager step is ~400 launches from Python and the host needs some steps to reach its pace (with 6,
    # one evidence run timed 16.2 ms of host time per step where every other run had 10.3)
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(args.forced_steps), "--warmup", "12",
           "--config", args.config, "--batch", str(args.batch), "--size", str(args.size), "--labels", args.labels,
           "--dtype", args.dtype, "--network", args.network, "--optimizer", args.optimizer, "--graph", "0",
           "--no-cpu-baseline", "--no-ohem-probe", "--no-psa-probe", "--no-kernel-timing", "--i64-steps", "0",
           "--ref-steps", "0", "--fp32-steps", "0", "--forced-steps", "0"]
    env = dict(os.environ, TSG_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0",
               WORLD_SIZE="1", LOCAL_RANK="0")
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            return {"failed": (r.stderr or r.stdout)[-300:]}
        d = json.loads(line[-1])
        return {"value": d["value"], "unit": "img/s", "steps": d["steps"], "ms_per_step": d["ms_per_step"],
                "host_enqueue_ms_per_step": d["config"]["host_enqueue_ms_per_step"], "hip_graph": d["config"]["hip_graph"],
                "note": "TSG_FORCE_COLLECTIVES=1 on a 1-rank RCCL group, eager: SyncBN exchanges + gradient buckets + loss "
                        "all-reduce through tsg_comm; no wire time in it"}
    except (OSError, subprocess.TimeoutExpired, ValueError, KeyError) as e:
        return {"failed": "%s: %s" % (type(e).__name__, e)}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher: become the launcher.  Re-executes this file under
    torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1, a free port) exactly as the driver's
    multi-GPU command line does, and passes rank 0's JSON line through as the last line of stdout."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    sys.stdout.flush()
    return subprocess.call(cmd, env=env)


def launch_check(world, rank, args=None):
    """--launch-check: rendezvous only (gloo when there is no GPU), so that the launcher logic is testable on a CPU
    box: every rank joins, the world size is all-reduced, rank 0 prints it (and the batch split --scaling implies)."""
    backend = "nccl" if torch.cuda.is_available() and torch.cuda.device_count() >= world else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend=backend, init_method="env://")
    one = torch.ones(1, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(one)
    n = dist.get_world_size()
    ok = int(one.item()) == n
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        rec = {"launch_check": ok, "n_gpus": n, "backend": backend}
        if args is not None:
            rec.update({"scaling": args.scaling, "per_rank_batch": args.batch, "global_batch": args.batch * n,
                        "min_kept": int(args.batch * args.size * args.size // 16)})
        print(json.dumps(rec), flush=True)
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)       # >= 20 warm-up + >= 50 timed steps
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="bisenet", choices=sorted(CONFIGS),
                    help="BASELINE.json configuration at its per-rank shape (default: configs[1], the headline)")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the config's)")
    ap.add_argument("--size", type=int, default=None, help="crop (default: the config's)")
    ap.add_argument("--labels", default=None, choices=["u8", "i64"],
                    help="label dtype on the device: u8 (what the GPU loader emits; default for the ignore-255 families) "
                         "or i64 (what the reference's DataLoader hands over)")
    ap.add_argument("--no-ohem-probe", action="store_true", help="skip the k-th-branch head record (bisenet only)")
    ap.add_argument("--no-psa-probe", action="store_true", help="skip the PSA attention record (SURVEY 8 row a9)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: the config's batch on EVERY rank (default); strong: the config's batch is the GLOBAL batch, "
                         "split over the ranks as the reference does (dataloader.py:51-53; min_kept follows, train.py:48-49)")
    ap.add_argument("--i64-steps", type=int, default=10,
                    help="after the timed region, time this many extra steps with int64 labels (what the reference's "
                         "DataLoader hands over) and report them beside the uint8 default; 0 = skip")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--network", default="native", choices=["native", "reference"],
                    help="native: the architecture re-typed on the furnace surface (torchseg_amd/workloads, calls the fused "
                         "operators directly); reference: the reference's UNCHANGED model/<family>/<experiment>/network.py "
                         "(staged by tools/stage_reference.py), fused through fusion.FuseMode behind our DDP wrapper")
    ap.add_argument("--ref-steps", type=int, default=20,
                    help="after the timed region of the native network, time this many steps of the reference's unchanged "
                         "network.py (same shape, same seed) and report them as config.reference_network; 0 = skip")
    ap.add_argument("--fp32-steps", type=int, default=3,
                    help="after the timed region, time this many steps in the fp32 parity mode (exact convolutions, fp64 "
                         "BatchNorm statistics: the kernels the 1e-4 parity claim is made with) -> fp32_mode; 0 = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-headline", type=int, default=1,
                    help="also time ONE CPU step at the headline 1024x1024 shape (batch 2; about a minute of host time)")
    ap.add_argument("--launch-check", action="store_true", help="rendezvous of --gpus N ranks only, no GPU work (CPU-testable)")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("TSG_GRAPH", "-1")),
                    help="0: eager step; 1: replay zero_grad+forward+backward from a hipGraph, optimizer eager after each "
                         "replay; 2: the FusedSGD step is captured too.  Default (-1): 2 on one GPU without a process group "
                         "(the step is one linear graph: host cost 0.2 ms instead of 11-13 ms per step), 0 whenever "
                         "gradients are reduced (N > 1, TSG_FORCE_COLLECTIVES) or the optimizer is torch's; falls back to "
                         "eager, and says so in config.hip_graph_fallback, if the capture fails")
    ap.add_argument("--mode-probe", type=int, default=6,
                    help="with the default --graph (-1) on one GPU: time this many replayed and this many eagerly launched steps "
                         "after the capture and run the timed region in the faster mode (config.mode_probe); 0 = always replay")
    ap.add_argument("--forced-steps", type=int, default=20,
                    help="after everything else (rank 0, N = 1): time this many steps of the N > 1 CODE PATH on one rank "
                         "(TSG_FORCE_COLLECTIVES=1 in a process of its own: SyncBN exchanges + gradient buckets through "
                         "tsg_comm on a 1-rank RCCL group) -> config.forced_collectives; 0 = skip")
    ap.add_argument("--optimizer", default="fused", choices=["fused", "torch"])
    ap.add_argument("--trace-loss", action="store_true", help="debug: print the loss of every timed step (syncs)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--miopen-find", type=int, default=int(os.environ.get("TSG_MIOPEN_FIND", "-1")),
                    help="torch.backends.cudnn.benchmark (train.py:35); 0 = immediate mode on the shipped MIOpen find-db (same "
                         "speed, 100 s faster start: the default)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    if args.batch is None:
        args.batch = cfg["batch"]
    if args.size is None:
        args.size = cfg["size"]
    if args.labels is None:
        args.labels = "u8" if args.config in ("bisenet", "dfn") else "i64"

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))             # plain `python bench.py --gpus N`
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.scaling == "strong":
        if args.batch % world or args.batch // world < 2:
            sys.exit(f"bench.py: --scaling strong needs the global batch {args.batch} to split into >= 2 images on each of "
                     f"{world} ranks (the global-context BatchNorm sees [B, 128, 1, 1])")
        args.batch //= world                                           # dataloader.py:51-53
    if args.launch_check:
        sys.exit(launch_check(world, rank, args))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
        os.environ["NCCL_DEBUG"] = "WARN"          # keep RCCL's banner off stdout (one JSON line contract)
    os.environ["TSG_DTYPE"] = args.dtype
    from torchseg_amd.tuning import use_shipped_miopen_db
    use_shipped_miopen_db(rank=rank)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an AMD GPU (the HIP path has no CPU fallback)")
    if torch.cuda.device_count() < world:
        sys.exit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) visible (one rank per GPU)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    force_coll = os.environ.get("TSG_FORCE_COLLECTIVES", "0") == "1"     # 1-rank run of the N>1 code path
    if world > 1 or force_coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", init_method="env://")
    if args.miopen_find < 0:
        # immediate mode on the shipped find-db (torchseg_amd/miopen_db: the entries MIOpen's find mode wrote for the
        # four configs at their bench shapes, tools/gpu_r3_g.sh; same speed as cudnn.benchmark = True — train.py:35 —
        # and a 100 s faster start).  Without entries immediate mode picks kernels 5-10x off (1 ms igemm forwards for
        # PSPNet's dilated 3x3 layers, profiles/r03_kernel_stats_pspnet.csv, before the db had them)
        args.miopen_find = 0
    torch.backends.cudnn.benchmark = bool(args.miopen_find)            # train.py:35

    from torchseg_amd import kernels as K
    from torchseg_amd.ddp import DistributedDataParallel
    from torchseg_amd.losses import ProbOhemCrossEntropy2d, SigmoidFocalLoss
    from torchseg_amd.syncbn import SyncBatchNorm
    from torchseg_amd.workloads import ensure_furnace_on_path
    ensure_furnace_on_path()
    from engine.lr_policy import PolyLR

    auto_mode = args.graph < 0
    if args.graph < 0:
        args.graph = 2 if (world == 1 and not force_coll and args.optimizer == "fused") else 0
    use_graph = bool(args.graph)
    mode_probe = None
    graph_fallback = None
    model, opt, base_lr = build_model(device, args.batch, args.size, ProbOhemCrossEntropy2d, SyncBatchNorm,
                                      seed=12345 if world == 1 else local_rank,       # train.py:37-40
                                      fused_sgd=args.optimizer == "fused", config=args.config,
                                      focal_cls=SigmoidFocalLoss, network=args.network)
    model = DistributedDataParallel(model)                             # train.py:98-99
    model.train()
    batch = synthetic_batch(device, args.batch, args.size, seed=rank, config=args.config,
                            label_dtype=torch.uint8 if args.labels == "u8" else torch.int64)
    pol = PolyLR(base_lr, 0.9, 80 * 1000)

    def sync():
        if world > 1 or force_coll:
            dist.barrier()
        torch.cuda.synchronize()

    # Per-kernel timing brackets every launch with HIP events, which costs host time
    # (~600 event records per step); it must not distort `value`.  So: the last eager
    # warm-up step runs fully instrumented to learn which of our kernels dominates, and
    # (eager mode only) the timed region instruments ONLY that kernel.  Under graph
    # replay there are no host-side launches to bracket: the roofline then comes from
    # the instrumented warm-up step of the same process.
    timer = None
    dominant = None
    all_kernels = None
    n_eager = max(args.warmup, 3) if use_graph else args.warmup        # capture needs warmed-up libraries
    # Graph mode: the eager warm-up, the capture, every replay and the optimizer run on ONE stream (GraphedStep.stream).
    run_stream = GraphedStep.capture_stream() if use_graph else None
    if run_stream is not None:
        run_stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(run_stream) if run_stream is not None else contextlib.nullcontext():
        probe = None
        n_probe = min(3, n_eager)                        # instrumented warm-up steps: every launch = the minimum of its brackets
        for it in range(n_eager):
            if not args.no_kernel_timing and it == n_eager - n_probe:
                probe = K.KernelTimer(K.provider())
            with serial_kernels(probe is not None):
                loss = train_step(model, opt, batch, pol, it, world)
            if probe is not None:
                torch.cuda.synchronize()
                probe.mark_step()
            if probe is not None and it == n_eager - 1:
                probe.stop()
                dom_family = probe.dominant_family()
                fam = probe.family_stats().get(dom_family, {}).get("members") or []
                # the timed region brackets only the LARGEST LABEL of the dominant family (two event records per launch
                # cost GPU time: ~120 SyncBN launches per step would show in `value`); the family figures come from this
                # fully instrumented warm-up step
                dominant = max(fam, key=lambda n: probe.stats[n]["total_ms"]) if fam else probe.dominant()
                all_kernels = probe.summary()
                roof_probe = probe
        sync()
        graphed = None
        if use_graph:
            try:
                graphed = GraphedStep(model, opt, batch, world, opt_inside=args.graph == 2)
                for it in range(2):                                        # untimed replays
                    set_lr(opt, pol, n_eager + it)
                    loss = graphed()
                sync()
                if not bool(torch.isfinite(loss.detach()).all()):
                    raise RuntimeError("non-finite loss after two replays")
            except Exception as e:                                         # noqa: BLE001 - the eager step is always there
                graph_fallback = "%s: %s" % (type(e).__name__, str(e)[:200])
                graphed, use_graph = None, False
                torch.cuda.synchronize()
                for it in range(2):
                    loss = train_step(model, opt, batch, pol, n_eager + it, world)
                sync()
        replay = graphed is not None
        segmented = None
        if graphed is not None and auto_mode and args.mode_probe > 0 and os.environ.get("TSG_SEGMENTED_GRAPH", "1") != "0" \
                and SegmentedStep.applies(model, world):
            try:
                segmented = SegmentedStep(model, opt, batch)
                for it in range(2):
                    set_lr(opt, pol, n_eager + it)
                    lseg = segmented()
                sync()
                if not bool(torch.isfinite(lseg.detach()).all()):
                    raise RuntimeError("non-finite loss after two segmented replays")
            except Exception as e:                                         # noqa: BLE001 - the other two modes are always there
                graph_fallback = "segmented: %s: %s" % (type(e).__name__, str(e)[:160])
                segmented = None
                torch.cuda.synchronize()
        if graphed is not None and auto_mode and args.mode_probe > 0:
            # The replayed graph is ONE stream: host cost 0.2 ms, no overlap.  The eager step forks the weight gradients and the
            # two auxiliary heads onto side streams (VALU-bound criteria beside matrix-core-bound convolutions: +2-4 %) but needs
            # the host to enqueue ~400 launches in less than the GPU takes for them — which holds on some boxes and not on
            # others (host 9-13 ms against a 12.2-13 ms step).  Same kernels, same arithmetic either way: time both, keep the faster.
            def _time(fn, n):
                sync()
                t = time.perf_counter()
                for i in range(n):
                    fn(i)
                sync()
                return (time.perf_counter() - t) / n * 1e3

            def _replayed(i):
                set_lr(opt, pol, n_eager + 2 + i)
                graphed()

            for i in range(2):
                train_step(model, opt, batch, pol, n_eager + 2 + i, world)
            ms_e = _time(lambda i: train_step(model, opt, batch, pol, n_eager + 4 + i, world), args.mode_probe)
            ms_g = _time(_replayed, args.mode_probe)
            ms_s, ms_whole = None, ms_g
            if segmented is not None:
                def _seg(i):
                    set_lr(opt, pol, n_eager + 2 + i)
                    segmented()
                ms_s = _time(_seg, args.mode_probe)
                if ms_s < ms_g:                            # the five-graph form of the replayed step takes its place
                    graphed, ms_g = segmented, ms_s
            replay = ms_g <= ms_e * 1.005
            mode_probe = {"replayed_ms_per_step": round(ms_g, 3), "eager_ms_per_step": round(ms_e, 3), "steps": args.mode_probe,
                          "whole_graph_ms_per_step": round(ms_whole, 3),
                          "segmented_ms_per_step": None if ms_s is None else round(ms_s, 3),
                          "chosen": ("replay" if graphed is not segmented else "segmented replay") if replay else "eager",
                          "note": "timed region = the fastest of hipGraph replay (one stream), segmented replay (fifteen linear graphs "
                                  "on three streams: detail branch, heads and deferred weight gradients beside the context path) and "
                                  "eager launches (weight gradients, auxiliary heads and detail branch on side streams); same "
                                  "kernels and results"}
        if graphed is None and dominant is not None:
            timer = K.KernelTimer(K.provider(), names=[dominant])
        t0 = time.perf_counter()
        for it in range(args.steps):
            if replay:
                set_lr(opt, pol, args.warmup + it)
                loss = graphed()
            else:
                if timer is not None:
                    # the dominant label is bracketed with HIP events on every 5th timed step only: 34 bracketed SyncBN
                    # launches per step cost 0.16 ms of the step (1177 against 1191 img/s in one process, round 5) —
                    # sampled, the roofline figure is still measured live inside the timed region (10 of 50 default steps)
                    timer.enabled = (it % 5 == 0)
                with serial_kernels(timer is not None and timer.enabled):
                    loss = train_step(model, opt, batch, pol, args.warmup + it, world)
            if args.trace_loss and rank == 0:
                print("step", it, "loss", float(loss.item()), "lr", opt.param_groups[0]["lr"], file=sys.stderr, flush=True)
        t_host = time.perf_counter() - t0                              # all K steps enqueued (the GPU may still be running)
        sync()
        dt = time.perf_counter() - t0
        final_loss = float(loss.item())
        eager_rec = None
        if graphed is not None:
            # Replayed launches cannot be bracketed from the host.  The same step, eager, right behind the replayed region on
            # the same stream: (1) `eager_steps` = what the host costs (value / host ms beside the replayed figure), (2) the
            # dominant family's largest label bracketed with HIP events on every step of a second short run = the live
            # `timed_region` roofline sample (the fully instrumented warm-up step stays the family table's source)
            n_e = max(4, min(10, args.steps))
            for it in range(2):
                train_step(model, opt, batch, pol, args.warmup + args.steps + it, world)
            sync()
            t1 = time.perf_counter()
            for it in range(n_e):
                train_step(model, opt, batch, pol, args.warmup + args.steps + 2 + it, world)
            th = time.perf_counter() - t1
            sync()
            de = time.perf_counter() - t1
            eager_rec = {"value": round(args.batch * world * n_e / de, 2), "unit": "img/s", "steps": n_e,
                         "ms_per_step": round(de / n_e * 1e3, 3), "host_enqueue_ms_per_step": round(th / n_e * 1e3, 3),
                         "note": "the same step launched eagerly (weight gradients on their side stream), timed right after "
                                 "the replayed region"}
            if dominant is not None:
                timer = K.KernelTimer(K.provider(), names=[dominant])
                with serial_kernels(True):
                    for it in range(4):
                        train_step(model, opt, batch, pol, args.warmup + args.steps + 2 + n_e + it, world)
                sync()
        # The reference's DataLoader hands the criterion int64 labels; the GPU loader of this package emits uint8 (the default
        # here, disclosed as config.labels).  A few extra steps with int64 labels, outside the timed region, put the other
        # figure beside it (the label width must be visibly neutral).
        i64_rec = None
        if args.i64_steps > 0 and args.labels == "u8":
            b64 = tuple(t.to(torch.int64) if t.dtype == torch.uint8 else t for t in batch)
            for it in range(2):
                train_step(model, opt, b64, pol, args.warmup + args.steps + it, world)
            sync()
            t1 = time.perf_counter()
            for it in range(args.i64_steps):
                train_step(model, opt, b64, pol, args.warmup + args.steps + 2 + it, world)
            sync()
            d64 = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=device)
            if world > 1:
                dist.all_reduce(d64, op=dist.ReduceOp.MAX)
            i64_rec = {"value": round(args.batch * world * args.i64_steps / float(d64.item()), 2), "unit": "img/s",
                       "steps": args.i64_steps, "ms_per_step": round(float(d64.item()) / args.i64_steps * 1e3, 3),
                       "note": "same step with int64 labels (what the reference's DataLoader hands over), launched eagerly, "
                               "timed after the uint8 region"}
    if run_stream is not None:
        torch.cuda.current_stream().wait_stream(run_stream)
    if timer is not None:
        timer.stop()
    elif dominant is not None:
        timer = roof_probe

    def side_run(net, optim, data, steps, warm, first_it):
        """`steps` timed steps of another (model, optimizer) at the same shape, outside the headline's timed region."""
        from torchseg_amd import fusion
        for it in range(warm):
            train_step(net, optim, data, pol, first_it + it, world)
        sync()
        before = dict(fusion.stats)
        cc = K.CallCounter(K.provider())
        train_step(net, optim, data, pol, first_it + warm, world)
        sync()
        calls = cc.stop()
        fused = {k: fusion.stats[k] - before[k] for k in fusion.stats if fusion.stats[k] != before[k]}
        t1 = time.perf_counter()
        for it in range(steps):
            last = train_step(net, optim, data, pol, first_it + warm + 1 + it, world)
        th = time.perf_counter() - t1
        sync()
        d = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(d, op=dist.ReduceOp.MAX)
        d = float(d.item())
        return {"value": round(args.batch * world * steps / d, 2), "unit": "img/s", "steps": steps,
                "ms_per_step": round(d / steps * 1e3, 3), "host_enqueue_ms_per_step": round(th / steps * 1e3, 3),
                "final_loss": round(float(last.item()), 4)}, calls, fused

    # The reference's UNCHANGED network.py: same shape, same seed, same wrapper; its fused operators
    # are reached through fusion.FuseMode instead of being called by the builder.  Timed after the headline region.
    ref_rec = None
    if args.ref_steps > 0 and args.network == "native" and args.dtype == "bf16" and world == 1:   # rank 0 at N = 1 only, like cpu_baseline
        try:
            rmodel, ropt, _ = build_model(device, args.batch, args.size, ProbOhemCrossEntropy2d, SyncBatchNorm,
                                          seed=12345 if world == 1 else local_rank, fused_sgd=args.optimizer == "fused",
                                          config=args.config, focal_cls=SigmoidFocalLoss, network="reference")
        except (FileNotFoundError, ImportError) as e:
            ref_rec = {"skipped": "%s: %s" % (type(e).__name__, e)}
        else:
            rmodel = DistributedDataParallel(rmodel)
            rmodel.train()
            ref_rec, rcalls, rfused = side_run(rmodel, ropt, batch, args.ref_steps, 6, 0)
            take = ("ohem_up_fwd", "ohem_up_bwd", "upsample_presum_fwd", "stem_conv_fwd_stats", "stem_conv_wrw_bn",
                    "conv3x3_c64_fwd", "conv3x3_gen_fwd", "conv3x3_wrw", "cls_head_fwd", "psa_fwd", "psa_bwd", "ohem_fwd",
                    "focal_fwd", "bn_apply_fwd", "bn_bwd_apply")
            ref_rec.update({"file": CONFIGS[args.config]["ref"] + "/network.py (unchanged; tools/stage_reference.py)",
                            "fused_by_FuseMode_per_step": rfused,
                            "kernel_calls_per_step": {k: rcalls[k] for k in take if k in rcalls},
                            "launches_via_c_abi_per_step": int(sum(rcalls.values()))})
            del rmodel, ropt
    # fp32 = the parity mode (the kernels the 1e-4 claim is made with: exact convolutions, fp64 BatchNorm statistics);
    # what it costs, beside the bf16 headline (VERDICT r4 missing #4)
    fp32_rec = None
    if args.fp32_steps > 0 and args.dtype == "bf16" and world == 1:
        fmodel, fopt, _ = build_model(device, args.batch, args.size, ProbOhemCrossEntropy2d, SyncBatchNorm, seed=12345,
                                      fused_sgd=args.optimizer == "fused", config=args.config, focal_cls=SigmoidFocalLoss,
                                      network=args.network)
        fmodel = DistributedDataParallel(fmodel, compute_dtype=torch.float32)
        fmodel.train()
        fp32_rec, _, _ = side_run(fmodel, fopt, batch, args.fp32_steps, 1, 0)
        fp32_rec["note"] = ("TSG_DTYPE=fp32: every convolution on tsg_conv2d_f32_exact_* (fp64 accumulation), fp64 BatchNorm "
                            "statistics; the mode tests/test_headline_gpu.py holds to 1e-4 against the CPU path")
        del fmodel, fopt
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    n_ranks = dist.get_world_size() if dist.is_initialized() else 1
    assert n_ranks == world == args.gpus, (n_ranks, world, args.gpus)

    out = None
    if rank == 0:
        global_batch = args.batch * world
        value = global_batch * args.steps / dt
        out = {
            "metric": "training images/sec (1024x1024) BiSeNet-R18" if args.config == "bisenet"
                      else f"training images/sec ({args.size}x{args.size}) {cfg['model']}",
            "value": round(value, 2), "unit": "img/s", "n_gpus": n_ranks, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{cfg['model']} {args.dtype} batch {args.batch}/GPU {args.size}x{args.size} synthetic crops, "
                                   + {"bisenet": "SyncBN + OHEM", "pspnet": "SyncBN, 150 classes, 2 CE heads",
                                      "dfn": "SyncBN, 4 CE + 4 sigmoid-focal heads",
                                      "psanet": "SyncBN, 150 classes, collect/distribute attention (MFMA)"}[args.config]
                                   + f" (BASELINE configs[{cfg['idx']}], per-rank shape; {cfg['ref']})",
                       "network": ("reference: unchanged %s/network.py behind fusion.FuseMode" % cfg["ref"])
                       if args.network == "reference" else "native: torchseg_amd/workloads builder (same architecture, "
                       "fused operators called directly)",
                       "reference_network": ref_rec, "fp32_mode": fp32_rec,
                       "host_enqueue_ms_per_step": round(t_host / args.steps * 1e3, 3),
                       "labels": args.labels, "labels_i64": i64_rec,
                       "global_batch": global_batch, "per_rank_batch": args.batch, "parallelism": f"dp{world}",
                       "channels_last": model.channels_last, "final_loss": round(final_loss, 4),
                       "hip_graph": bool(use_graph and replay), "hip_graph_mode": int(args.graph) if (use_graph and replay) else 0,
                       "mode_probe": mode_probe,
                       "hip_graph_fallback": graph_fallback, "eager_steps": eager_rec,
                       "forced_collectives": None, "optimizer": args.optimizer},
        }
        if timer is not None:
            # the dominant kernel FAMILY (all bn_* passes are one family: SyncBN) from the fully instrumented last warm-up
            # step, the three largest families beside it; `timed_region` = the family's largest label bracketed with HIP
            # events over the K timed steps (the same kernels, measured live where `value` is measured)
            out["roofline"] = roof_probe.roofline(HBM_PEAK_GBS, os.path.join(ROOT, "profiles"), by_family=True)
            out["roofline"]["measured_over"] = ("the last three warm-up steps, fully instrumented (every launch of every family bracketed; "
                                               "a launch = the minimum of its three brackets: a bracket holds the host-side call too)")
            if timer is not roof_probe:
                out["roofline"]["timed_region"] = timer.roofline(HBM_PEAK_GBS, os.path.join(ROOT, "profiles"))
                out["roofline"]["timed_region"]["sampled"] = (
                    "4 eagerly launched steps right behind the timed region (the launches of a replayed hipGraph cannot be "
                    "bracketed from the host, and an eager timed region is left unbracketed: two event records per launch cost GPU "
                    "time; same kernels, same process)" if use_graph else
                    "every 5th timed step; bracketed steps (and the instrumented warm-up step) keep the weight gradients on "
                    "the compute stream")
            out["kernels_last_warmup_step"] = all_kernels
        if args.config == "bisenet" and args.dtype == "bf16" and args.size == 1024:
            # whole-step HBM roofline of SURVEY.md 8(d): ~3.4 GB of algorithmic traffic per image in bf16 (our
            # kernels 2.16 GB + the convolutions' operands once each + pools) => 2350 img/s per GPU at 8 TB/s
            gb_per_img = 3.4
            per_gpu = value / world
            out["step_roofline"] = {"bound": "hbm", "algo_GB_per_img": gb_per_img,
                                    "achieved": round(per_gpu * gb_per_img, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(per_gpu * gb_per_img / HBM_PEAK_GBS, 4), "target_frac": 0.70}
        if args.config == "bisenet" and world == 1 and not args.no_ohem_probe and args.dtype == "bf16":
            out["ohem_kth_branch"] = ohem_kth_branch_probe(device, args.batch, args.size)
            if not args.no_cpu_baseline:                   # the checker beside the device's selection (host time ~15 s)
                try:
                    out["ohem_kth_branch"]["trained_like"]["oracle"] = ohem_kth_branch_oracle(device, args.batch, args.size)
                except ImportError:
                    pass
        if world == 1 and not args.no_psa_probe and args.dtype == "bf16":
            out["psa_probe"] = psa_probe(device, with_oracle=not args.no_cpu_baseline)
        if world == 1 and not force_coll and args.forced_steps > 0:
            torch.cuda.synchronize()
            out["config"]["forced_collectives"] = forced_collectives_run(args)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(headline=bool(args.cpu_headline)) if args.config == "bisenet" \
                else cpu_baseline_family(args.config)
    if world > 1 or force_coll:
        from torchseg_amd import comm as tsg_comm
        tsg_comm.shutdown()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner through C stdio, which is block-buffered when stdout is a
        # pipe/file: flush it first so that the JSON line is the LAST thing on stdout.
        import ctypes
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
