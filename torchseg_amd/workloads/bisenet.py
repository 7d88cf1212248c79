"""BiSeNet (ResNet-18 context path) — the BASELINE.json metric model.

Architecture of model/bisenet/cityscapes.bisenet.R18/network.py:18-168
(BiSeNet :18-111, SpatialPath :114-137, BiSeNetHead :140-168), reproduced on
top of the furnace surface so that bench.py / smoke() can run where the
reference tree is absent.  Module attribute names and construction order match
the reference file, hence state dicts are interchangeable and a fixed seed
initialises both identically (tests/test_dropin_cpu.py checks both).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import add_then_upsample, ensure_furnace_on_path, upsample_logits

ensure_furnace_on_path()
from base_model import resnet18  # noqa: E402
from seg_opr.seg_oprs import AttentionRefinement, ConvBnRelu, FeatureFusion, cbr_chain  # noqa: E402


import os as _os
# TSG_FORK_SPATIAL=2|3|1|0 (default 2, round 6): SpatialPath on a side HIP stream beside the context path (forward and, through
# autograd, backward) — 1: from the start (round 5: +1.4 % / -0.2 %, the context path's stem and layer1 are as HBM-bound as the
# detail branch); 2 / 3: started behind the context path's layer1 / layer2, beside the matrix-core-bound deep layers: eager
# 12.43-12.62 -> 12.20-12.29 ms (gpurun_out/r6b_call14.txt).  The stream is auxiliary head 0's: HIP maps a process's streams onto
# four hardware queues (compute, weight gradients, two heads) and a fifth stream shares one of them — which is why the fork
# measured nothing while it had a stream of its own.  Eager launches only (see TSG_FORK_HEADS); the unchanged network.py gets the
# early fork through TSG_FORK_MODULES=spatial_path (ddp.py).
_FORK_SPATIAL_MODE = int(_os.environ.get("TSG_FORK_SPATIAL", "2"))     # 1: from the start; 2: behind the context path's layer1
_FORK_SPATIAL = _FORK_SPATIAL_MODE > 0
# TSG_FORK_IN_GRAPH=1: keep the fork while the step is being captured into a hipGraph (ONE fork / join per direction)
_FORK_IN_GRAPH = _os.environ.get("TSG_FORK_IN_GRAPH", "0") == "1"
# TSG_FORK_HEADS=1|0 (default 1, round 6): the two auxiliary heads + their criteria on side streams of their own beside the main
# head (they share nothing: network.py:103-108).  A fused head is VALU-bound (tsg_ohem_up_fwd / _bwd: 100-160 us with HBM and the
# matrix cores idle), the 3x3 convolution in front of it is matrix-core bound: run side by side they fill each other's gaps —
# eager step 1 247-1 258 -> 1 280-1 282 img/s on one box, 1 285 -> 1 315 on another (gpurun_out/r6b_call9/10.txt).  Eager launches
# in a single process only: inside a captured hipGraph the same fork is SLOWER (1 268 -> 1 230: the runtime replays a graph with
# parallel branches node by node), and with a process group the exchanges of three streams would interleave.
_FORK_HEADS = _os.environ.get("TSG_FORK_HEADS", "1") == "1"
_SIDE = {}


def _side_stream(device, which=0):
    import torch
    s = _SIDE.get((device, which))
    if s is None:
        s = _SIDE[(device, which)] = torch.cuda.Stream(device=device)
    return s


def _single_process():
    """no process group at all: with one (even of a single rank: TSG_FORCE_COLLECTIVES) the SyncBN exchanges of three streams
    would be issued on one communicator concurrently"""
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized())


def _cbr(cin, cout, k, s, p, norm_layer, relu=True):
    return ConvBnRelu(cin, cout, k, s, p, has_bn=True, norm_layer=norm_layer, has_relu=relu, has_bias=False)


def _up(x, size=None, scale=None):
    return F.interpolate(x, size=size, scale_factor=scale, mode='bilinear', align_corners=True)


class SpatialPath(nn.Module):
    """7x7/2 -> 3x3/2 -> 3x3/2 -> 1x1: 1/8-resolution detail branch (network.py:114-137)."""

    def __init__(self, in_planes, out_planes, norm_layer=nn.BatchNorm2d):
        super(SpatialPath, self).__init__()
        mid = 64
        self.conv_7x7 = _cbr(in_planes, mid, 7, 2, 3, norm_layer)
        self.conv_3x3_1 = _cbr(mid, mid, 3, 2, 1, norm_layer)
        self.conv_3x3_2 = _cbr(mid, mid, 3, 2, 1, norm_layer)
        self.conv_1x1 = _cbr(mid, out_planes, 1, 1, 0, norm_layer)

    def forward(self, x):
        return cbr_chain([self.conv_7x7, self.conv_3x3_1, self.conv_3x3_2, self.conv_1x1], x)


class BiSeNetHead(nn.Module):
    """3x3 CBR -> 1x1 classifier -> bilinear x`scale` (network.py:140-168)."""

    def __init__(self, in_planes, out_planes, scale, is_aux=False, norm_layer=nn.BatchNorm2d):
        super(BiSeNetHead, self).__init__()
        mid = 256 if is_aux else 64
        self.conv_3x3 = _cbr(in_planes, mid, 3, 1, 1, norm_layer)
        self.conv_1x1 = nn.Conv2d(mid, out_planes, kernel_size=1, stride=1, padding=0)
        self.scale = scale

    def forward(self, x):
        out = self.conv_1x1(self.conv_3x3(x))
        return upsample_logits(out, scale=self.scale) if self.scale > 1 else out


class BiSeNet(nn.Module):
    tsg_native_fusions = True      # calls the fused operators itself (workloads/__init__.py)

    def __init__(self, out_planes, is_training, criterion, pretrained_model=None,
                 norm_layer=nn.BatchNorm2d, bn_eps=1e-5, bn_momentum=0.1):
        super(BiSeNet, self).__init__()
        self.context_path = resnet18(pretrained_model, norm_layer=norm_layer, bn_eps=bn_eps,
                                     bn_momentum=bn_momentum, deep_stem=False, stem_width=64)
        self.business_layer = []
        self.is_training = is_training
        self.spatial_path = SpatialPath(3, 128, norm_layer)
        ch = 128
        self.global_context = nn.Sequential(nn.AdaptiveAvgPool2d(1), _cbr(512, ch, 1, 1, 0, norm_layer))
        arms = [AttentionRefinement(512, ch, norm_layer), AttentionRefinement(256, ch, norm_layer)]
        refines = [_cbr(ch, ch, 3, 1, 1, norm_layer), _cbr(ch, ch, 3, 1, 1, norm_layer)]
        heads = [BiSeNetHead(ch, out_planes, 16, True, norm_layer),
                 BiSeNetHead(ch, out_planes, 8, True, norm_layer),
                 BiSeNetHead(ch * 2, out_planes, 8, False, norm_layer)]
        self.ffm = FeatureFusion(ch * 2, ch * 2, 1, norm_layer)
        self.arms = nn.ModuleList(arms)
        self.refines = nn.ModuleList(refines)
        self.heads = nn.ModuleList(heads)
        self.business_layer += [self.spatial_path, self.global_context, self.arms, self.refines,
                                self.heads, self.ffm]
        if is_training:
            self.criterion = criterion

    def features(self, data):
        """-> [1/16 aux fm, 1/8 aux fm, fused 1/8 fm] (network.py:75-101)."""
        fork = None
        late = _FORK_SPATIAL_MODE >= 2
        can_fork = (data.is_cuda and _FORK_SPATIAL and (_FORK_IN_GRAPH or not torch.cuda.is_current_stream_capturing())
                    and _single_process())

        def run_spatial():
            # the two paths share nothing until the fusion module: the detail branch (large maps: HBM-bound BatchNorm passes
            # and stems) on a side stream beside the context path's deep layers (small maps: matrix-core bound, too few tiles
            # to fill the chip on their own); autograd replays each node on its forward stream, so the backward overlaps the
            # same way.  The stream is auxiliary head 0's (idle until the heads): a FIFTH stream would share a hardware queue.
            cur = torch.cuda.current_stream(data.device)
            side = _side_stream(data.device, 1)
            side.wait_stream(cur)
            data.record_stream(side)
            with torch.cuda.stream(side):
                out = self.spatial_path(data)
            return out, cur, side

        if can_fork and not late:
            spatial_out, cur, fork = run_spatial()
        if can_fork and late:
            # the context path's stem and layer1 are HBM-bound like the detail branch: start the branch behind them
            cp = self.context_path
            c2 = self.context_head(data)
            if _FORK_SPATIAL_MODE == 2:
                spatial_out, cur, fork = run_spatial()
                outs = self.context_tail(c2)
            else:
                c3 = cp.layer2(c2)
                spatial_out, cur, fork = run_spatial()
                outs = self.context_tail(c2, c3)
        else:
            if fork is None:
                spatial_out = self.spatial_path(data)
            outs = self.context_tail(None, None, self.context_path(data))
        if fork is not None:
            cur.wait_stream(fork)
            spatial_out.record_stream(cur)
        outs.append(self.ffm(spatial_out, outs[-1]))
        return outs

    def context_head(self, data):
        """stem + layer1 of the context path (what bench.SegmentedStep and the late fork above run in front of the detail branch)"""
        cp = self.context_path
        return cp.layer1(cp._stem(data))

    def context_tail(self, c2, c3=None, blocks=None):
        """layer2 .. layer4, global context, the two attention-refinement stages -> [1/16 fm, 1/8 fm] (network.py:76-99)"""
        if blocks is None:
            cp = self.context_path
            if c3 is None:
                c3 = cp.layer2(c2)
            c4 = cp.layer3(c3)
            c5 = cp.layer4(c4)
        else:
            c2, c3, c4, c5 = blocks
        f16 = self._refine_stage(0, c5, _up(self.global_context(c5), size=c5.shape[2:]), c4)
        return [f16, self._refine_stage(1, c4, f16, c3)]

    def _refine_stage(self, k, fm, last_fm, nxt):
        """one pass of network.py:91-95: refine(upsample(arm(fm) + last_fm)) at the resolution of `nxt`"""
        return self.refines[k](add_then_upsample(self.arms[k](fm), last_fm, nxt.shape[2:]))

    def context_tail_first(self, c2):
        """context_tail up to the 1/16 feature map, which auxiliary head 0 needs and nothing else of the forward waits for
        (bench.SegmentedStep starts that head beside the second stage) -> (1/16 fm, c3, c4)"""
        cp = self.context_path
        c3 = cp.layer2(c2)
        c4 = cp.layer3(c3)
        c5 = cp.layer4(c4)
        return self._refine_stage(0, c5, _up(self.global_context(c5), size=c5.shape[2:]), c4), c3, c4

    def context_tail_second(self, f16, c3, c4):
        """the second attention-refinement stage -> 1/8 fm"""
        return self._refine_stage(1, c4, f16, c3)

    def forward(self, data, label=None):
        f16, f8, fused = self.features(data)
        if self.is_training:
            if (data.is_cuda and _FORK_HEADS and (_FORK_IN_GRAPH or not torch.cuda.is_current_stream_capturing())
                    and _single_process()):
                cur = torch.cuda.current_stream(data.device)
                aux = []
                for which, (head, fm) in enumerate(((self.heads[0], f16), (self.heads[1], f8))):
                    side = _side_stream(data.device, 1 + which)
                    side.wait_stream(cur)
                    fm.record_stream(side)
                    label.record_stream(side)
                    with torch.cuda.stream(side):
                        aux.append(self.criterion(head(fm), label))
                main = self.criterion(self.heads[-1](fused), label)
                for which, l in enumerate(aux):
                    cur.wait_stream(_side_stream(data.device, 1 + which))
                    l.record_stream(cur)
                return main + aux[0] + aux[1]                  # network.py:108
            aux0 = self.criterion(self.heads[0](f16), label)
            aux1 = self.criterion(self.heads[1](f8), label)
            main = self.criterion(self.heads[-1](fused), label)
            return main + aux0 + aux1                      # network.py:108
        return F.log_softmax(self.heads[-1](fused), dim=1)  # network.py:111

    def logits(self, data):
        """The three full-resolution head outputs (parity checks)."""
        f16, f8, fused = self.features(data)
        return self.heads[0](f16), self.heads[1](f8), self.heads[-1](fused)
