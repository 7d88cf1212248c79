"""The path BASELINE.json's north_star names, on the device: the reference's UNCHANGED model/<family>/<experiment>/
network.py + config.py (staged from the checkout, or on the GPU box from the archive tools/stage_reference.py packed:
oracle/_ref/reference_models.tar.gz) imported against OUR furnace/ + apex surface, wrapped in our
DistributedDataParallel exactly as train.py:98-99 does, one training step on cuda:0.

Asserted:
  (a) the fused operators were TAKEN for the verbatim file — provider call counters (K.CallCounter) and the
      interception counters of fusion.FuseMode: the fused up-sampling + OHEM head (tsg_ohem_up_fwd/bwd x 3), the summing
      up-sampling of `fm += last_fm; F.interpolate(fm)` (x 2), the re-classed convolutions (stem, conv64, conv3g/h,
      weight gradients), the SpatialPath chain (stem + BN + next convolution as one node, BN-on-load), and for PSANet the
      collect / distribute contraction (tsg_psa_fwd x 2);
  (b) loss, OHEM kept count and EVERY parameter gradient equal, bit for bit, what our own builder of the same
      architecture (torchseg_amd/workloads, the network bench.py times by default) produces from the same seed — in fp32
      (the parity mode) and in bf16 (the benched mode): the drop-in path runs the same kernels, not similar ones."""
import json
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _dropin import have_staged_reference, run_in, stage  # noqa: E402

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_staged_reference(),
                                                  reason="neither the reference checkout nor its staged archive")]

_BISENET = r'''
import json, os, sys, torch, torch.nn as nn
from config import config            # unchanged reference config.py
from network import BiSeNet          # unchanged reference network.py
from torchseg_amd import fusion, kernels as K
from torchseg_amd.ddp import DistributedDataParallel
from torchseg_amd.losses import ProbOhemCrossEntropy2d
from torchseg_amd.syncbn import SyncBatchNorm
from torchseg_amd.workloads.bisenet import BiSeNet as Native
from utils.init_func import init_weight

dtype = {"fp32": torch.float32, "bf16": torch.bfloat16}["%(dtype)s"]
B, S = %(B)d, %(S)d
dev = torch.device("cuda:0")
min_kept = B * S * S // 16                                     # train.py:48-49

def build(cls):
    torch.manual_seed(config.seed)
    crit = ProbOhemCrossEntropy2d(ignore_label=255, thresh=0.7, min_kept=min_kept, use_weight=False)
    m = cls(config.num_classes, is_training=True, criterion=crit, pretrained_model=None, norm_layer=SyncBatchNorm)
    init_weight(m.business_layer, nn.init.kaiming_normal_, SyncBatchNorm, config.bn_eps, config.bn_momentum,
                mode='fan_in', nonlinearity='relu')                 # train.py:61-63
    return DistributedDataParallel(m.to(dev), compute_dtype=dtype), crit

g = torch.Generator().manual_seed(0)
x = torch.randn(B, 3, S, S, generator=g).to(dev)
y = torch.randint(0, 19, (B, S, S), generator=g)
y[:, :8] = 255
y = y.to(dev)

def step(net, crit):
    net.train()
    before = dict(fusion.stats)
    cc = K.CallCounter(K.provider())
    loss = net(x, y)
    loss.backward()
    torch.cuda.synchronize()
    calls = cc.stop()
    sel = crit.last_selection.cpu()
    return dict(loss=loss.item(), kept=int(sel[1]), calls=calls,
                fuse={k: fusion.stats[k] - before[k] for k in fusion.stats},
                grads=[p.grad.detach().float().cpu() for p in net.module.parameters()],
                bufs=[b.detach().float().cpu() for b in net.module.buffers()])

from torchseg_amd.tuning import use_shipped_miopen_db
use_shipped_miopen_db()
ref_net, ref_crit = build(BiSeNet)
nat_net, nat_crit = build(Native)
assert not getattr(ref_net.module, "tsg_native_fusions", False) and nat_net.module.tsg_native_fusions
assert [k for k, _ in ref_net.module.named_parameters()] == [k for k, _ in nat_net.module.named_parameters()]
assert all(torch.equal(a, b) for a, b in zip(ref_net.module.state_dict().values(), nat_net.module.state_dict().values()))
# Three steps from the same seed: the native builder twice (what differs between THOSE two is run-to-run noise of kernels
# that are not ours: the vendor library's split-K weight gradients of the 1x1 convolutions accumulate with atomics), the
# reference network.py in between.
n0 = step(*build(Native))
r = step(ref_net, ref_crit)
n = step(nat_net, nat_crit)
names = [k for k, _ in ref_net.module.named_parameters()]
noisy = [k for k, a, b in zip(names, n0["grads"], n["grads"]) if not torch.equal(a, b)]
differ = [k for k, a, b in zip(names, r["grads"], n["grads"]) if not torch.equal(a, b)]
rel = {k: float((a - b).norm() / (b.norm() + 1e-30)) for k, a, b in zip(names, r["grads"], n["grads"]) if k in differ}
bdiff = max(float((a - b).abs().max()) for a, b in zip(r["bufs"], n["bufs"]))
print(json.dumps(dict(loss_ref=r["loss"], loss_nat=n["loss"], loss_nat0=n0["loss"], kept_ref=r["kept"], kept_nat=n["kept"],
                      noisy=noisy, differ=differ, rel=rel, bdiff=bdiff, nparams=len(names),
                      calls_ref=r["calls"], calls_nat=n["calls"], fuse_ref=r["fuse"], fuse_nat=n["fuse"],
                      classes=sorted({type(m).__name__ for m in ref_net.module.modules()}))))
'''


def _run(tmp_path, dtype, B, S):
    d = stage(tmp_path, "bisenet", "cityscapes.bisenet.R18")
    return json.loads(run_in(d, _BISENET % dict(dtype=dtype, B=B, S=S), timeout=900).strip().splitlines()[-1])


def _same_kernels(out):
    """The drop-in step reached the C-ABI through exactly the entry points, exactly as often, as our own builder."""
    assert out["calls_ref"] == out["calls_nat"], {k: (out["calls_ref"].get(k), out["calls_nat"].get(k))
                                                  for k in set(out["calls_ref"]) | set(out["calls_nat"])
                                                  if out["calls_ref"].get(k) != out["calls_nat"].get(k)}


def _vendor_wgrad(name):
    """Parameters whose weight gradient is still the vendor library's (1x1 convolutions on full maps)."""
    return name.endswith("downsample.0.weight") or name.endswith("conv_1x1.conv.weight")


def test_reference_bisenet_bf16_takes_the_fused_kernels_and_equals_the_native_builder(tmp_path):
    """At the BENCHED configuration (BASELINE configs[1]: 16 x 1024^2, bf16) — the shapes the shipped find-db of the vendor
    library covers; at untuned shapes its stride-2 forward kernels are not even run-to-run reproducible."""
    out = _run(tmp_path, "bf16", 16, 1024)
    calls, fuse = out["calls_ref"], out["fuse_ref"]
    # (a) interception: network.py:91-95 (`fm += last_fm` + F.interpolate) x 2, network.py:164-166 (head up-sampling) x 3,
    #     network.py:131-137 (SpatialPath chain: stem fused with its successor, two BN-on-load hand-overs)
    assert (fuse["iadd_deferred"], fuse["presum_fused"], fuse["head_deferred"]) == (2, 2, 3), fuse
    assert fuse["iadd_declined_alias"] == 0 and fuse["iadd_declined_not_augmented"] == 0, fuse
    assert (fuse["cbr_stem_fused"], fuse["cbr_fed"]) == (1, 2), fuse
    assert calls.get("ohem_up_fwd") == 3 and calls.get("ohem_up_bwd") == 3, calls
    assert calls.get("upsample_presum_fwd") == 2, calls
    assert calls.get("stem_conv_fwd_stats", 0) >= 2 and calls.get("stem_conv_wrw_bn") == 1, calls   # both 7x7 stems; fused node
    assert calls.get("conv3x3_c64_fwd", 0) >= 6 and calls.get("conv3x3_gen_fwd", 0) >= 20, calls
    assert calls.get("conv3x3_wrw", 0) >= 20 and calls.get("cls_head_fwd") == 3, calls
    assert calls.get("ohem_fwd", 0) == 0 and calls.get("upsample_fwd", 0) == 1, calls     # no unfused head; the one plain
    #                                                  up-sampling is network.py:82-84 (global context, 1x1 -> c5's size)
    for cls in ("WrwConv2d", "StemConv2d"):
        assert cls in out["classes"], out["classes"]
    _same_kernels(out)
    # (b) the same arithmetic as the builder bench.py times, bit for bit: loss, OHEM kept count, every BatchNorm's running
    #     statistics, and every gradient except those two runs of the SAME builder do not reproduce either
    assert out["loss_ref"] == out["loss_nat"] == out["loss_nat0"] and out["kept_ref"] == out["kept_nat"], out
    assert out["bdiff"] == 0.0, out["bdiff"]
    #     Round 6: the full-map 1x1 weight gradients left the vendor library's split-K atomics (torchseg_amd/pwconv.py): two
    #     runs of the benched step now reproduce bit for bit, so nothing is exempted any more.
    assert out["noisy"] == [], out["noisy"]
    assert out["differ"] == [], (out["differ"], out["rel"])
    assert "PointwiseConv2d" in out["classes"], out["classes"]


def test_reference_bisenet_fp32_parity_mode_equals_the_native_builder(tmp_path):
    """fp32 = the parity mode: every convolution on our exact kernels, nothing left to the vendor library — strict equality
    of everything."""
    out = _run(tmp_path, "fp32", 2, 256)
    calls, fuse = out["calls_ref"], out["fuse_ref"]
    assert (fuse["iadd_deferred"], fuse["presum_fused"], fuse["head_deferred"]) == (2, 2, 3), fuse
    assert calls.get("ohem_up_fwd") == 3 and calls.get("upsample_presum_fwd") == 2, calls
    assert calls.get("conv2d_f32_exact_fwd", 0) >= 30, calls           # fp32 = exact convolutions (exactconv.py)
    _same_kernels(out)
    assert out["loss_ref"] == out["loss_nat"] and out["kept_ref"] == out["kept_nat"], out
    assert out["bdiff"] == 0.0 and out["differ"] == [] and out["noisy"] == [], (out["bdiff"], out["differ"], out["noisy"])


_PSANET = r'''
import json, torch, torch.nn as nn
from config import config
import network                         # unchanged reference psanet network.py
from torchseg_amd import fusion, kernels as K
from torchseg_amd.ddp import DistributedDataParallel
from torchseg_amd.syncbn import SyncBatchNorm
dev = torch.device("cuda:0")
torch.manual_seed(1)
crit = nn.CrossEntropyLoss(reduction='mean', ignore_index=-1)      # train.py:48-49
model = network.PSPNet(config.num_classes, criterion=crit, pretrained_model=None, norm_layer=SyncBatchNorm)
for m in model.modules():
    if isinstance(m, nn.Dropout2d):
        m.p = 0.0
net = DistributedDataParallel(model.to(dev))
net.train()
S = 480
x = torch.randn(2, 3, S, S, device=dev)
y = torch.randint(0, config.num_classes, (2, S, S), device=dev)
before = dict(fusion.stats)
cc = K.CallCounter(K.provider())
loss = net(x, y)
loss.backward()
torch.cuda.synchronize()
calls = cc.stop()
nograd = [n for n, p in model.named_parameters() if p.grad is None]
finite = all(bool(torch.isfinite(p.grad).all()) for p in model.parameters() if p.grad is not None)
print(json.dumps(dict(loss=loss.item(), calls=calls, fuse={k: fusion.stats[k] - before[k] for k in fusion.stats},
                      nograd=len(nograd), finite=finite)))
'''


def test_reference_psanet_takes_the_psa_contraction_and_the_ce_kernels(tmp_path):
    d = stage(tmp_path, "psanet", "ade.psanet.R101_v1c")
    out = json.loads(run_in(d, _PSANET, timeout=900).strip().splitlines()[-1])
    calls, fuse = out["calls"], out["fuse"]
    assert fuse["psa_deferred"] == 2 and calls.get("psa_fwd") == 2 and calls.get("psa_bwd") == 2, (fuse, calls)  # network.py:119-137
    assert fuse["ce_fused"] == 2, fuse                     # network.py:50-56: main + aux CE on log_softmax (150 classes: unfused head)
    assert calls.get("ohem_fwd") == 2 and calls.get("ohem_bwd") == 2, calls
    assert out["finite"] and out["loss"] == out["loss"] and 3.0 < out["loss"] < 12.0, out
