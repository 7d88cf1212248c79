"""bench.py's `cpu_baseline` of kind "reference" (SURVEY 8(d) "CPU reference timing"): the REFERENCE'S OWN
code — the unchanged model/bisenet/cityscapes.bisenet.R18/network.py on the reference's own furnace/seg_opr/seg_oprs.py,
base_model/resnet.py, utils/init_func.py and seg_opr/loss_opr.py (ProbOhemCrossEntropy2d; the one-token `1 - valid_mask` ->
`~valid_mask` edit torch >= 1.2 needs is applied IN MEMORY, as tests/golden/make_golden.py does) with nn.BatchNorm2d and
torch.optim.SGD over the 14 parameter groups of train.py:70-89 — timed on this host's cores.

Runs as a process of its own (`python tools/cpu_reference.py --size 1024 --batch 2 ...`, one JSON line on stdout): the
reference's module names (seg_opr, base_model, utils, engine) are the ones our furnace/ answers to in the parent.  The files
come from /root/reference where it exists and from oracle/_ref/reference_models.tar.gz (tools/stage_reference.py, git-ignored,
built by __graft_entry__.build()) on the GPU box.  Nothing under torchseg_amd/ imports this; no GPU is touched."""
import argparse
import json
import os
import sys
import tempfile
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--budget", type=float, default=10.0, help="seconds of timed steps (at least one step)")
    ap.add_argument("--max-steps", type=int, default=5)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--check", action="store_true", help="also print the first step's loss (parity runs)")
    args = ap.parse_args()
    import torch
    import torch.nn as nn
    ncpu = os.cpu_count() or 1
    cores = args.threads or min(ncpu, 64)
    torch.set_num_threads(cores)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import stage_reference
    base = tempfile.mkdtemp(prefix="tsg_cpuref_")
    stage_reference.stage_reference_furnace(base)
    exp_dir = stage_reference.stage(base, "bisenet", "cityscapes.bisenet.R18")
    furnace = os.path.join(base, "TorchSeg", "furnace")
    assert not os.path.islink(furnace)
    sys.path.insert(0, os.path.join(ROOT, "torchseg_amd", "shims"))          # easydict only (config.py:11)
    sys.path.insert(0, furnace)
    os.chdir(exp_dir)
    sys.path.insert(0, exp_dir)
    import utils.pyt_utils  # noqa: F401  (must precede engine.logger: circular import in the reference)
    assert os.path.realpath(utils.pyt_utils.__file__).startswith(os.path.realpath(furnace))
    # loss_opr.py with the `~` edit, in memory, under its own module name
    src = open(os.path.join(furnace, "seg_opr", "loss_opr.py")).read()
    assert src.count("1 - valid_mask") == 2
    mod = types.ModuleType("seg_opr.loss_opr")
    mod.__file__ = os.path.join(furnace, "seg_opr", "loss_opr.py")
    exec(compile(src.replace("1 - valid_mask", "~valid_mask"), mod.__file__, "exec"), mod.__dict__)
    sys.modules["seg_opr.loss_opr"] = mod
    from config import config
    from network import BiSeNet
    from utils.init_func import group_weight, init_weight
    from engine.lr_policy import PolyLR
    import seg_opr.seg_oprs as ref_oprs
    import base_model.resnet as ref_resnet
    for m in (ref_oprs, ref_resnet):
        assert os.path.realpath(m.__file__).startswith(os.path.realpath(furnace)), m.__file__

    B, S = args.batch, args.size
    torch.manual_seed(12345)                                             # config.py:20
    min_kept = int(B * S * S // 16)                                      # train.py:48-49
    criterion = mod.ProbOhemCrossEntropy2d(ignore_label=255, thresh=0.7, min_kept=min_kept, use_weight=False)
    BatchNorm2d = nn.BatchNorm2d
    model = BiSeNet(config.num_classes, is_training=True, criterion=criterion, pretrained_model=None, norm_layer=BatchNorm2d)
    init_weight(model.business_layer, nn.init.kaiming_normal_, BatchNorm2d, config.bn_eps, config.bn_momentum,
                mode='fan_in', nonlinearity='relu')                      # train.py:61-63
    base_lr = config.lr
    params_list = []
    params_list = group_weight(params_list, model.context_path, BatchNorm2d, base_lr)
    for part in (model.spatial_path, model.global_context, model.arms, model.refines, model.heads, model.ffm):
        params_list = group_weight(params_list, part, BatchNorm2d, base_lr * 10)              # train.py:70-84
    optimizer = torch.optim.SGD(params_list, lr=base_lr, momentum=config.momentum, weight_decay=config.weight_decay)
    lr_policy = PolyLR(base_lr, config.lr_power, config.nepochs * config.niters_per_epoch)
    model.train()
    g = torch.Generator().manual_seed(0)                                 # the synthetic batch of bench.py
    imgs = torch.randn(B, 3, S, S, generator=g)
    gts = torch.randint(0, config.num_classes, (B, S, S), generator=g)
    gts[:, :8] = 255

    def step(it):
        optimizer.zero_grad()
        loss = model(imgs, gts)
        lr = lr_policy.get_lr(it)
        for i in range(len(optimizer.param_groups)):
            optimizer.param_groups[i]['lr'] = lr if i < 2 else lr * 10   # train.py:133-139
        loss.backward()
        optimizer.step()
        return loss

    first = float(step(0).item())                                        # warm-up (thread pool, allocator)
    t0 = time.perf_counter()
    done = 0
    while done < args.max_steps and (done < 1 or time.perf_counter() - t0 < args.budget):
        step(done + 1)
        done += 1
    dt = time.perf_counter() - t0
    out = {"value": round(B * done / dt, 3), "unit": "img/s", "cores": cores, "host_cpus": ncpu, "kind": "reference",
           "steps": done, "batch": B, "size": S, "params": sum(p.numel() for p in model.parameters()),
           "source": "reference checkout" if stage_reference.have_reference() else "oracle/_ref/reference_models.tar.gz"}
    if args.check:
        out["first_loss"] = first
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
