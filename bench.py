#!/usr/bin/env python
"""Headline benchmark: training images/sec of BiSeNet-R18 on 1024x1024 synthetic
Cityscapes-shaped crops (BASELINE.json `metric`, configs[1]): bf16 activations,
per-GPU batch 16, SyncBN + OHEM, SGD with the reference's 14 parameter groups.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = zero_grad -> loss = model(imgs, gts) -> backward (incl. bucketed RCCL
gradient all-reduce) -> SGD step, i.e. the body of the reference's loop
(model/bisenet/cityscapes.bisenet.R18/train.py:115-142) without its tqdm/.item()
display.  Inputs are resident in HBM before the timed region.  Weak scaling:
every rank keeps batch 16.  Rank 0 prints ONE JSON line.
"""
import argparse
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


def build_model(device, batch, size, criterion_cls, norm_layer, seed=12345):
    from torchseg_amd.workloads import ensure_furnace_on_path
    ensure_furnace_on_path()
    from torchseg_amd.workloads.bisenet import BiSeNet
    from utils.init_func import group_weight, init_weight
    torch.manual_seed(seed)
    min_kept = int(batch * size * size // 16)                          # train.py:48-49
    criterion = criterion_cls(ignore_label=255, thresh=0.7, min_kept=min_kept, use_weight=False)
    model = BiSeNet(NUM_CLASSES, is_training=True, criterion=criterion, pretrained_model=None,
                    norm_layer=norm_layer)
    init_weight(model.business_layer, nn.init.kaiming_normal_, norm_layer, 1e-5, 0.1,
                mode='fan_in', nonlinearity='relu')                    # train.py:61-63
    base_lr = 1e-2
    groups = []
    groups = group_weight(groups, model.context_path, norm_layer, base_lr)
    for part in (model.spatial_path, model.global_context, model.arms, model.refines, model.heads, model.ffm):
        groups = group_weight(groups, part, norm_layer, base_lr * 10)  # train.py:70-84
    model.to(device)
    opt = torch.optim.SGD(groups, lr=base_lr, momentum=0.9, weight_decay=5e-4)   # train.py:86-89
    return model, opt, base_lr


def synthetic_batch(device, batch, size, seed=0):
    g = torch.Generator(device=device).manual_seed(seed)
    imgs = torch.randn(batch, 3, size, size, generator=g, device=device)
    gts = torch.randint(0, NUM_CLASSES, (batch, size, size), generator=g, device=device)
    gts[:, :8] = 255
    return imgs, gts


def train_step(model, opt, imgs, gts, lr_policy, it, world):
    from utils.pyt_utils import all_reduce_tensor
    opt.zero_grad()
    loss = model(imgs, gts)
    if world > 1:
        all_reduce_tensor(loss, world_size=world)                      # train.py:129-131
    lr = lr_policy.get_lr(it)
    for i, gparam in enumerate(opt.param_groups):
        gparam['lr'] = lr if i < 2 else lr * 10                        # train.py:133-139
    loss.backward()
    opt.step()
    return loss


def cpu_baseline(size, batch=2, steps=2):
    """The oracle (CPU port of the reference path: plain nn.BatchNorm2d + the
    loss_opr restatement) timed on this host's cores.  Bounded sample."""
    from oracle.ohem_ref import ProbOhemCrossEntropy2d as OracleOhem
    from torchseg_amd.workloads import ensure_furnace_on_path
    ensure_furnace_on_path()
    from engine.lr_policy import PolyLR
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    dev = torch.device("cpu")
    model, opt, base_lr = build_model(dev, batch, size, OracleOhem, nn.BatchNorm2d)
    model.train()
    imgs, gts = synthetic_batch(dev, batch, size)
    pol = PolyLR(base_lr, 0.9, 80000)
    train_step(model, opt, imgs, gts, pol, 0, 1)                       # warm-up
    t0 = time.perf_counter()
    for it in range(steps):
        train_step(model, opt, imgs, gts, pol, it + 1, 1)
    dt = time.perf_counter() - t0
    return {"value": round(batch * steps / dt, 3), "unit": "img/s", "cores": cores, "kind": "port",
            "sample": f"{steps} steps (after 1 warm-up) of batch {batch} at {size}x{size}, fp32, torch CPU, "
                      f"oracle BiSeNet-R18 (nn.BatchNorm2d + loss_opr restatement)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--miopen-find", type=int, default=int(os.environ.get("TSG_MIOPEN_FIND", "0")),
                    help="torch.backends.cudnn.benchmark (train.py:35); 0 = immediate mode on the shipped MIOpen find-db (same speed, 100 s faster start)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["TSG_DTYPE"] = args.dtype
    from torchseg_amd.tuning import use_shipped_miopen_db
    use_shipped_miopen_db(rank=rank)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an AMD GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", init_method="env://")
    torch.backends.cudnn.benchmark = bool(args.miopen_find)            # train.py:35

    from torchseg_amd import kernels as K
    from torchseg_amd.ddp import DistributedDataParallel
    from torchseg_amd.losses import ProbOhemCrossEntropy2d
    from torchseg_amd.syncbn import SyncBatchNorm
    from torchseg_amd.workloads import ensure_furnace_on_path
    ensure_furnace_on_path()
    from engine.lr_policy import PolyLR

    model, opt, base_lr = build_model(device, args.batch, args.size, ProbOhemCrossEntropy2d, SyncBatchNorm,
                                      seed=12345 if world == 1 else local_rank)      # train.py:37-40
    model = DistributedDataParallel(model)                             # train.py:98-99
    model.train()
    imgs, gts = synthetic_batch(device, args.batch, args.size, seed=rank)
    pol = PolyLR(base_lr, 0.9, 80 * 1000)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Per-kernel timing brackets every launch with HIP events, which costs host time
    # (~600 event records per step); it must not distort `value`.  So: the warm-up
    # steps run fully instrumented to learn which of our kernels dominates, and the
    # timed region instruments ONLY that kernel (~35 launches per step).
    timer = None
    dominant = None
    for it in range(args.warmup):
        probe = None
        if not args.no_kernel_timing and it == args.warmup - 1:
            probe = K.KernelTimer(K.provider())
        loss = train_step(model, opt, imgs, gts, pol, it, world)
        if probe is not None:
            probe.stop()
            dominant = probe.dominant()
            all_kernels = probe.summary()
    sync()
    if dominant is not None:
        timer = K.KernelTimer(K.provider(), names=[dominant])
    t0 = time.perf_counter()
    for it in range(args.steps):
        loss = train_step(model, opt, imgs, gts, pol, args.warmup + it, world)
    sync()
    dt = time.perf_counter() - t0
    if timer is not None:
        timer.stop()
    final_loss = float(loss.item())
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    if rank == 0:
        global_batch = args.batch * world
        value = global_batch * args.steps / dt
        out = {
            "metric": "training images/sec (1024x1024) BiSeNet-R18",
            "value": round(value, 2), "unit": "img/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"BiSeNet-R18 {args.dtype} batch {args.batch}/GPU {args.size}x{args.size} "
                                   f"synthetic crops, SyncBN + OHEM (BASELINE configs[1])",
                       "global_batch": global_batch, "parallelism": f"dp{world}",
                       "channels_last": model.channels_last, "final_loss": round(final_loss, 4)},
        }
        if timer is not None:
            out["roofline"] = timer.roofline(HBM_PEAK_GBS, os.path.join(ROOT, "profiles"))
            out["kernels_last_warmup_step"] = all_kernels
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.size)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
