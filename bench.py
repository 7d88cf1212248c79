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


def build_model(device, batch, size, criterion_cls, norm_layer, seed=12345, fused_sgd=False):
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
    if fused_sgd:
        from torchseg_amd.optim import FusedSGD                        # same update, one HIP kernel per tensor
        opt = FusedSGD(groups, lr=base_lr, momentum=0.9, weight_decay=5e-4)
    else:
        opt = torch.optim.SGD(groups, lr=base_lr, momentum=0.9, weight_decay=5e-4)   # train.py:86-89
    return model, opt, base_lr


def synthetic_batch(device, batch, size, seed=0):
    g = torch.Generator(device=device).manual_seed(seed)
    imgs = torch.randn(batch, 3, size, size, generator=g, device=device)
    gts = torch.randint(0, NUM_CLASSES, (batch, size, size), generator=g, device=device)
    gts[:, :8] = 255
    return imgs, gts


def set_lr(opt, lr_policy, it):
    lr = lr_policy.get_lr(it)
    for i, gparam in enumerate(opt.param_groups):
        gparam['lr'] = lr if i < 2 else lr * 10                        # train.py:133-139
    if hasattr(opt, "refresh_lr"):
        opt.refresh_lr()                                               # device-side lr for graph replay


def step_body(model, opt, imgs, gts, world, with_optimizer=True):
    from utils.pyt_utils import all_reduce_tensor
    opt.zero_grad()
    loss = model(imgs, gts)
    if world > 1 or dist.is_initialized():
        all_reduce_tensor(loss, world_size=world)                      # train.py:129-131
    loss.backward()
    if with_optimizer:
        opt.step()
    return loss


def train_step(model, opt, imgs, gts, lr_policy, it, world):
    set_lr(opt, lr_policy, it)
    return step_body(model, opt, imgs, gts, world)


class GraphedStep(object):
    """zero_grad -> forward -> backward (incl. collectives) captured once into a hipGraph and
    replayed, which removes ~1000 host-side launches per step; optimizer.step() stays eager
    after each replay (62 launches), so the reference's per-iteration lr schedule
    (train.py:133-139) needs no special handling."""

    def __init__(self, model, opt, imgs, gts, world):
        self.graph = torch.cuda.CUDAGraph()
        self.opt = opt
        opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.loss = step_body(model, opt, imgs, gts, world, with_optimizer=False)

    def __call__(self):
        self.graph.replay()
        self.opt.step()
        return self.loss


def cpu_baseline(size=512, batch=2, steps=1):
    """The oracle (CPU port of the reference path: plain nn.BatchNorm2d + the
    loss_opr restatement) timed on this host's cores.  Bounded sample: BASELINE
    configs[0]'s shape (2 x 512 x 512); at the headline 1024 x 1024 one CPU step of
    batch 2 takes ~2 minutes (measured 0.016 img/s on a 256-thread host), far beyond
    the 10-30 s budget of this leg."""
    from oracle.ohem_ref import ProbOhemCrossEntropy2d as OracleOhem
    from torchseg_amd.workloads import ensure_furnace_on_path
    ensure_furnace_on_path()
    from engine.lr_policy import PolyLR
    cores = min(os.cpu_count() or 1, 64)       # torch's CPU conv stops scaling (and oversubscribes) beyond this
    torch.set_num_threads(cores)
    dev = torch.device("cpu")
    model, opt, base_lr = build_model(dev, batch, size, OracleOhem, nn.BatchNorm2d)
    model.train()
    imgs, gts = synthetic_batch(dev, batch, size)
    pol = PolyLR(base_lr, 0.9, 80000)
    train_step(model, opt, imgs, gts, pol, 0, 1)                       # warm-up
    t0 = time.perf_counter()
    done = 0
    while done < 5 and (done < steps or time.perf_counter() - t0 < 10.0):   # ~10-20 s of CPU work
        train_step(model, opt, imgs, gts, pol, done + 1, 1)
        done += 1
    dt = time.perf_counter() - t0
    return {"value": round(batch * done / dt, 3), "unit": "img/s", "cores": cores, "kind": "port",
            "sample": f"{done} steps (after 1 warm-up) of batch {batch} at {size}x{size} (BASELINE configs[0] shape), "
                      f"fp32, torch CPU, oracle BiSeNet-R18 (nn.BatchNorm2d + loss_opr restatement)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("TSG_GRAPH", "0")),
                    help="EXPERIMENTAL: replay zero_grad+forward+backward from a hipGraph (default off)")
    ap.add_argument("--optimizer", default="fused", choices=["fused", "torch"])
    ap.add_argument("--trace-loss", action="store_true", help="debug: print the loss of every timed step (syncs)")
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
    if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
        os.environ["NCCL_DEBUG"] = "WARN"          # keep RCCL's banner off stdout (one JSON line contract)
    os.environ["TSG_DTYPE"] = args.dtype
    from torchseg_amd.tuning import use_shipped_miopen_db
    use_shipped_miopen_db(rank=rank)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an AMD GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    force_coll = os.environ.get("TSG_FORCE_COLLECTIVES", "0") == "1"     # 1-rank run of the N>1 code path
    if world > 1 or force_coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", init_method="env://")
    torch.backends.cudnn.benchmark = bool(args.miopen_find)            # train.py:35

    from torchseg_amd import kernels as K
    from torchseg_amd.ddp import DistributedDataParallel
    from torchseg_amd.losses import ProbOhemCrossEntropy2d
    from torchseg_amd.syncbn import SyncBatchNorm
    from torchseg_amd.workloads import ensure_furnace_on_path
    ensure_furnace_on_path()
    from engine.lr_policy import PolyLR

    use_graph = bool(args.graph)
    model, opt, base_lr = build_model(device, args.batch, args.size, ProbOhemCrossEntropy2d, SyncBatchNorm,
                                      seed=12345 if world == 1 else local_rank,       # train.py:37-40
                                      fused_sgd=args.optimizer == "fused")
    model = DistributedDataParallel(model)                             # train.py:98-99
    model.train()
    imgs, gts = synthetic_batch(device, args.batch, args.size, seed=rank)
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
    side = torch.cuda.Stream() if use_graph else None
    if side is not None:
        side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
        for it in range(n_eager):
            probe = None
            if not args.no_kernel_timing and it == n_eager - 1:
                probe = K.KernelTimer(K.provider())
            loss = train_step(model, opt, imgs, gts, pol, it, world)
            if probe is not None:
                probe.stop()
                dominant = probe.dominant()
                all_kernels = probe.summary()
                roof_probe = probe
    if side is not None:
        torch.cuda.current_stream().wait_stream(side)
    sync()
    graphed = None
    if use_graph:
        graphed = GraphedStep(model, opt, imgs, gts, world)
        for it in range(2):                                            # untimed replays
            set_lr(opt, pol, n_eager + it)
            loss = graphed()
        sync()
    elif dominant is not None:
        timer = K.KernelTimer(K.provider(), names=[dominant])
    t0 = time.perf_counter()
    for it in range(args.steps):
        if graphed is not None:
            set_lr(opt, pol, args.warmup + it)
            loss = graphed()
        else:
            loss = train_step(model, opt, imgs, gts, pol, args.warmup + it, world)
        if args.trace_loss and rank == 0:
            print("step", it, "loss", float(loss.item()), "lr", opt.param_groups[0]["lr"], file=sys.stderr, flush=True)
    sync()
    dt = time.perf_counter() - t0
    if timer is not None:
        timer.stop()
    elif dominant is not None:
        timer = roof_probe
    final_loss = float(loss.item())
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    out = None
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
                       "channels_last": model.channels_last, "final_loss": round(final_loss, 4),
                       "hip_graph": bool(use_graph), "optimizer": args.optimizer},
        }
        if timer is not None:
            out["roofline"] = timer.roofline(HBM_PEAK_GBS, os.path.join(ROOT, "profiles"))
            out["roofline"]["measured_over"] = ("instrumented eager warm-up step (hipGraph replay has no host-side "
                                                "launches to bracket)") if use_graph else "timed region"
            out["kernels_last_warmup_step"] = all_kernels
        if args.dtype == "bf16" and args.size == 1024:
            # whole-step HBM roofline of SURVEY.md 8(d): ~3.4 GB of algorithmic traffic per image in bf16 (our
            # kernels 2.16 GB + the convolutions' operands once each + pools) => 2350 img/s per GPU at 8 TB/s
            gb_per_img = 3.4
            per_gpu = value / world
            out["step_roofline"] = {"bound": "hbm", "algo_GB_per_img": gb_per_img,
                                    "achieved": round(per_gpu * gb_per_img, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(per_gpu * gb_per_img / HBM_PEAK_GBS, 4), "target_frac": 0.70}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
    if world > 1 or force_coll:
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
