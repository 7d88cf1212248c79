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

    def __init__(self, model, opt, imgs, gts, world, opt_inside=False):
        self.graph = torch.cuda.CUDAGraph()
        self.opt = opt
        self.opt_inside = bool(opt_inside)
        opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph, stream=self.capture_stream()):
            self.loss = step_body(model, opt, imgs, gts, world, with_optimizer=self.opt_inside)

    def __call__(self):
        self.graph.replay()
        if not self.opt_inside:
            self.opt.step()
        return self.loss


def _cpu_leg(size, batch, cores, budget_s, max_steps):
    """img/s of the oracle network (1 untimed warm-up step, then steps until `budget_s` or `max_steps`)."""
    from oracle.ohem_ref import ProbOhemCrossEntropy2d as OracleOhem
    from engine.lr_policy import PolyLR
    dev = torch.device("cpu")
    model, opt, base_lr = build_model(dev, batch, size, OracleOhem, nn.BatchNorm2d)
    model.train()
    imgs, gts = synthetic_batch(dev, batch, size)
    pol = PolyLR(base_lr, 0.9, 80000)
    train_step(model, opt, imgs, gts, pol, 0, 1)                       # warm-up (thread pool, allocator)
    t0 = time.perf_counter()
    done = 0
    while done < max_steps and (done < 1 or time.perf_counter() - t0 < budget_s):
        train_step(model, opt, imgs, gts, pol, done + 1, 1)
        done += 1
    dt = time.perf_counter() - t0
    return round(batch * done / dt, 3), done


def cpu_baseline(headline=True):
    """The oracle (CPU port of the reference path: the reference's BiSeNet-R18 architecture with plain
    nn.BatchNorm2d + the loss_opr.py restatement + torch.optim.SGD, fp32) timed on this host's cores.  Bounded
    sample, ~10-30 s of CPU work: `value` is the headline 1024 x 1024 shape with the batch reduced to 2 (batch 1
    is not a legal training batch: the global-context BN sees [B,128,1,1]); BASELINE configs[0]'s own 2 x 512 x 512
    shape is reported next to it.  The reference's train.py itself cannot run (apex, cv2, .next()), and
    /root/reference does not exist on the GPU box, hence kind = "port"."""
    from torchseg_amd.workloads import ensure_furnace_on_path
    ensure_furnace_on_path()
    cores = min(os.cpu_count() or 1, 64)       # torch's CPU conv stops scaling (and oversubscribes) beyond this
    torch.set_num_threads(cores)
    v512, n512 = _cpu_leg(512, 2, cores, 6.0, 5)
    out = {"value": v512, "unit": "img/s", "cores": cores, "kind": "port",
           "sample": f"{n512} steps (after 1 warm-up) of batch 2 at 512x512 (BASELINE configs[0] shape), fp32, "
                     f"torch CPU, oracle BiSeNet-R18 (nn.BatchNorm2d + loss_opr restatement)"}
    if headline:
        v1024, n1024 = _cpu_leg(1024, 2, cores, 15.0, 3)
        out = {"value": v1024, "unit": "img/s", "cores": cores, "kind": "port",
               "sample": f"{n1024} steps (after 1 warm-up) of batch 2 at 1024x1024 (the headline crop of BASELINE "
                         f"configs[1]; batch reduced from 16 to bound the sample), fp32, torch CPU, oracle "
                         f"BiSeNet-R18 (nn.BatchNorm2d + loss_opr restatement + torch.optim.SGD)",
               "configs0": {"value": v512, "unit": "img/s", "sample": out["sample"]}}
    return out


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


def launch_check(world, rank):
    """--launch-check: rendezvous only (gloo when there is no GPU), so that the launcher logic is testable on a CPU
    box: every rank joins, the world size is all-reduced, rank 0 prints it."""
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
        print(json.dumps({"launch_check": ok, "n_gpus": n, "backend": backend}), flush=True)
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)       # >= 20 warm-up + >= 50 timed steps
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-headline", type=int, default=1,
                    help="also time ONE CPU step at the headline 1024x1024 shape (batch 2; about a minute of host time)")
    ap.add_argument("--launch-check", action="store_true", help="rendezvous of --gpus N ranks only, no GPU work (CPU-testable)")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("TSG_GRAPH", "0")),
                    help="1: replay zero_grad+forward+backward from a hipGraph, optimizer eager after each replay; "
                         "2: the FusedSGD step is captured too (default 0 = eager)")
    ap.add_argument("--optimizer", default="fused", choices=["fused", "torch"])
    ap.add_argument("--trace-loss", action="store_true", help="debug: print the loss of every timed step (syncs)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--miopen-find", type=int, default=int(os.environ.get("TSG_MIOPEN_FIND", "0")),
                    help="torch.backends.cudnn.benchmark (train.py:35); 0 = immediate mode on the shipped MIOpen find-db (same speed, 100 s faster start)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))             # plain `python bench.py --gpus N`
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.launch_check:
        sys.exit(launch_check(world, rank))
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
    # Graph mode: the eager warm-up, the capture, every replay and the optimizer run on ONE stream (GraphedStep.stream).
    run_stream = GraphedStep.capture_stream() if use_graph else None
    if run_stream is not None:
        run_stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(run_stream) if run_stream is not None else contextlib.nullcontext():
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
        sync()
        graphed = None
        if use_graph:
            graphed = GraphedStep(model, opt, imgs, gts, world, opt_inside=args.graph == 2)
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
    if run_stream is not None:
        torch.cuda.current_stream().wait_stream(run_stream)
    if timer is not None:
        timer.stop()
    elif dominant is not None:
        timer = roof_probe
    final_loss = float(loss.item())
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
            "metric": "training images/sec (1024x1024) BiSeNet-R18",
            "value": round(value, 2), "unit": "img/s", "n_gpus": n_ranks, "steps": args.steps,
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
            out["cpu_baseline"] = cpu_baseline(headline=bool(args.cpu_headline))
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
