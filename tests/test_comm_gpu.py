"""GPU tests of the tsg_comm_* C-ABI (RCCL on the caller's stream) and of the one-shot mailbox all-reduce.

Only 1-GPU boxes exist for these tests, so: (1) the RCCL entry points run on a 1-rank communicator bootstrapped
exactly like the N-rank one (unique id through the torch.distributed store), on a non-default stream; (2) the
mailbox protocol (peer stores, flags, parity double-buffering, rank-ordered sum) runs between TWO PROCESSES that
share the one GPU through hipIpc handles, which exercises everything except the xGMI wire itself."""
import os
import socket
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rccl_worker(port, q):
    try:
        import torch.distributed as dist
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1)
        from torchseg_amd import comm
        c = comm.get(None, like=torch.empty(1, device="cuda"))
        assert c is not None and c.world == 1 and c.has_rccl
        side = torch.cuda.Stream()
        x = torch.arange(1030, dtype=torch.float32, device="cuda")
        with torch.cuda.stream(side):
            y = x * 2                                   # producer on the side stream ...
            c.all_reduce(y)                             # ... collective on the same stream, no handshake
            c.small_all_reduce(y)
            c.broadcast(y, 0)
            out = torch.empty_like(y)
            c.all_gather(y, out)
            z = out + 1                                 # ... consumer
            b = torch.ones(64, dtype=torch.bfloat16, device="cuda")
            c.all_reduce(b)
        side.synchronize()
        ok = torch.equal(z, x * 2 + 1) and torch.equal(b.float(), torch.ones(64, device="cuda"))
        comm.shutdown()
        dist.destroy_process_group()
        q.put(("ok" if ok else "wrong values", None))
    except Exception as e:                               # noqa: BLE001
        import traceback
        q.put(("exc", traceback.format_exc()))


def test_rccl_entry_points_on_the_callers_stream(cuda):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    status, info = q.get(timeout=240)
    p.join(60)
    assert status == "ok", info


def _mailbox_worker(rank, world, port, q, iters):
    try:
        import torch.distributed as dist
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          HSA_ENABLE_IPC_MODE_LEGACY="0")
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)       # bootstrap only
        from torchseg_amd.comm import Comm
        c = Comm(None, device=0, rccl=False, xgmi=True)
        assert c.one_shot
        g = torch.Generator().manual_seed(123)                             # same stream of test vectors on every rank
        results = []
        for it in range(iters):
            n = [2 * 64 + 2, 2 * 128 + 2, 2 * 512 + 2, 5, 2 * 2048 + 2][it % 5]
            base = torch.randn(world, n, generator=g)
            mine = base[rank].clone().cuda()
            c.small_all_reduce(mine)
            if it % 7 == 0:
                torch.cuda.synchronize()                                   # ranks drift apart in between
            results.append((mine, base))
        torch.cuda.synchronize()
        bad = 0
        outs = []
        for mine, base in results:
            ref = base[0].clone()
            for r in range(1, world):
                ref += base[r]                                             # rank order, fp32: bit-exact expectation
            bad += int(not torch.equal(mine.cpu(), ref))
            outs.append(mine.cpu())
        digest = torch.cat(outs).view(torch.int32).sum(dtype=torch.int64).item()
        dist.barrier()
        c.destroy()
        dist.destroy_process_group()
        q.put((rank, bad, digest, None))
    except Exception:                                                      # noqa: BLE001
        import traceback
        q.put((rank, -1, 0, traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 3])
def test_one_shot_mailbox_allreduce_between_processes_sharing_the_gpu(cuda, world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mailbox_worker, args=(r, world, port, q, 200)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in range(world):
            res.append(q.get(timeout=300))
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.kill()
    for rank, bad, digest, info in res:
        assert bad == 0, (rank, bad, info)
    assert len({d for _, _, d, _ in res}) == 1           # every rank holds bit-identical sums


def _mailbox_graph_worker(rank, world, port, q):
    """The one-shot all-reduce captured ONCE in a hipGraph and replayed: the call counter lives in device memory and is
    advanced by the kernel, so every replay uses a fresh sequence number / parity (round-2 hazard: a host-side counter
    passed by value made the second replay read stale flags)."""
    try:
        import torch.distributed as dist
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          HSA_ENABLE_IPC_MODE_LEGACY="0")
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from torchseg_amd.comm import Comm
        c = Comm(None, device=0, rccl=False, xgmi=True)
        n = 2 * 64 + 2
        g = torch.Generator().manual_seed(321)
        base = torch.randn(40, world, n, generator=g)
        static = torch.zeros(n, device="cuda")
        outs = []

        def eager(i):
            t = base[i][rank].clone().cuda()
            c.small_all_reduce(t)
            outs.append((i, t))

        for i in range(3):                                   # odd number of eager calls first: the graph starts on parity 1
            eager(i)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                c.small_all_reduce(static)                   # captured, not executed
        torch.cuda.current_stream().wait_stream(side)
        for i in range(3, 30):
            static.copy_(base[i][rank])
            graph.replay()
            outs.append((i, static.clone()))
            if i % 5 == 0:
                torch.cuda.synchronize()
        for i in range(30, 34):                              # and eager calls again after the replays
            eager(i)
        torch.cuda.synchronize()
        bad = 0
        for i, t in outs:
            ref = base[i][0].clone()
            for r in range(1, world):
                ref += base[i][r]
            bad += int(not torch.equal(t.cpu(), ref))
        dist.barrier()
        c.destroy()
        dist.destroy_process_group()
        q.put((rank, bad, None))
    except Exception:                                        # noqa: BLE001
        import traceback
        q.put((rank, -1, traceback.format_exc()))


def test_mailbox_under_graph_replay(cuda):
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mailbox_graph_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in range(world):
            res.append(q.get(timeout=300))
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.kill()
    for rank, bad, info in res:
        assert bad == 0, (rank, bad, info)


def _ddp_bucket_worker(port, q, mode, rs):
    """The N > 1 gradient path on a 1-rank RCCL group (TSG_FORCE_COLLECTIVES): buckets all-reduced through tsg_comm on
    the reducer's side stream, fenced by events; gradients must equal the unwrapped model's."""
    try:
        import torch.distributed as dist
        import torch.nn as nn
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
                          TSG_FORCE_COLLECTIVES="1", TSG_DDP_COMM=mode, TSG_DDP_RS=rs, TSG_DTYPE="fp32",
                          TSG_CHANNELS_LAST="0")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1)
        from torchseg_amd import comm
        from torchseg_amd.ddp import DistributedDataParallel

        def make():
            torch.manual_seed(5)
            return nn.Sequential(nn.Conv2d(3, 16, 3, padding=1), nn.ReLU(), nn.Conv2d(16, 16, 3, padding=1), nn.ReLU(),
                                 nn.Conv2d(16, 7, 1)).cuda()
        x = torch.randn(4, 3, 32, 32, device="cuda")
        ref = make()
        ref(x).square().mean().backward()
        model = DistributedDataParallel(make(), message_size=1000)
        for step in range(3):                                # step 0 builds the plan, 1-2 launch from the hooks
            for p in model.parameters():
                p.grad = None
            model(x).square().mean().backward()
            torch.cuda.synchronize()
            for (n, p), pr in zip(model.module.named_parameters(), ref.parameters()):
                torch.testing.assert_close(p.grad, pr.grad, rtol=1e-5, atol=1e-6, msg=f"step {step} {n}")
        red = model.reducer
        used = red._comm is not None and red._side is not None and len(red.buckets) >= 2
        comm.shutdown()
        dist.destroy_process_group()
        q.put(("ok" if used else "buckets did not go through tsg_comm", None))
    except Exception:                                        # noqa: BLE001
        import traceback
        q.put(("exc", traceback.format_exc()))


@pytest.mark.parametrize("mode,rs", [("shared", "0"), ("shared", "1"), ("separate", "0")])
def test_ddp_buckets_through_tsg_comm_on_a_side_stream(cuda, mode, rs):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_ddp_bucket_worker, args=(_free_port(), q, mode, rs))
    p.start()
    status, info = q.get(timeout=300)
    p.join(60)
    assert status == "ok", info


def _parity_gather_worker(port, q):
    """fp32 SyncBatchNorm on the N > 1 code path of one rank (TSG_FORCE_COLLECTIVES=1): the statistics cross the "rank
    boundary" as gathered hi / lo rows through tsg_comm_allgather and keep their fp64 accumulation."""
    try:
        import torch.distributed as dist
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
                          TSG_FORCE_COLLECTIVES="1")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1)
        from torchseg_amd import syncbn
        from torchseg_amd.syncbn import SyncBatchNorm
        g = torch.Generator().manual_seed(11)
        base = 5.0 + torch.rand(128, generator=g, dtype=torch.float64) * 5.0
        delta = (torch.rand(2, 128, generator=g, dtype=torch.float64) - 0.5) * 2e-2
        x32 = (base + delta).float().reshape(2, 128, 1, 1)
        w = (torch.arange(1, 129, dtype=torch.float64) / 64.0).reshape(1, 128, 1, 1) * torch.tensor([1.0, -0.5]).reshape(2, 1, 1, 1)
        ref = torch.nn.BatchNorm2d(128, eps=1e-5).double()
        ref.train()
        xr = x32.double().requires_grad_(True)
        yr = ref(xr)
        (yr * w).sum().backward()
        errs = {}
        for gather in (True, False):
            syncbn._FP32_GATHER = gather
            bn = SyncBatchNorm(128, eps=1e-5).cuda()
            bn.train()
            x = x32.cuda().requires_grad_(True)
            y = bn(x)
            (y * w.float().cuda()).sum().backward()
            torch.cuda.synchronize()
            errs[gather] = (float((y.double().cpu() - yr).abs().max()),
                            float((x.grad.double().cpu() - xr.grad).abs().max() / xr.grad.abs().max()))
        from torchseg_amd import comm
        comm.shutdown()
        dist.destroy_process_group()
        q.put(("ok", errs))
    except Exception:                                    # noqa: BLE001
        import traceback
        q.put(("exc", traceback.format_exc()))


def test_fp32_statistics_gathered_as_hi_lo_rows_on_the_forced_path(cuda):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_parity_gather_worker, args=(_free_port(), q))
    p.start()
    status, info = q.get(timeout=240)
    p.join(60)
    assert status == "ok", info
    print("gathered hi/lo (y, dx):", info[True], "  fp32 all-reduce:", info[False])
    assert info[True][0] < 5e-4 and info[True][1] < 5e-4
    assert info[False][0] > 5 * info[True][0]            # the exchange of rounds 1-5 loses the statistics on this case
