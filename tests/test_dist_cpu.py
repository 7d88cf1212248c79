"""CPU, gloo, world_size 2: the N>1 host path (SyncBN exchange steps + bucketed
overlapped gradient averaging) — identical numbers to one process on the
concatenated batch.  The SyncBN math here comes from a tests/-only stand-in
provider (tests/_cpu_provider.py); the HIP provider is covered by -m gpu tests."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Net(nn.Module):
    def __init__(self, norm):
        super().__init__()
        self.c1 = nn.Conv2d(3, 8, 3, padding=1, bias=False)
        self.b1 = norm(8)
        self.c2 = nn.Conv2d(8, 8, 3, padding=1, bias=False)
        self.b2 = norm(8)
        self.unused = nn.Conv2d(8, 4, 1)           # never used: DDP must tolerate (DFN has 5 such tensors)
        self.head = nn.Conv2d(8, 5, 1)

    def forward(self, x, y):
        from torchseg_amd.syncbn import SyncBatchNorm
        h = self.c1(x)
        h = self.b1(h, relu=True) if isinstance(self.b1, SyncBatchNorm) else torch.relu(self.b1(h))
        r = h
        h = self.c2(h)
        h = self.b2(h, residual=r, relu=True) if isinstance(self.b2, SyncBatchNorm) else torch.relu(self.b2(h) + r)
        return nn.functional.cross_entropy(self.head(h), y)


def _worker(rank, world, port, q, rs="0"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      TSG_DDP_RS=rs)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from _cpu_provider import OracleProvider
    from torchseg_amd import kernels as K
    from torchseg_amd.ddp import DistributedDataParallel
    from torchseg_amd.syncbn import SyncBatchNorm
    K._set_provider_for_tests(OracleProvider())
    torch.manual_seed(100 + rank)                 # different init per rank: broadcast must fix it
    net = Net(SyncBatchNorm)
    net.c2.weight.data = net.c2.weight.data.contiguous(memory_format=torch.channels_last)   # strided bucket view
    model = DistributedDataParallel(net, message_size=300)
    g = torch.Generator().manual_seed(7)
    xs = torch.randn(world, 3, 3, 6, 6, generator=g)       # unequal batches are allowed: rank r uses 3 (+1 for rank 1)
    ys = torch.randint(0, 5, (world, 3, 6, 6), generator=g)
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    losses = []
    for step in range(3):                          # step 0 builds the bucket plan, 1-2 use overlapped buckets
        opt.zero_grad()
        loss = model(xs[rank], ys[rank])
        loss.backward()
        opt.step()
        losses.append(loss.item())
    sd = {k: v.clone().numpy() for k, v in model.module.state_dict().items()}   # numpy: no fd-sharing through the queue
    q.put((rank, losses, sd, len(model.reducer.buckets)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("rs", ["0", "1"], ids=["allreduce", "reduce_scatter+all_gather"])
def test_syncbn_and_ddp_world2_match_single_process(rs):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, rs)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, losses, sd, nb = q.get(timeout=180)
        res[r] = (losses, sd, nb)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0

    # single-process reference: plain BatchNorm2d on the concatenated batch; mean loss over ranks
    torch.manual_seed(100)                        # rank 0's init is what gets broadcast
    ref = Net(nn.BatchNorm2d)
    g = torch.Generator().manual_seed(7)
    xs = torch.randn(world, 3, 3, 6, 6, generator=g)
    ys = torch.randint(0, 5, (world, 3, 6, 6), generator=g)
    opt = torch.optim.SGD(ref.parameters(), lr=0.1)
    ref_losses = []
    for step in range(3):
        opt.zero_grad()
        x = xs.reshape(-1, 3, 6, 6)
        h = ref.c1(x); h = torch.relu(ref.b1(h)); r_ = h
        h = torch.relu(ref.b2(ref.c2(h)) + r_)
        logits = ref.head(h)
        per_rank = [nn.functional.cross_entropy(logits[i * 3:(i + 1) * 3], ys[i]) for i in range(world)]
        loss = sum(per_rank) / world
        loss.backward()
        opt.step()
        ref_losses.append([l.item() for l in per_rank])
    assert res[0][2] >= 2                          # message_size=300 elements => several buckets
    for r in range(world):
        for step in range(3):
            assert abs(res[r][0][step] - ref_losses[step][r]) < 2e-5, (r, step)
    sd_ref = ref.state_dict()
    for k, v in sd_ref.items():
        for r in range(world):
            torch.testing.assert_close(torch.from_numpy(res[r][1][k]), v, rtol=2e-4, atol=2e-5, msg=k)
    for k in res[0][1]:
        assert (res[0][1][k] == res[1][1][k]).all(), k   # replicas stay bit-identical


class _FakeLib(object):
    """Stands in for libtsg_hip.so's tsg_comm_* entry points on a CPU box: `fail` names the call that refuses."""

    def __init__(self, fail=None):
        self.fail = fail
        self.created = 0
        self.destroyed = 0

    def tsg_comm_unique_id_bytes(self):
        return 128

    def tsg_comm_get_unique_id(self, buf):
        if self.fail == "get_unique_id":
            return 3
        buf.raw = bytes(range(128))
        return 0

    def tsg_comm_create(self, ident, rank, world, device, out):
        if self.fail == "create":
            return 5
        assert bytes(ident.raw) == bytes(range(128))           # the id rank 0 made arrived on every rank
        self.created += 1
        return 0

    def tsg_comm_destroy(self, handle):
        self.destroyed += 1
        return 0

    def tsg_comm_xgmi_handle_bytes(self):
        return 64

    def tsg_comm_xgmi_export(self, handle, cap, buf):
        return 7 if self.fail == "export" else 0

    def tsg_comm_xgmi_attach(self, handle, allh):
        return 0


def _agree_worker(rank, world, port, q, case):
    """One rank of comm._create_agreed with a failure injected on ONE rank (the round-5 code raised where the
    error happened, so the other rank sat in a collective the failing one had skipped)."""
    import warnings
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    from torchseg_amd import _lib as L
    from torchseg_amd import comm
    bad_rank, what = case
    fake = _FakeLib(what if rank == bad_rank else None)

    def lib():
        if rank == bad_rank and what == "load":
            raise OSError("librccl.so: cannot open shared object file")
        return fake
    L.lib = lib
    if what == "export":
        os.environ["TSG_XGMI_ONESHOT"] = "1"
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        c = comm._create_agreed(None)
    t = torch.ones(1)
    dist.all_reduce(t)                             # the ranks' collective sequences are still aligned
    msgs = [str(x.message) for x in w]
    q.put((rank, c is False, bool(c) and c.one_shot, msgs, fake.created, fake.destroyed, float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", [(0, "load"), (1, "load"), (0, "get_unique_id"), (1, "create"), (1, "export"), (None, None)],
                         ids=["rank0-no-library", "rank1-no-library", "rank0-no-unique-id", "rank1-create-refuses",
                              "rank1-mailbox-export-fails", "all-fine"])
def test_communicator_that_cannot_be_built_falls_back_on_every_rank(case):
    """comm.Comm keeps the collective sequence identical on every rank whatever fails locally, and comm._create_agreed
    returns False on EVERY rank (torch.distributed's collectives, with a warning) when any rank failed: no rank hangs in the
    id broadcast, none splits off."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_agree_worker, args=(r, world, port, q, case)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=120)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    bad_rank, what = case
    for r in range(world):
        is_false, one_shot, msgs, created, destroyed, total = res[r]
        assert total == world
        if what in (None, "export"):
            assert not is_false                    # the communicator exists on both ranks ...
            assert not one_shot                    # ... and a mailbox that one rank cannot export is used by none
            assert created == 1
            if what == "export" and r == bad_rank:
                assert any("mailbox export failed" in m for m in msgs)
        else:
            assert is_false, (case, r)
            assert len(msgs) == 1 and "torch.distributed" in msgs[0], msgs
            if r == bad_rank:
                assert ("librccl" in msgs[0]) if what == "load" else ("tsg_comm" in msgs[0]), msgs
            assert created == destroyed            # a communicator built on the healthy rank is destroyed again


def test_get_does_not_retry_a_failed_communicator(monkeypatch):
    """comm.get stores the agreed `False` and answers None from then on (not one rendezvous attempt per exchange)."""
    from torchseg_amd import comm
    calls = []
    monkeypatch.setattr(comm, "_create_agreed", lambda group: calls.append(1) or False)
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_backend", lambda group=None: "nccl")
    monkeypatch.setattr(comm, "_comms", {})

    class FakeCuda(object):
        is_cuda = True
    assert comm.get(None, like=FakeCuda()) is None and comm.get(None, like=FakeCuda()) is None
    assert len(calls) == 1


def _parity_worker(rank, world, port, q, gather):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      TSG_BN_FP32_GATHER=gather)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from _cpu_provider import OracleProvider
    from torchseg_amd import kernels as K
    from torchseg_amd.syncbn import SyncBatchNorm
    K._set_provider_for_tests(OracleProvider())
    # BiSeNet's global-context BatchNorm across two ranks with ONE image each: [1, C, 1, 1] per rank, values a, b with
    # |a - b| << |a|: var = ((a - b) / 2)^2 against a^2
    g = torch.Generator().manual_seed(11)
    base = 5.0 + torch.rand(8, generator=g, dtype=torch.float64) * 5.0
    delta = (torch.rand(2, 8, generator=g, dtype=torch.float64) - 0.5) * 2e-2
    x = (base + delta[rank]).float().reshape(1, 8, 1, 1).requires_grad_(True)
    bn = SyncBatchNorm(8, eps=1e-5)
    bn.train()
    y = bn(x)
    w = torch.arange(1, 9, dtype=torch.float32).reshape(1, 8, 1, 1) * (1.0 if rank == 0 else -0.5)
    (y * w).sum().backward()
    q.put((rank, y.detach().double().numpy(), x.grad.double().numpy(), bn.running_var.double().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_fp32_statistics_keep_their_fp64_accumulation_across_ranks():
    """in the fp32 parity mode the statistics cross the rank boundary as gathered hi / lo rows
    (syncbn._gather_hilo), so the E[x^2] - mean^2 cancellation of a [B, C, 1, 1] BatchNorm does not return with world > 1;
    the round-5 exchange (an fp32 all-reduce of the rounded sums, TSG_BN_FP32_GATHER=0) loses it."""
    import numpy as np
    world = 2
    res = {}
    for gather in ("1", "0"):
        port = _free_port()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_parity_worker, args=(r, world, port, q, gather)) for r in range(world)]
        for p in procs:
            p.start()
        out = {}
        for _ in range(world):
            r = q.get(timeout=120)
            out[r[0]] = r[1:]
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
        res[gather] = out
    # float64 truth: BatchNorm over the two-image batch of the SAME fp32 inputs
    g = torch.Generator().manual_seed(11)
    base = 5.0 + torch.rand(8, generator=g, dtype=torch.float64) * 5.0
    delta = (torch.rand(2, 8, generator=g, dtype=torch.float64) - 0.5) * 2e-2
    x = torch.stack([(base + delta[r]).float().double() for r in range(2)]).reshape(2, 8, 1, 1).requires_grad_(True)
    bn = nn.BatchNorm2d(8, eps=1e-5).double()
    bn.train()
    y = bn(x)
    w = torch.arange(1, 9, dtype=torch.float64).reshape(1, 8, 1, 1) * torch.tensor([1.0, -0.5]).reshape(2, 1, 1, 1)
    (y * w).sum().backward()

    def err(out):
        ey = max(np.abs(out[r][0] - y[r:r + 1].detach().numpy()).max() for r in range(2))
        eg = max(np.abs(out[r][1] - x.grad[r:r + 1].numpy()).max() / np.abs(x.grad.numpy()).max() for r in range(2))
        return ey, eg
    ey1, eg1 = err(res["1"])
    ey0, eg0 = err(res["0"])
    print("gathered hi/lo: y %.2e dx %.2e   fp32 all-reduce: y %.2e dx %.2e" % (ey1, eg1, ey0, eg0))
    assert ey1 < 5e-4 and eg1 < 5e-4            # |y| ~ 1: what is left is the fp32 rounding of y = a x + b (a ~ 170, b ~ -1300), not of the sums
    assert ey0 > 10 * ey1                        # the case is one where the old exchange does lose the statistics
    for r in range(2):                           # both ranks folded the same rows in the same order
        assert (res["1"][0][2] == res["1"][1][2]).all()
