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


def test_communicator_that_cannot_be_built_falls_back_on_every_rank(tmp_path):
    """comm._create_agreed: a tsg_comm that fails to build (no librccl, ncclCommInitRank refusing) is agreed on through the
    process group; every rank then gets None from comm.get (torch.distributed's collectives), once, with a warning."""
    import subprocess
    import sys
    script = tmp_path / "agree.py"
    script.write_text('''
import socket, sys, warnings
import torch, torch.distributed as dist
sys.path.insert(0, %r)
from torchseg_amd import comm
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%%d" %% port, rank=0, world_size=1)
calls = []
class Broken(object):
    def __init__(self, group=None, **kw):
        calls.append(1)
        raise RuntimeError("librccl.so: cannot open shared object file")
comm.Comm = Broken
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    assert comm._create_agreed(None) is False
assert len(w) == 1 and "torch.distributed" in str(w[0].message) and "librccl" in str(w[0].message)
comm._comms[None] = False                  # what get() stores
class FakeCuda(object):
    is_cuda = True
dist.get_backend = lambda group=None: "nccl"
assert comm.get(None, like=FakeCuda()) is None and comm.get(None, like=FakeCuda()) is None
assert len(calls) == 1                     # not retried on every exchange
comm.shutdown()
dist.destroy_process_group()
print("agreed-fallback-ok")
''' % ROOT)
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "agreed-fallback-ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
