"""GPU parity: FusedSGD (HIP kernel, device-side lr) vs torch.optim.SGD, incl. channels_last
conv weights, per-group lr / weight decay, lr changes between steps, and hipGraph replay."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _models(cuda):
    torch.manual_seed(0)
    def make():
        m = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU(), nn.Conv2d(8, 16, 3, padding=1),
                          nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(16, 5)).to(cuda)
        return m
    a = make(); b = make(); b.load_state_dict(a.state_dict())
    b[3].weight.data = b[3].weight.data.contiguous(memory_format=torch.channels_last)
    return a, b


def _groups(m):
    dec = [p for n, p in m.named_parameters() if p.dim() > 1]
    nod = [p for n, p in m.named_parameters() if p.dim() <= 1]
    return [dict(params=dec, lr=0.05), dict(params=nod, lr=0.5, weight_decay=0.0)]


def test_fused_sgd_matches_torch(cuda):
    from torchseg_amd.optim import FusedSGD
    a, b = _models(cuda)
    oa = torch.optim.SGD(_groups(a), lr=0.05, momentum=0.9, weight_decay=5e-4)
    ob = FusedSGD(_groups(b), lr=0.05, momentum=0.9, weight_decay=5e-4)
    g = torch.Generator(device=cuda).manual_seed(1)
    for step in range(5):
        x = torch.randn(4, 3, 16, 16, device=cuda, generator=g)
        y = torch.randint(0, 5, (4,), device=cuda, generator=g)
        for m, o in ((a, oa), (b, ob)):
            for i, grp in enumerate(o.param_groups):
                grp["lr"] = (0.05 if i == 0 else 0.5) * (1 - step / 10)
            o.zero_grad()
            nn.functional.cross_entropy(m(x), y).backward()
            o.step()
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        torch.testing.assert_close(q, p, rtol=1e-5, atol=1e-6, msg=n)


def test_fused_sgd_step_in_two_parts_equals_one_call(cuda):
    """FusedSGD.step(only=...) (bench.SegmentedStep's early optimizer graph): two calls over disjoint parameter subsets =
    the one-call step bit for bit — parameters, momentum buffers, and the bf16 weight shadows the update refreshes (a
    subset refresh rewrites the shadows of its parameters and leaves the others alone)."""
    from torchseg_amd import shadow
    from torchseg_amd.optim import FusedSGD
    torch.manual_seed(3)
    ws = [torch.randn(32, 16, 3, 3, device=cuda).contiguous(memory_format=torch.channels_last) for _ in range(3)]
    bs = [torch.randn(32, device=cuda) for _ in range(3)]
    gs = [torch.randn_like(t) for t in ws + bs]

    def run(split):
        ps = [nn.Parameter(t.clone(memory_format=torch.preserve_format)) for t in ws + bs]
        opt = FusedSGD([dict(params=ps[:3], lr=0.05), dict(params=ps[3:], lr=0.5, weight_decay=0.0)], lr=0.05,
                       momentum=0.9, weight_decay=5e-4)
        shadows = [shadow.bank.get(p, want_rot=True) for p in ps[:3]]
        for it in range(3):
            for p, g in zip(ps, gs):
                p.grad = g * (it + 1)
            if split:
                early = [ps[0], ps[2], ps[4]]
                opt.step(only=early)
                w0_stale = shadows[1][0].clone()
                opt.step(only=[p for p in ps if all(p is not q for q in early)])
                if it == 0:
                    assert not torch.equal(w0_stale, shadows[1][0]), "the second part refreshes its own shadows"
            else:
                opt.step()
        torch.cuda.synchronize()
        return ([p.detach().clone() for p in ps], [opt.state[p]["momentum_buffer"].clone() for p in ps],
                [(a.clone(), b.clone()) for a, b in shadows])

    one, two = run(False), run(True)
    for a, b in zip(one[0] + one[1], two[0] + two[1]):
        assert torch.equal(a, b)
    for (a0, a1), (b0, b1), p in zip(one[2], two[2], one[0]):
        assert torch.equal(a0, b0) and torch.equal(a1, b1)
        assert torch.equal(a0, p.to(torch.bfloat16))


def test_fused_sgd_under_hip_graph_follows_lr(cuda):
    from torchseg_amd.optim import FusedSGD
    p_eager = torch.randn(1000, device=cuda)
    p_graph = p_eager.clone()
    grad = torch.randn(1000, device=cuda)
    pe = nn.Parameter(p_eager); pg = nn.Parameter(p_graph)
    oe = FusedSGD([pe], lr=0.1, momentum=0.9, weight_decay=1e-3)
    og = FusedSGD([pg], lr=0.1, momentum=0.9, weight_decay=1e-3)
    pe.grad = grad.clone(); pg.grad = grad.clone()
    og.step(); oe.step()                                   # warm-up: creates buffers / lr vector
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(graph):
            og.step()
    torch.cuda.current_stream().wait_stream(s)
    for lr in (0.05, 0.02, 0.3):
        oe.param_groups[0]["lr"] = lr; og.param_groups[0]["lr"] = lr
        og.refresh_lr()
        oe.step(); graph.replay()
    torch.cuda.synchronize()
    torch.testing.assert_close(pg.detach(), pe.detach(), rtol=1e-6, atol=1e-7)


def test_fused_sgd_many_tensors_one_launch(cuda):
    """150 tensors (two launches of the multi-tensor kernel), ragged sizes incl. 1, 4095..4097 and a tensor larger
    than several chunks, three groups, and a gradient that is an unaligned view into a flat bucket."""
    from torchseg_amd.optim import FusedSGD
    g = torch.Generator().manual_seed(2)
    sizes = [1, 2, 3, 5, 4095, 4096, 4097, 8191, 70001] + [int(v) for v in torch.randint(1, 3000, (141,), generator=g)]
    ref = [nn.Parameter(torch.randn(n, generator=g).to(cuda)) for n in sizes]
    our = [nn.Parameter(p.detach().clone()) for p in ref]
    def groups(ps):
        return [dict(params=ps[0::3], lr=0.1), dict(params=ps[1::3], lr=0.02, weight_decay=0.0),
                dict(params=ps[2::3], lr=0.3, momentum=0.5)]
    oa = torch.optim.SGD(groups(ref), lr=0.1, momentum=0.9, weight_decay=5e-4)
    ob = FusedSGD(groups(our), lr=0.1, momentum=0.9, weight_decay=5e-4)
    flat = torch.zeros(sum(sizes) + 1, device=cuda)
    for step in range(3):
        off = 1                                            # +1: every view starts 4 bytes off a 16-byte boundary
        for p, q in zip(ref, our):
            gr = torch.randn(p.numel(), generator=g).to(cuda)
            p.grad = gr.clone()
            if step == 1:
                view = flat[off:off + q.numel()]
                view.copy_(gr)
                q.grad = view
                off += q.numel()
            else:
                q.grad = gr.clone()
        oa.step(); ob.step()
    for i, (p, q) in enumerate(zip(ref, our)):
        torch.testing.assert_close(q, p, rtol=1e-5, atol=1e-6, msg="tensor %d (n=%d)" % (i, sizes[i]))


def test_fused_sgd_state_dict_interchanges_with_torch_sgd(cuda):
    """Checkpoint compatibility (engine.py:103-137 stores optimizer.state_dict()): momentum buffers keep the
    parameter's logical shape even for channels_last weights, torch.optim.SGD loads a FusedSGD state dict and
    FusedSGD loads torch's, and both continue identically."""
    import copy
    from torchseg_amd.optim import FusedSGD
    a, b = _models(cuda)                                   # b[3].weight is channels_last
    oa = torch.optim.SGD(_groups(a), lr=0.05, momentum=0.9, weight_decay=5e-4)
    ob = FusedSGD(_groups(b), lr=0.05, momentum=0.9, weight_decay=5e-4)
    g = torch.Generator(device=cuda).manual_seed(3)

    def one_step(pairs):
        x = torch.randn(4, 3, 16, 16, device=cuda, generator=g)
        y = torch.randint(0, 5, (4,), device=cuda, generator=g)
        for m, o in pairs:
            o.zero_grad()
            nn.functional.cross_entropy(m(x), y).backward()
            o.step()

    for _ in range(2):
        one_step(((a, oa), (b, ob)))
    sa, sb = copy.deepcopy(oa.state_dict()), copy.deepcopy(ob.state_dict())
    for k, st in sb["state"].items():
        assert st["momentum_buffer"].shape == sa["state"][k]["momentum_buffer"].shape
        torch.testing.assert_close(st["momentum_buffer"].contiguous(), sa["state"][k]["momentum_buffer"], rtol=1e-5, atol=1e-6)
    oa2 = torch.optim.SGD(_groups(a), lr=0.05, momentum=0.9, weight_decay=5e-4)
    ob2 = FusedSGD(_groups(b), lr=0.05, momentum=0.9, weight_decay=5e-4)
    oa2.load_state_dict(sb)                                # torch <- ours
    ob2.load_state_dict(sa)                                # ours  <- torch (contiguous buffers for a channels_last weight)
    for _ in range(2):
        one_step(((a, oa2), (b, ob2)))
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        torch.testing.assert_close(q, p, rtol=1e-5, atol=1e-6, msg=n)


def test_ddp_bucket_gather_multi_tensor_copy(cuda):
    """The N>1 gradient path on the GPU: fresh autograd gradients (contiguous, channels_last, 1-D) are gathered into
    the flat all-reduce bucket by one multi-tensor launch, param.grad become views with the parameter's strides,
    a planned parameter without a gradient leaves a zeroed hole, and a second pass reuses the views."""
    from torchseg_amd.ddp import Reducer, _Bucket
    g = torch.Generator().manual_seed(4)
    shapes = [(8, 3, 3, 3), (16, 8, 3, 3), (16,), (5, 16, 1, 1), (5,), (4097,)]
    params = [nn.Parameter(torch.randn(*s, generator=g).to(cuda)) for s in shapes]
    params[1].data = params[1].data.contiguous(memory_format=torch.channels_last)
    bucket = _Bucket(params, cuda)
    for rnd in range(2):
        grads = []
        for i, p in enumerate(params):
            gr = torch.randn(*p.shape, generator=g).to(cuda)
            if i == 1:
                gr = gr.contiguous(memory_format=torch.channels_last)
            grads.append(gr)
            p.grad = None if i == 4 else gr.clone(memory_format=torch.preserve_format)
            bucket.ready[i] = i != 4
        Reducer._gather(bucket)
        torch.cuda.synchronize()
        for i, p in enumerate(params):
            v = bucket.views[i]
            assert v.stride() == p.stride() and v.shape == p.shape
            if i == 4:
                assert p.grad is None and float(v.abs().sum()) == 0.0
            else:
                assert p.grad.data_ptr() == v.data_ptr()
                assert torch.equal(p.grad, grads[i])
        # the slots hold the gradients in the parameters' memory order
        off = bucket.offsets[1]
        assert torch.equal(bucket.flat[off:off + params[1].numel()], grads[1].permute(0, 2, 3, 1).reshape(-1))
    # the averaging factor (1 / world) travels with the copy: fresh gradients are scaled while gathered, gradients
    # accumulated in place in the views (zero_grad(set_to_none=False)) are scaled in place, holes stay zero
    fresh = [torch.randn(*p.shape, generator=g).to(cuda) for p in params]
    fresh[1] = fresh[1].contiguous(memory_format=torch.channels_last)
    for i, p in enumerate(params):
        if i == 4:
            p.grad, bucket.ready[i] = None, False
        elif i == 3:
            p.grad.copy_(fresh[i])                       # stays the bucket view: accumulated in place
            bucket.ready[i] = True
        else:
            p.grad, bucket.ready[i] = fresh[i].clone(memory_format=torch.preserve_format), True
    Reducer._gather(bucket, 0.125)
    torch.cuda.synchronize()
    for i, p in enumerate(params):
        if i == 4:
            assert float(bucket.views[i].abs().sum()) == 0.0
        else:
            assert p.grad.data_ptr() == bucket.views[i].data_ptr()
            assert torch.equal(p.grad, fresh[i] * 0.125)     # power-of-two scale: exact
