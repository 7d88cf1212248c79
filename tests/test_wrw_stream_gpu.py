"""The 3x3 weight gradients on a side HIP stream (torchseg_amd.convwrw.wrw_on_side_stream, round 5): same gradients as on
the compute stream, bit for bit, in the cases that decide whether the side stream may be taken at all —
  * one backward pass per step, gradients None before it (the training loop: side stream taken);
  * a second backward pass onto existing gradients (accumulation: AccumulateGrad adds on the compute stream right after
    the node returns, so the node must stay on the compute stream);
  * zero_grad(set_to_none=False); a tensor hook on the weight.
The join: the optimizer (and anything after .backward()) must see finished gradients."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _stack(cuda, seed=0, cl_weights=False):
    from torchseg_amd.convwrw import install_conv_wrw
    torch.manual_seed(seed)
    # shapes whose forward and data-gradient kernels are ours (repeatable bit for bit; the vendor's are not at untuned shapes)
    net = nn.Sequential(nn.Conv2d(64, 64, 3, 1, 1, bias=False), nn.ReLU(), nn.Conv2d(64, 64, 3, 1, 1, bias=False), nn.ReLU(),
                        nn.Conv2d(64, 128, 3, 2, 1, bias=False)).to(cuda)
    for m in net.modules():                             # the layout the training wrapper gives every filter, and the one the
        if isinstance(m, nn.Conv2d):                    # weight-gradient kernels write: AccumulateGrad keeps their tensor
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    assert install_conv_wrw(net) == 3
    return net


def _run(cuda, side, passes=1, set_to_none=True, hook=False, cl_weights=False):
    from torchseg_amd import convwrw
    old = convwrw._WRW_STREAM
    convwrw._WRW_STREAM = side
    try:
        net = _stack(cuda, cl_weights=cl_weights)
        if hook:
            net[0].weight.register_hook(lambda g: g * 1.0)
        g = torch.Generator(device=cuda).manual_seed(1)
        x = torch.randn(8, 64, 64, 64, device=cuda, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        if not set_to_none:
            for p in net.parameters():
                p.grad = torch.zeros_like(p)
        for i in range(passes):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = net(x * (1.0 + i))
            before = torch.cuda.current_stream(cuda)
            y.float().square().mean().backward()
            assert torch.cuda.current_stream(cuda) == before
        grads = [p.grad.detach().clone() for p in net.parameters()]       # on the compute stream, right after backward
        torch.cuda.synchronize()
        return grads
    finally:
        convwrw._WRW_STREAM = old


@pytest.mark.parametrize("passes,set_to_none,hook,cl_weights", [(1, True, False, False), (2, True, False, False),
                                                                (1, False, False, False), (1, True, True, False)])
def test_side_stream_gradients_equal_compute_stream_gradients(cuda, passes, set_to_none, hook, cl_weights):
    want = _run(cuda, False, passes, set_to_none, hook, cl_weights)
    again = _run(cuda, False, passes, set_to_none, hook, cl_weights)
    for a, b in zip(again, want):
        assert torch.equal(a, b), "the kernels themselves are not repeatable here"
    for _ in range(3):                                  # a race would not show every time
        got = _run(cuda, True, passes, set_to_none, hook, cl_weights)
        for a, b in zip(got, want):
            assert torch.equal(a, b)


def test_side_stream_is_taken_only_for_parameters_without_a_gradient(cuda, monkeypatch):
    from torchseg_amd import convwrw
    calls = {"side": 0, "main": 0}
    ptrs = {}
    real = convwrw.wrw_on_side_stream

    def spy(fn, param, *ops, **kw):
        cur = torch.cuda.current_stream(ops[0].device)
        seen = {}

        def wrapped(**fkw):
            seen["s"] = torch.cuda.current_stream(ops[0].device)
            return fn(**fkw)
        out = real(wrapped, param, *ops, **kw)
        calls["side" if seen["s"] != cur else "main"] += 1
        ptrs[id(param)] = out.data_ptr()
        return out
    monkeypatch.setattr(convwrw, "wrw_on_side_stream", spy)
    monkeypatch.setattr(convwrw, "_WRW_STREAM", True)
    net = _stack(cuda)
    x = torch.randn(4, 64, 32, 32, device=cuda).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        net(x).float().sum().backward()
    assert calls == {"side": 3, "main": 0}, calls
    for p in net.parameters():                          # AccumulateGrad kept the kernel's own tensor (no copy on the compute
        assert p.grad.data_ptr() == ptrs[id(p)]         # stream under the running kernel)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        net(x).float().sum().backward()                 # gradients exist now: AccumulateGrad will add on the compute stream
    assert calls == {"side": 3, "main": 3}, calls
    torch.cuda.synchronize()


def _run_twice_in_one_backward(cuda, side):
    """loss = f(net(x1)) + f(net(x2)): every weight is used twice in ONE backward pass."""
    from torchseg_amd import convwrw
    old = convwrw._WRW_STREAM
    convwrw._WRW_STREAM = side
    try:
        net = _stack(cuda)
        g = torch.Generator(device=cuda).manual_seed(3)
        x1 = torch.randn(8, 64, 64, 64, device=cuda, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        x2 = torch.randn(8, 64, 64, 64, device=cuda, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = net(x1).float().square().mean() + net(x2).float().square().mean()
        loss.backward()
        grads = [p.grad.detach().clone() for p in net.parameters()]
        torch.cuda.synchronize()
        return grads
    finally:
        convwrw._WRW_STREAM = old


def test_parameter_used_twice_in_one_backward_pass(cuda):
    """Both uses see `grad is None` when their nodes run; the engine sums the two results on the compute stream.  The second
    use must therefore wait for the side stream and stay on the compute stream (round 5 sent both to the side stream: a race)."""
    want = _run_twice_in_one_backward(cuda, False)
    for _ in range(4):
        got = _run_twice_in_one_backward(cuda, True)
        for a, b in zip(got, want):
            assert torch.equal(a, b)


def test_foreign_post_accumulate_hook_keeps_the_compute_stream(cuda, monkeypatch):
    """An optimizer-in-backward hook reads p.grad as soon as AccumulateGrad has run: such a parameter's weight gradient
    stays on the compute stream; the DDP reducer's own hook (which joins the side stream first) does not count."""
    from torchseg_amd import convwrw
    monkeypatch.setattr(convwrw, "_WRW_STREAM", True)
    net = _stack(cuda)
    seen = {}

    def step_in_backward(p):
        seen["grad"] = p.grad.detach().clone()          # on the compute stream, right now
    net[0].weight.register_post_accumulate_grad_hook(step_in_backward)
    assert convwrw._foreign_grad_hooks(net[0].weight) and not convwrw._foreign_grad_hooks(net[2].weight)
    x = torch.randn(8, 64, 64, 64, device=cuda).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        net(x).float().square().mean().backward()
    torch.cuda.synchronize()
    assert torch.equal(seen["grad"], net[0].weight.grad)
    monkeypatch.setattr(convwrw, "_WRW_STREAM", False)
    ref = _stack(cuda)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ref(x).float().square().mean().backward()
    torch.cuda.synchronize()
    for a, b in zip(net.parameters(), ref.parameters()):
        assert torch.equal(a.grad, b.grad)
