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

    def spy(fn, param, *ops):
        cur = torch.cuda.current_stream(ops[0].device)
        seen = {}

        def wrapped():
            seen["s"] = torch.cuda.current_stream(ops[0].device)
            return fn()
        out = real(wrapped, param, *ops)
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
