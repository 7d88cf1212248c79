"""CPU: host-side logic that needs no kernel — module re-classing and eligibility, the layout predicates the
optimizer / DDP gather rely on, bucket views, FusedSGD's torch-compatible group keys."""
import torch
import torch.nn as nn


def test_conv_module_swaps_select_only_the_intended_layers_and_pass_through_on_cpu():
    from torchseg_amd.convwrw import WrwConv2d, install_conv_wrw
    from torchseg_amd.stemconv import StemConv2d, install_stem_conv
    torch.manual_seed(0)
    net = nn.Sequential(
        nn.Conv2d(3, 64, 7, 2, 3, bias=False),          # image stem            -> StemConv2d
        nn.Conv2d(64, 64, 3, 1, 1, bias=False),         # layer1-style 3x3      -> WrwConv2d
        nn.Conv2d(64, 64, 3, 2, 1, bias=False),         # stride 2 (down-sampling block) -> WrwConv2d
        nn.Conv2d(64, 64, 3, 1, 2, dilation=2, bias=False),   # dilated          -> untouched
        nn.Conv2d(64, 64, 3, 1, 1, bias=True),          # biased                -> untouched
        nn.Conv2d(3, 64, 7, 2, 3, bias=True),           # biased stem           -> untouched
    )
    ref = [m.weight.detach().clone() for m in net]
    keys = list(net.state_dict().keys())
    assert install_stem_conv(net) == 1 and install_conv_wrw(net) == 2
    assert [type(m) for m in net] == [StemConv2d, WrwConv2d, WrwConv2d, nn.Conv2d, nn.Conv2d, nn.Conv2d]
    assert list(net.state_dict().keys()) == keys                       # same parameters, same names
    assert all(torch.equal(m.weight, w) for m, w in zip(net, ref))
    x = torch.randn(1, 3, 32, 32)
    y0 = nn.functional.conv2d(x, ref[0], None, 2, 3)
    y1 = net[0](x)                                                      # CPU tensor: the stock convolution
    assert torch.equal(y0, y1)
    h = torch.randn(1, 64, 8, 8, requires_grad=True)
    z = net[1](h)
    assert torch.equal(z, nn.functional.conv2d(h, ref[1], None, 1, 1))
    z.sum().backward()
    assert net[1].weight.grad is not None and h.grad is not None


def test_same_dense_order_predicate():
    from torchseg_amd.optim import _same_dense_order
    a = torch.randn(8, 4, 3, 3)
    assert _same_dense_order(a, a.clone())
    assert not _same_dense_order(a, a.contiguous(memory_format=torch.channels_last))
    w = torch.randn(8, 4, 1, 1)
    assert _same_dense_order(w, w.contiguous(memory_format=torch.channels_last))      # 1x1: identical memory order
    assert _same_dense_order(w.contiguous(memory_format=torch.channels_last), w)
    assert not _same_dense_order(w, torch.randn(8, 4, 1, 2)[..., :1])                  # not dense


def test_bucket_views_follow_the_parameter_layout():
    from torchseg_amd.ddp import _Bucket
    p0 = nn.Parameter(torch.randn(6, 4, 3, 3))
    p1 = nn.Parameter(torch.randn(6, 4, 3, 3).contiguous(memory_format=torch.channels_last))
    p2 = nn.Parameter(torch.randn(7))
    b = _Bucket([p0, p1, p2], torch.device("cpu"))
    assert all(o % 4 == 0 for o in b.offsets)                          # 16-byte aligned slots
    v0, v1, v2 = b.view(0), b.view(1), b.view(2)
    assert v0.stride() == p0.stride() and v1.stride() == p1.stride() and v2.shape == p2.shape
    v1.copy_(p1.detach())
    flat = b.flat[b.offsets[1]:b.offsets[1] + p1.numel()]
    assert torch.equal(flat, p1.detach().permute(0, 2, 3, 1).reshape(-1))   # the slot holds the parameter's memory order
    v0.fill_(1.0); v2.fill_(2.0)
    assert abs(float(b.flat.double().sum()) - (float(v1.double().sum()) + p0.numel() + 2.0 * p2.numel())) < 1e-4   # no overlap


def test_fused_sgd_group_keys_match_torch_sgd():
    from torchseg_amd.optim import FusedSGD
    p = [nn.Parameter(torch.randn(3))]
    ours = FusedSGD(p, lr=0.1, momentum=0.9, weight_decay=1e-4)
    theirs = torch.optim.SGD(p, lr=0.1, momentum=0.9, weight_decay=1e-4)
    assert set(theirs.param_groups[0].keys()) <= set(ours.param_groups[0].keys())
    theirs.load_state_dict(ours.state_dict())
    ours.load_state_dict(theirs.state_dict())
    assert ours.param_groups[0]["momentum"] == 0.9 and ours.param_groups[0]["nesterov"] is False


def test_cbr_chain_and_bn_relu_conv_are_the_module_sequence_on_cpu():
    """seg_oprs.cbr_chain / convwrw.bn_relu_conv only re-associate BN + ReLU with the NEXT convolution on the HIP path; on
    CPU tensors (and for anything the conv64 kernels do not cover) they must be exactly the modules called in order."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "torchseg_amd", "furnace"))
    from seg_opr.seg_oprs import ConvBnRelu, cbr_chain
    from torchseg_amd.convwrw import bn_relu_conv
    torch.manual_seed(2)
    mods = [ConvBnRelu(3, 64, 7, 2, 3), ConvBnRelu(64, 64, 3, 2, 1), ConvBnRelu(64, 64, 3, 2, 1),
            ConvBnRelu(64, 128, 1, 1, 0, has_relu=False)]
    x = torch.randn(2, 3, 40, 32)
    ref = x
    for m in mods:
        ref = m(ref)
    for m in mods:                                   # the reference pass updated the running statistics: rewind them
        m.bn.reset_running_stats()
    out = cbr_chain(mods, x)
    assert torch.equal(out, ref)
    bn, relu, conv = nn.BatchNorm2d(64), nn.ReLU(), nn.Conv2d(64, 64, 3, 1, 1, bias=False)
    h = torch.randn(2, 64, 9, 7)
    want = conv(relu(bn(h)))
    bn.reset_running_stats()
    assert torch.equal(bn_relu_conv(bn, relu, h, conv), want)


def test_native_builders_defer_only_hip_logits_and_shadows_refuse_cpu_parameters():
    import pytest
    from torchseg_amd import workloads
    from torchseg_amd._lib import TsgError
    from torchseg_amd.shadow import _Bank
    z = torch.randn(1, 19, 4, 4)
    up = workloads.upsample_logits(z, scale=8)
    assert isinstance(up, torch.Tensor) and tuple(up.shape) == (1, 19, 32, 32)
    assert torch.equal(up, nn.functional.interpolate(z, scale_factor=8, mode="bilinear", align_corners=True))
    with pytest.raises(TsgError):
        _Bank().register(nn.Parameter(torch.randn(64, 64, 3, 3)), want_rot=True)


def test_wrw_tile_coordinates_advance_with_carries():
    """conv3wrw.hip (BUF fetch): a block's tiles are slot, slot + bpp, ...; the kernel keeps (tile column, tile row,
    image) and adds the constant step (bpp % tiles_w, (bpp / tiles_w) % tiles_h, bpp / (tiles_w tiles_h)) with one carry
    per digit instead of dividing per tile.  Same arithmetic here against the divisions, over random geometries
    (bpp as w3gen_geom makes it: <= ntiles, rounded up to a multiple of 8)."""
    import random
    rnd = random.Random(3)
    for _ in range(3000):
        tw, th, B = rnd.randint(1, 40), rnd.randint(1, 300), rnd.randint(1, 16)
        ntiles = B * tw * th
        npairs = rnd.choice([1, 2, 4, 8, 16, 64])
        bpp = (rnd.choice([256, 512]) + npairs - 1) // npairs
        bpp = (max(1, min(bpp, ntiles)) + 7) // 8 * 8
        slot = rnd.randrange(bpp)
        dtw, dth, db = bpp % tw, (bpp // tw) % th, bpp // (tw * th)
        ftw, fth, fb = slot % tw, (slot // tw) % th, slot // (tw * th)
        t = slot
        for _ in range(24):
            assert (ftw, fth, fb) == (t % tw, (t // tw) % th, t // (tw * th))
            ftw += dtw
            if ftw >= tw:
                ftw -= tw; fth += 1
            fth += dth
            if fth >= th:
                fth -= th; fb += 1
            fb += db
            t += bpp


def test_wrw_buffer_fetch_addresses_equal_pointer_fetch():
    """conv3wrw.hip: the buffer-load fetch (descriptor base per tile + a thread's constant 32-bit byte offsets, invalid
    positions out of bounds) against the pointer fetch (full address per load, predicated), restated in Python over
    random geometries, tiles and threads: same validity, same byte address, offsets below the 2 GB the descriptor covers
    (the unsigned-compare form of the halo predicate included)."""
    import random
    rnd = random.Random(5)
    TH, TW = 4, 32
    checked = 0
    for _ in range(600):
        S = rnd.choice([1, 2]); B = rnd.randint(1, 4); Hin = rnd.randint(1, 70); Win = rnd.randint(1, 140)
        Cin, Cout = 64 * rnd.randint(1, 4), 64 * rnd.randint(1, 4)
        H, W = (Hin - 1) // S + 1, (Win - 1) // S + 1
        th, tw = (H + TH - 1) // TH, (W + TW - 1) // TW
        PR, PC = S * TH + 3 - S, S * TW + 3 - S
        NPX = PR * PC; XU = (NPX * 8 + 255) // 256
        oc0, ci0 = 64 * rnd.randrange(Cout // 64), 64 * rnd.randrange(Cin // 64)
        for _ in range(6):
            tile = rnd.randrange(B * th * tw); tid = rnd.randrange(256); spart, spix = tid & 7, tid >> 3
            ow0, oh0, b = (tile % tw) * TW, ((tile // tw) % th) * TH, tile // (tw * th)
            pix = (b * H + oh0) * W + ow0
            for u in range(TH):
                ok = oh0 + u < H and ow0 + spix < W
                if ok:
                    ptr = (pix + spix) * Cout + oc0 + spart * 8 + u * W * Cout
                    assert 2 * ptr == 2 * (pix * Cout + oc0) + ((u * W + spix) * Cout + spart * 8) * 2
                    assert 0 <= ptr and ptr + 8 <= B * H * W * Cout
                    checked += 1
            ih0, iw0 = S * oh0 - 1, S * ow0 - 1
            xbase = ((b * Hin + ih0) * Win + iw0) * Cin + ci0
            for u in range(XU):
                pp = spix + 32 * u
                xr, xc = (pp // PC if pp < NPX else -1), pp % PC
                ok_ptr = xr >= 0 and 0 <= ih0 + xr < Hin and 0 <= iw0 + xc < Win
                ok_buf = ((ih0 + xr) & 0xffffffff) < Hin and ((iw0 + xc) & 0xffffffff) < Win and xr >= 0
                assert ok_ptr == ok_buf
                if ok_ptr:
                    ptr = xbase + spart * 8 + (xr * Win + xc) * Cin
                    off = ((xr * Win + xc) * Cin + spart * 8) * 2
                    assert 2 * ptr == 2 * xbase + off and off < 0x80000000
                    assert 0 <= ptr and ptr + 8 <= B * Hin * Win * Cin
                    checked += 1
    assert checked > 10000


def test_every_python_source_compiles():
    """The DDP wrapper imports most of the package only on a GPU: a syntax error in such a module would first show on the
    GPU box.  Byte-compile everything here."""
    import os
    import py_compile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = 0
    for base in ("torchseg_amd", "tools", "tests", "oracle"):
        for d, _, files in os.walk(os.path.join(root, base)):
            for f in files:
                if f.endswith(".py"):
                    py_compile.compile(os.path.join(d, f), doraise=True)
                    n += 1
    for f in ("bench.py", "__graft_entry__.py"):
        py_compile.compile(os.path.join(root, f), doraise=True)
    assert n > 60


def test_pooled_conv_swap_and_kernel_choice_of_the_general_3x3():
    """CPU: install_pooled_conv re-classes only bias-free 1x1 / stride-1 convolutions with channel counts the kernel takes
    THAT FOLLOW A GLOBAL POOL inside an nn.Sequential (round 5, ADVICE r4: the same convolutions elsewhere — Bottleneck
    conv1 / conv3, shortcuts — keep their class) and passes CPU tensors to the stock forward; the host-side geometry queries
    of the library (no kernel launch) pick the 16-row LDS-DMA kernel exactly for the problems that fill 2 x 256 block slots,
    keep the statistics partial's row count independent of that choice, and refuse what the kernels cannot index."""
    from torchseg_amd.vecconv import PooledConv2d, install_pooled_conv
    from torchseg_amd import _lib as L
    plain = nn.Sequential(nn.Conv2d(128, 64, 1, bias=False), nn.Conv2d(64, 64, 1, bias=False))
    assert install_pooled_conv(plain) == 0 and [type(m) for m in plain] == [nn.Conv2d] * 2        # no pool in front: untouched
    net = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(128, 64, 1, bias=False), nn.Conv2d(64, 19, 1, bias=False),
                        nn.Conv2d(64, 64, 1, bias=True), nn.Conv2d(64, 64, 1, stride=2, bias=False),
                        nn.Conv2d(64, 64, 3, padding=1, bias=False), nn.Conv2d(64, 64, 1, bias=False))
    keys = list(net.state_dict().keys())
    assert install_pooled_conv(net) == 1 and isinstance(net[1], PooledConv2d)
    assert [type(m) for m in list(net)[2:]] == [nn.Conv2d] * 5 and list(net.state_dict().keys()) == keys   # the last 1x1 sits
    #                                                                          behind a 3x3: no pooled vector any more
    net = nn.Sequential(*list(net)[1:])
    x = torch.randn(2, 128, 1, 1)
    assert torch.equal(net[0](x), nn.functional.conv2d(x, net[0].weight))
    lib = L.lib()
    # (B, H, W, Cin, Cout) -> variant at tile width 64: BiSeNet-R18's 3x3 layers at the bench shape (tests/test_conv3g_gpu.py)
    for geom, want in [((16, 128, 128, 128, 128), 1), ((16, 64, 64, 256, 256), 1), ((16, 128, 128, 128, 256), 1),
                       ((16, 128, 128, 256, 64), 1), ((16, 64, 64, 128, 256), 1), ((16, 32, 32, 512, 512), 0),
                       ((16, 64, 64, 256, 128), 0), ((16, 64, 64, 128, 128), 0), ((2, 128, 128, 48, 64), 0)]:
        assert lib.tsg_conv3x3_gen_variant(*geom, 64, 0) == want, geom
        assert lib.tsg_conv3x3_gen_variant(*geom, 64, 1) == 0                  # never with normalise-on-load
        assert lib.tsg_conv3x3_gen_stats_partials(*geom, 64) > 0
    assert lib.tsg_conv3x3_gen_variant(16, 128, 128, 100, 128, 64, 0) < 0      # C_in not a multiple of 16
    assert lib.tsg_conv3x3_s2_dgrad_supported(L.BF16, 64, 128) == 1 and lib.tsg_conv3x3_s2_dgrad_supported(L.BF16, 48, 128) == 0
    assert lib.tsg_conv3x3_s2_dgrad(None, None, None, None, 1, 8, 8, 64, 64, None) == -5      # TSG_E_NULL before any launch
    assert lib.tsg_conv1x1_vec_supported(16, 512, 128) == 1 and lib.tsg_conv1x1_vec_supported(33, 512, 128) == 0
    assert lib.tsg_conv1x1_vec_supported(16, 100, 128) == 0
    assert lib.tsg_conv1x1_vec_fwd(None, None, None, 16, 64, 64, None) == -5


def test_side_stream_gradient_layout_contract():
    """convwrw._kept_as_gradient: AccumulateGrad keeps a first gradient only in the parameter's own layout."""
    import torch
    from torchseg_amd.convwrw import _kept_as_gradient, wrw_on_side_stream
    p = torch.zeros(8, 4, 3, 3).contiguous(memory_format=torch.channels_last)
    assert _kept_as_gradient(torch.empty(8, 4, 3, 3).contiguous(memory_format=torch.channels_last), p)
    assert not _kept_as_gradient(torch.empty(8, 4, 3, 3), p)
    assert not _kept_as_gradient(torch.empty(8, 4, 3, 3, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last), p)
    q = torch.zeros(8, 1, 3, 3)                         # a dimension of length 1: its stride is free
    assert _kept_as_gradient(torch.empty(8, 1, 3, 3).contiguous(memory_format=torch.channels_last), q)
    out = wrw_on_side_stream(lambda: torch.ones(2), torch.nn.Parameter(torch.zeros(2)), torch.zeros(2))   # host tensors: plain call
    assert out.tolist() == [1.0, 1.0]


def test_side_stream_default_yields_to_a_gradient_reducer(monkeypatch):
    """convwrw.side_stream_off_for_collectives: the default (TSG_WRW_STREAM unset) turns the side stream off once a
    reducer exists; an explicit setting is left alone."""
    from torchseg_amd import convwrw
    monkeypatch.setattr(convwrw, "_WRW_ENV", None)
    monkeypatch.setattr(convwrw, "_WRW_STREAM", True)
    convwrw.side_stream_off_for_collectives()
    assert convwrw._WRW_STREAM is False
    monkeypatch.setattr(convwrw, "_WRW_ENV", "1")
    monkeypatch.setattr(convwrw, "_WRW_STREAM", True)
    convwrw.side_stream_off_for_collectives()
    assert convwrw._WRW_STREAM is True


def test_side_stream_tells_the_reducers_hook_from_a_foreign_one():
    """convwrw._foreign_grad_hooks: a post-accumulate-grad hook reads p.grad before the end-of-backward join,
    so it keeps the weight gradient on the compute stream — unless it is the DDP reducer's, which joins first."""
    import torch
    from torchseg_amd import convwrw
    from torchseg_amd.ddp import Reducer
    p = torch.nn.Parameter(torch.zeros(4))
    assert not convwrw._foreign_grad_hooks(p)
    red = Reducer.__new__(Reducer)                       # (its constructor needs a process group: test_dist_cpu.py)
    convwrw.allow_grad_hook(red._on_grad)                # what Reducer.__init__ does before it registers the hook
    p.register_post_accumulate_grad_hook(red._on_grad)
    assert not convwrw._foreign_grad_hooks(p)            # a BOUND method of any reducer instance is recognised
    other = Reducer.__new__(Reducer)
    p.register_post_accumulate_grad_hook(other._on_grad)
    assert not convwrw._foreign_grad_hooks(p)
    q = torch.nn.Parameter(torch.zeros(4))
    h = q.register_post_accumulate_grad_hook(lambda t: None)
    assert convwrw._foreign_grad_hooks(q)
    h.remove()
    assert not convwrw._foreign_grad_hooks(q)
    del red
