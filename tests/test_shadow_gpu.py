"""GPU: torchseg_amd.shadow — the bf16 / rotated shadows of the fp32 filters equal the per-layer casts they replace, one
launch refreshes all of them, and they follow every way a parameter can change (torch in-place ops, FusedSGD's kernel)."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _filters(cuda):
    g = torch.Generator().manual_seed(0)
    shapes = [(64, 64, 3, 3), (128, 64, 3, 3), (256, 128, 3, 3), (64, 128, 3, 3), (512, 512, 3, 3)]
    return [nn.Parameter(torch.randn(*s, generator=g).to(cuda).contiguous(memory_format=torch.channels_last)) for s in shapes]


def test_shadows_equal_the_per_layer_casts_and_follow_updates(cuda):
    from torchseg_amd import kernels as K
    from torchseg_amd.optim import FusedSGD
    from torchseg_amd.shadow import _Bank
    kp = K.provider()
    bank = _Bank()
    ps = _filters(cuda)
    calls = []
    orig = kp.lib.tsg_weight_shadow_refresh

    def check_all():
        for p in ps:
            wb, wrt = bank.get(p, want_rot=True)
            assert wb.stride() == p.stride() and torch.equal(wb, p.detach().to(torch.bfloat16))
            assert torch.equal(wrt, kp.conv3x3_weight_rot180_t(p.detach()))
            # round 5: the fragment-order images of the general 3x3 kernel (forward, data gradient, parity data gradient)
            # come out of the same launch and equal what tsg_conv3x3_gen_prep_filter writes for the same filter
            O, I = p.shape[0], p.shape[1]
            for mode, bn in ((0, 64), (1, 64), (1, 32)):
                like = torch.empty(1, O if mode else I, 8, 8, device=p.device)
                want, _ = kp.conv3x3_gen_prep_filter(p.detach(), mode, like, bn=bn)     # detached: the per-call kernel
                assert torch.equal(bank.get_gen(p, mode, bn), want), (tuple(p.shape), mode, bn)
    check_all()
    # a torch-side change of ONE parameter is noticed through its version counter; one launch rewrites all shadows
    with torch.no_grad():
        ps[2].mul_(1.5)
    check_all()
    # FusedSGD writes parameters from its own kernel: it must leave fresh shadows behind
    import torchseg_amd.shadow as shadow_mod
    old_bank, shadow_mod.bank = shadow_mod.bank, bank
    try:
        opt = FusedSGD(ps, lr=0.1, momentum=0.9, weight_decay=1e-4)
        for p in ps:
            p.grad = torch.randn_like(p)
        before = [p.detach().clone() for p in ps]
        opt.step()
        assert all(not torch.equal(b, p.detach()) for b, p in zip(before, ps))
        for p in ps:                                   # no refresh is triggered by get(): versions did not move
            e = bank.entries[bank._key(p)]
            assert e.version == p._version
            assert torch.equal(e.wb, p.detach().to(torch.bfloat16))
            assert torch.equal(e.wrt, kp.conv3x3_weight_rot180_t(p.detach()))
    finally:
        shadow_mod.bank = old_bank


def test_wrw_conv_uses_the_shadows_and_trains_like_the_cast_path(cuda):
    """Two copies of a small stack of WrwConv2d layers, one with shadows and one casting per call, three FusedSGD steps
    under bf16 autocast: the same losses and weights (the shadows hold exactly the values the casts produce; the library's
    128-channel kernels are not run-to-run deterministic, hence a tolerance of a few fp32 ulps of the update)."""
    from torchseg_amd import convwrw
    from torchseg_amd.convwrw import install_conv_wrw
    from torchseg_amd.optim import FusedSGD
    torch.manual_seed(1)

    def make():
        m = nn.Sequential(nn.Conv2d(64, 64, 3, 1, 1, bias=False), nn.ReLU(), nn.Conv2d(64, 128, 3, 2, 1, bias=False), nn.ReLU(),
                          nn.Conv2d(128, 128, 3, 1, 1, bias=False)).to(cuda).to(memory_format=torch.channels_last)
        assert install_conv_wrw(m) == 3
        return m
    a = make(); b = make(); b.load_state_dict(a.state_dict())
    x = torch.randn(2, 64, 32, 32, device=cuda).contiguous(memory_format=torch.channels_last)
    out = {}
    for name, m, flag in (("shadow", a, True), ("cast", b, False)):
        convwrw._SHADOW = flag
        try:
            opt = FusedSGD(m.parameters(), lr=0.05, momentum=0.9)
            losses = []
            for _ in range(3):
                opt.zero_grad()
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    loss = m(x).float().square().mean()
                loss.backward()
                opt.step()
                losses.append(loss.item())
            out[name] = (losses, [p.detach().clone() for p in m.parameters()])
        finally:
            convwrw._SHADOW = False
    assert out["shadow"][0] == pytest.approx(out["cast"][0], rel=1e-5)
    for p, q in zip(out["shadow"][1], out["cast"][1]):
        torch.testing.assert_close(p, q, rtol=1e-5, atol=1e-6)
