"""A/B of the weight-gradient fetch: TSG_CONV_WRW_BUF=0 (pointer loads) writes its results to /tmp, =1 (raw buffer loads)
compares BIT-EXACTLY against them (same arithmetic, only the addressing differs) and both print HIP-event times.
usage: TSG_CONV_WRW_BUF=0 python tools/ab_wrw_buf.py; TSG_CONV_WRW_BUF=1 python tools/ab_wrw_buf.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchseg_amd import kernels as K
kp = K.provider(); dev = torch.device("cuda:0")
buf = os.environ.get("TSG_CONV_WRW_BUF", "0")
# (name, B, Cin, Cout, Hin, Win, stride, with in_ab)
CASES = [("layer3 256->256 @64", 16, 256, 256, 64, 64, 1, False), ("layer2 128->128 @128", 16, 128, 128, 128, 128, 1, False),
         ("layer4 512->512 @32", 16, 512, 512, 32, 32, 1, False), ("layer3.0 s2 128->256 @128", 16, 128, 256, 128, 128, 2, False),
         ("ragged 64->128 19x45", 3, 64, 128, 19, 45, 1, False), ("ragged s2 128->64 21x37", 2, 128, 64, 21, 37, 2, False),
         ("bn-on-load 64->128 @40", 2, 64, 128, 40, 72, 1, True)]
ok = True
for name, B, Cin, Cout, H, W, s, aff in CASES:
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn(B, Cin, H, W, device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, Cout, (H - 1) // s + 1, (W - 1) // s + 1, device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    ab = None
    if aff:
        ab = torch.stack([torch.rand(Cin, device=dev, generator=g) + 0.5, torch.randn(Cin, device=dev, generator=g) * 0.1]).contiguous()
    dw = kp.conv3x3_wrw(x, dy, variant="gen", stride=s, in_ab=ab)
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(10):
        kp.conv3x3_wrw(x, dy, variant="gen", stride=s, in_ab=ab)
    en.record(); torch.cuda.synchronize()
    us = st.elapsed_time(en) / 10 * 1e3
    f = os.path.join(os.environ.get("TSG_AB_DIR", "/tmp"), "ab_wrw_%s_%d_%d_%d_%d.pt" % (name.split()[0].replace("/", "_"), Cin, Cout, H, s))
    if buf == "0":
        torch.save(dw.cpu(), f); msg = "saved"
    else:
        ref = torch.load(f); same = torch.equal(dw.cpu(), ref); ok &= same; msg = "bit-equal" if same else "DIFFERENT max|d| %.3e" % (dw.cpu() - ref).abs().max().item()
    print("BUF=%s %-28s %8.1f us  %s" % (buf, name, us, msg), flush=True)
print("RESULT", "OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
