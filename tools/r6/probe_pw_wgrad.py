"""1x1 weight gradient as a chunked batched GEMM (deterministic: no split-K atomics) against the vendor's backward-filter:
time, run-to-run bit equality, distance from float64.  Shapes: the five full-map 1x1 convolutions of BiSeNet-R18."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torchseg_amd.tuning import use_shipped_miopen_db
use_shipped_miopen_db(0)
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
cl = dict(memory_format=torch.channels_last)
SH = [("ffm 256->256 @128", 256, 256, 128, 1), ("sp 64->128 @128", 64, 128, 128, 1), ("l2.ds 64->128 s2 @256", 64, 128, 256, 2),
      ("l3.ds 128->256 s2 @128", 128, 256, 128, 2), ("l4.ds 256->512 s2 @64", 256, 512, 64, 2)]
for name, K_, N_, H, st in SH:
    x = torch.randn(16, K_, H, H, device=dev).bfloat16().contiguous(**cl)
    OH = (H - 1) // st + 1
    dy = torch.randn(16, N_, OH, OH, device=dev).bfloat16().contiguous(**cl)
    w = torch.randn(N_, K_, 1, 1, device=dev).bfloat16()
    def vendor():
        return torch.ops.aten.convolution_backward(dy, x, w, None, [st, st], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    def vendor_both():
        return torch.ops.aten.convolution_backward(dy, x, w, None, [st, st], [0, 0], [1, 1], False, [0, 0], 1, [True, True, False])
    def vendor_dx():
        return torch.ops.aten.convolution_backward(dy, x, w, None, [st, st], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0]
    def ours(nb=64, f32=True):
        xs = x if st == 1 else x[:, :, ::st, ::st].contiguous(**cl)
        M = 16 * OH * OH
        a = dy.permute(0, 2, 3, 1).reshape(nb, M // nb, N_).transpose(1, 2)
        b = xs.permute(0, 2, 3, 1).reshape(nb, M // nb, K_)
        if f32:
            p = torch.bmm(a, b, out_dtype=torch.float32)
        else:
            p = torch.bmm(a, b).float()
        return p.sum(0)
    ref = torch.nn.grad.conv2d_weight(x.double().cpu()[:2], w.shape, dy.double().cpu()[:2], stride=st) if False else None
    v1, v2 = vendor(), vendor()
    try:
        o1, o2 = ours(), ours()
        f32 = True
    except Exception as ex:
        print("out_dtype failed:", str(ex)[:100]); f32 = False
        o1, o2 = ours(f32=False), ours(f32=False)
    # truth in fp64 on the GPU (small enough)
    xs = x if st == 1 else x[:, :, ::st, ::st]
    truth = torch.einsum("bnhw,bkhw->nk", dy.double(), xs.double())
    rel = lambda t: ((t.double().reshape(N_, K_) - truth).norm() / truth.norm()).item()
    print("%-24s vendor wgrad %6.1f us (run-to-run equal %s, rel %.1e) | vendor dx %6.1f us, both %6.1f us | bmm nb=64 %6.1f us, nb=32 %6.1f us, nb=128 %6.1f us (equal %s, rel %.1e, fp32 partials %s)"
          % (name, timeit(vendor), torch.equal(v1, v2), rel(v1), timeit(vendor_dx), timeit(vendor_both),
             timeit(lambda: ours(64, f32)), timeit(lambda: ours(32, f32)), timeit(lambda: ours(128, f32)), torch.equal(o1, o2), rel(o1), f32), flush=True)
