"""GPU microbench: achieved GB/s of each SyncBN kernel at BiSeNet-R18 config-2 shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchseg_amd import kernels as K, _lib as L
kp = K.provider()
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
shapes = [(16, 64, 512, 512), (16, 64, 256, 256), (16, 128, 128, 128), (16, 256, 64, 64), (16, 512, 32, 32), (16, 128, 1, 1)]
only = os.environ.get("BENCH_BN_ONLY")          # "bf16nhwc": the benched dtype / layout only
for dtype in ((torch.bfloat16,) if only else (torch.bfloat16, torch.float32)):
    for fmt in ((torch.channels_last,) if only else (torch.contiguous_format, torch.channels_last)):
        for shp in shapes:
            N, C, H, W = shp
            x = torch.randn(shp, device=dev).to(dtype).contiguous(memory_format=fmt)
            dy = torch.randn(shp, device=dev).to(dtype).contiguous(memory_format=fmt)
            lay, N_, C_, HW = K.bn_layout(x)
            g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
            nb = x.numel() * x.element_size()
            t_stats = timeit(lambda: kp.bn_stats(x, lay, N, C, HW))
            partial, S = kp.bn_stats(x, lay, N, C, HW)
            mean, invstd, fp = kp.bn_finalize(partial, S, C, N * HW, None, 1e-5, 0.1, g, b, None, None, None)
            y = torch.empty_like(x)
            t_fwd = timeit(lambda: kp.bn_apply_fwd(x, None, lay, N, C, HW, fp, True, out=y))
            t_red = timeit(lambda: kp.bn_bwd_reduce(dy, x, None, lay, N, C, HW, fp, True))
            p2, S2 = kp.bn_bwd_reduce(dy, x, None, lay, N, C, HW, fp, True)
            _, _, bp = kp.bn_bwd_coeffs(p2, S2, C, N * HW, None, True, invstd, fp, True, True)
            t_bwd = timeit(lambda: kp.bn_bwd_apply(dy, x, None, lay, N, C, HW, bp, True, False))
            t_fin = timeit(lambda: kp.bn_finalize(partial, S, C, N * HW, None, 1e-5, 0.1, g, b, None, None, None))
            print(f"{str(dtype).split('.')[-1]:8s} {'nhwc' if lay else 'nchw'} {shp}  S={S:4d} "
                  f"stats {nb/t_stats/1e9:7.0f} GB/s ({t_stats*1e6:6.1f}us)  fwd {2*nb/t_fwd/1e9:7.0f} GB/s ({t_fwd*1e6:6.1f}us)  "
                  f"bwd_red {2*nb/t_red/1e9:7.0f} GB/s ({t_red*1e6:6.1f}us)  bwd_apply {3*nb/t_bwd/1e9:7.0f} GB/s ({t_bwd*1e6:6.1f}us)  fin {t_fin*1e6:.1f}us",
                  flush=True)
