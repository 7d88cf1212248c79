"""GPU probe: MIOpen conv fwd+bwd time for BiSeNet-R18 shapes, NCHW vs NHWC, bf16/fp32."""
import os, sys, time, itertools
import torch
import torch.nn as nn

torch.backends.cudnn.benchmark = os.environ.get("BENCH", "0") == "1"
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "16"))
shapes = [  # (cin, cout, k, stride, H)
    (3, 64, 7, 2, 1024), (64, 64, 3, 2, 512), (64, 64, 3, 2, 256), (64, 128, 1, 1, 128),
    (64, 64, 3, 1, 256), (64, 128, 3, 2, 256), (128, 128, 3, 1, 128), (128, 256, 3, 2, 128),
    (256, 256, 3, 1, 64), (256, 512, 3, 2, 64), (512, 512, 3, 1, 32), (256, 64, 3, 1, 128),
    (128, 256, 3, 1, 64), (128, 256, 3, 1, 128),
]
def run(cin, cout, k, s, H, fmt, dtype):
    conv = nn.Conv2d(cin, cout, k, s, k // 2, bias=False).to(dev).to(memory_format=fmt)
    x = torch.randn(B, cin, H, H, device=dev).contiguous(memory_format=fmt).requires_grad_(True)
    def step():
        with torch.autocast("cuda", dtype=dtype, enabled=dtype != torch.float32):
            y = conv(x)
        y.backward(torch.ones_like(y))
    for _ in range(3): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 5
    for _ in range(n): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
tot = {}
for (cin, cout, k, s, H) in shapes:
    row = []
    for fmt, dtype in itertools.product([torch.contiguous_format, torch.channels_last], [torch.bfloat16, torch.float32]):
        try:
            ms = run(cin, cout, k, s, H, fmt, dtype)
        except Exception as e:
            ms = float("nan"); print("ERR", e)
        key = ("nhwc" if fmt == torch.channels_last else "nchw", str(dtype).split(".")[-1])
        tot[key] = tot.get(key, 0) + ms
        row.append(f"{key[0]}/{key[1]}={ms:.2f}ms")
    fl = 2 * B * cin * cout * k * k * (H // s) ** 2 * 3 / 1e12
    print(f"conv {cin}->{cout} k{k} s{s} H{H}  TFLOP(train)={fl:.3f}  " + "  ".join(row), flush=True)
print("TOTAL", {k: round(v, 2) for k, v in tot.items()})
