"""Build and run tools/probes/tr_probe.hip on the GPU box; prints, for a few per-lane address patterns, which LDS
element index (in 16-bit units) each lane receives in each of its four result slots."""
import ctypes, os, subprocess, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join("/tmp", "libtrprobe.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(here, "tr_probe.hip"), "-o", so])
lib = ctypes.CDLL(so)
dev = torch.device("cuda:0")
patterns = {
    "lane*8 (each lane its own 4 contiguous elements)": np.arange(64, dtype=np.uint32) * 8,
    "uniform 0": np.zeros(64, dtype=np.uint32),
    "row-major [16 rows][64 B]: lane l -> (l&15)*64 + (l>>4)*8": ((np.arange(64) & 15) * 64 + (np.arange(64) >> 4) * 8).astype(np.uint32),
    "guide image: (l&15)*2 + (l>>4)*128": ((np.arange(64) & 15) * 2 + (np.arange(64) >> 4) * 128).astype(np.uint32),
}
for name, a in patterns.items():
    ad = torch.from_numpy(a.astype(np.int32)).to(dev)
    out = torch.zeros(64 * 4, dtype=torch.int16, device=dev)
    rc = lib.tr_probe_launch(ctypes.c_void_p(ad.data_ptr()), ctypes.c_void_p(out.data_ptr()), None)
    torch.cuda.synchronize()
    r = out.cpu().numpy().astype(np.uint16).reshape(64, 4)
    print("== %s (rc=%d)" % (name, rc))
    for l in range(64):
        print("  lane %2d addr %4d B (elem %4d): %s" % (l, a[l], a[l] // 2, r[l].tolist()))
