"""CPU emulation of the index arithmetic of csrc/conv3wrw.hip (LDS images, staging descriptors, fragment
addressing, output layout) in numpy, checked against the weight gradient of F.conv2d; plus the LDS bank check of
the B-fragment reads (32-lane groups, bank = dword mod 32).  Development aid for the kernel."""
import numpy as np, torch, torch.nn.functional as F

C, TH, TW, PR, PP, RD, CS, DS = 64, 4, 32, 6, 17, 20, 121, 68
COPY = C * CS
NCH = 8 * PR * PP


def wrw(x, dy):                      # x, dy: [B, H, W, 64] (channels_last storage)
    B, H, W, _ = x.shape
    acc = np.zeros((C, 9 * C))
    for b in range(B):
        for th_ in range((H + TH - 1) // TH):
            for tw_ in range((W + TW - 1) // TW):
                oh0, ow0 = th_ * TH, tw_ * TW
                dyT = np.zeros(C * DS * 2)                       # in bf16 elements
                xT = np.full(2 * COPY * 2, np.nan)
                for tid in range(256):
                    spp, strow, spart = tid & 15, (tid >> 4) & 1, tid >> 5
                    for rs in range(2):
                        t = strow + 2 * rs
                        for px in range(2):
                            oh, ow = oh0 + t, ow0 + 2 * spp + px
                            v = dy[b, oh, ow, spart * 8: spart * 8 + 8] if (oh < H and ow < W) else np.zeros(8)
                            for e in range(8):
                                dyT[2 * ((spart * 8 + e) * DS + 16 * t + spp) + px] = v[e]
                    for u in range(4):
                        q = tid + 256 * u
                        if q >= NCH:
                            continue
                        cpart, rem = divmod(q, PR * PP); r, pc = divmod(rem, PP)
                        ih, iw = oh0 - 1 + r, ow0 - 1 + 2 * pc
                        a = x[b, ih, iw, cpart * 8: cpart * 8 + 8] if (0 <= ih < H and 0 <= iw < W) else np.zeros(8)
                        bb = x[b, ih, iw + 1, cpart * 8: cpart * 8 + 8] if (0 <= ih < H and 0 <= iw + 1 < W) else np.zeros(8)
                        for e in range(8):
                            d = (cpart * 8 + e) * CS + r * RD + pc
                            xT[2 * d], xT[2 * d + 1] = a[e], bb[e]                   # copy 0
                            xT[2 * (COPY + d) + 1], xT[2 * (COPY + d) + 2] = a[e], bb[e]   # copy 1
                for wm in range(2):
                    for wh in range(2):
                        for n in range(32):
                            for half in range(2):
                                for ks in range(8):
                                    fa_base = 2 * ((32 * wm + np.arange(32)) * DS + 4 * half + ks * 8)
                                    for kh in range(3):
                                        for kw in range(3):
                                            sg = kw & 1
                                            qd = (32 * wh + n) * CS + 4 * half + sg * COPY + ((ks >> 1) + kh) * RD + (ks & 1) * 8 + ((kw + sg) >> 1)
                                            fb = xT[2 * qd: 2 * qd + 8]
                                            assert not np.isnan(fb).any(), (n, half, ks, kh, kw)
                                            col = (kh * 3 + kw) * C + 32 * wh + n
                                            for m in range(32):
                                                fa = dyT[fa_base[m]: fa_base[m] + 8]
                                                acc[32 * wm + m, col] += fa @ fb
    return acc.reshape(C, 3, 3, C)            # [oc][kh][kw][ci]


def banks():
    worst = 0
    for kh in range(3):
        for kw in range(3):
            for ks in range(8):
                for half in range(2):
                    for wh in range(2):
                        for i in range(4):
                            seen = {}
                            for n in range(32):
                                sg = kw & 1
                                a = (32 * wh + n) * CS + 4 * half + sg * COPY + ((ks >> 1) + kh) * RD + (ks & 1) * 8 + ((kw + sg) >> 1) + i
                                seen.setdefault(a % 32, set()).add(a)
                            worst = max(worst, max(len(v) for v in seen.values()))
    return worst


if __name__ == "__main__":
    torch.manual_seed(0)
    for (B, H, W) in [(1, 8, 32), (2, 6, 40)]:
        x = torch.randn(B, C, H, W, dtype=torch.float64)
        w = torch.randn(C, C, 3, 3, dtype=torch.float64, requires_grad=True)
        y = F.conv2d(x, w, None, 1, 1)
        dy = torch.randn_like(y)
        y.backward(dy)
        got = wrw(x.permute(0, 2, 3, 1).contiguous().numpy(), dy.permute(0, 2, 3, 1).contiguous().numpy())
        want = w.grad.permute(0, 2, 3, 1).numpy()          # [oc][kh][kw][ci]
        print((B, H, W), "wrw max|d|", np.abs(got - want).max())
    print("worst distinct-address bank multiplicity of a B-fragment read:", banks())
