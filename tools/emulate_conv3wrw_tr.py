"""CPU emulation of csrc/conv3wrw.hip variant 2 (conv3_wrw_tr_k): pixel-major LDS tiles and fragments built by
ds_read_b64_tr_b16, modelled with the lane/element mapping measured by tools/probes/tr_probe.hip
(lane i of a 16-lane group contributes S[i][0..3] from its own address; lane l receives S[4j + (l >> 2)][l & 3])."""
import numpy as np, torch, torch.nn.functional as F

C, TH, TW, PR, PC, RBE = 64, 4, 32, 6, 34, 96
NPX = PR * PC
DYN = TH * TW * RBE


def tr_read(lds, addr):
    """addr: 64 element offsets (multiples of 4).  Returns [64][4] per the measured semantics."""
    out = np.zeros((64, 4))
    for g in range(4):
        S = np.stack([lds[addr[16 * g + i]: addr[16 * g + i] + 4] for i in range(16)])
        for l in range(16):
            for j in range(4):
                out[16 * g + l, j] = S[4 * j + (l >> 2), l & 3]
    return out


def wrw(x, dy):
    B, H, W, _ = x.shape
    acc = np.zeros((C, 9 * C))
    lane = np.arange(64)
    half, sub, i16 = lane >> 5, (lane >> 4) & 1, lane & 15
    fbase = (8 * half + (i16 >> 2)) * RBE + 16 * sub + 4 * (i16 & 3)
    for b in range(B):
        for th_ in range((H + TH - 1) // TH):
            for tw_ in range((W + TW - 1) // TW):
                oh0, ow0 = th_ * TH, tw_ * TW
                lds = np.full(DYN + NPX * RBE, np.nan)
                for tid in range(256):
                    spart, spix = tid & 7, tid >> 3
                    for u in range(TH):
                        ok = oh0 + u < H and ow0 + spix < W
                        v = dy[b, oh0 + u, ow0 + spix, spart * 8: spart * 8 + 8] if ok else np.zeros(8)
                        lds[(spix + 32 * u) * RBE + spart * 8: (spix + 32 * u) * RBE + spart * 8 + 8] = v
                    for u in range(7):
                        pp = spix + 32 * u
                        if pp >= NPX:
                            continue
                        r, c = divmod(pp, PC)
                        ih, iw = oh0 - 1 + r, ow0 - 1 + c
                        v = x[b, ih, iw, spart * 8: spart * 8 + 8] if (0 <= ih < H and 0 <= iw < W) else np.zeros(8)
                        lds[DYN + pp * RBE + spart * 8: DYN + pp * RBE + spart * 8 + 8] = v
                for wm in range(2):
                    for wh in range(2):
                        for ks in range(8):
                            fa = np.concatenate([tr_read(lds, fbase + 32 * wm + (ks * 16 + q4) * RBE) for q4 in (0, 4)], axis=1)
                            assert not np.isnan(fa).any()
                            for kh in range(3):
                                for kw in range(3):
                                    px = ((ks >> 1) + kh) * PC + (ks & 1) * 16 + kw
                                    fb = np.concatenate([tr_read(lds, DYN + fbase + 32 * wh + (px + q4) * RBE) for q4 in (0, 4)], axis=1)
                                    assert not np.isnan(fb).any()
                                    # MFMA 32x32x16: lane (m = lane & 31, k-half = lane >> 5) on both operands
                                    for hh in range(2):
                                        A = fa[32 * hh: 32 * hh + 32]            # [m][8]
                                        Bm = fb[32 * hh: 32 * hh + 32]           # [n][8]
                                        acc[32 * wm: 32 * wm + 32, (kh * 3 + kw) * C + 32 * wh: (kh * 3 + kw) * C + 32 * wh + 32] += A @ Bm.T
    return acc.reshape(C, 3, 3, C)


if __name__ == "__main__":
    torch.manual_seed(0)
    for (B, H, W) in [(1, 8, 32), (2, 6, 40)]:
        x = torch.randn(B, C, H, W, dtype=torch.float64)
        w = torch.randn(C, C, 3, 3, dtype=torch.float64, requires_grad=True)
        y = F.conv2d(x, w, None, 1, 1)
        dy = torch.randn_like(y)
        y.backward(dy)
        got = wrw(x.permute(0, 2, 3, 1).contiguous().numpy(), dy.permute(0, 2, 3, 1).contiguous().numpy())
        print((B, H, W), "wrw(tr) max|d|", np.abs(got - w.grad.permute(0, 2, 3, 1).numpy()).max())
    # bank check of a transposing read: 32-lane halves, 64 banks of 4 B, 8 B per lane
    worst = 0
    lane = np.arange(64); half, sub, i16 = lane >> 5, (lane >> 4) & 1, lane & 15
    fb = ((8 * half + (i16 >> 2)) * RBE + 16 * sub + 4 * (i16 & 3)) * 2          # bytes
    for grp in range(2):
        banks = {}
        for l in range(32 * grp, 32 * grp + 32):
            for d in range(2):
                banks.setdefault(((fb[l] // 4) + d) % 64, set()).add(fb[l] // 4 + d)
        worst = max(worst, max(len(v) for v in banks.values()))
    print("worst bank multiplicity of a transposing read (32-lane group):", worst)
