"""CPU emulation of the index arithmetic of csrc/stemconv.hip (LDS layouts, fragment addressing,
K-axis padding) in numpy, checked against F.conv2d and its weight gradient.  Development aid: the
MFMA lane->(row, k) convention itself is the one psa.hip's gemm_nt_bf16 is tested with."""
import numpy as np, torch, torch.nn.functional as F

KP, PR, PD, TH, TW = 176, 13, 36, 4, 32
DS, RS, RIC, PLANE = 136, 24, 15, 1088

def pack_w(w):
    wp = np.zeros((64, KP), np.float64)
    for k in range(KP):
        r, s = k >> 3, k & 7
        if r < 21 and s >= 1:
            wp[:, k] = w.reshape(64, 147)[:, r * 7 + s - 1]
    return wp

def load_patch(x, b, oh0, ow0):
    _, _, H, W = x.shape
    patch = np.zeros(3 * PR * PD * 2)
    for idx in range(3 * PR * PD):
        pr, dc = divmod(idx, PD); ic, rr = divmod(pr, PR)
        ih, iw = 2 * oh0 - 3 + rr, 2 * ow0 - 4 + 2 * dc
        if 0 <= ih < H and 0 <= iw < W:
            patch[2 * idx] = x[b, ic, ih, iw]; patch[2 * idx + 1] = x[b, ic, ih, iw + 1]
    return patch

def fwd(x, w):
    B, _, H, W = x.shape
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    wp = pack_w(w)
    y = np.zeros((B, OH, OW, 64))
    for b in range(B):
        for th in range((OH + TH - 1) // TH):
            for tw in range((OW + TW - 1) // TW):
                oh0, ow0 = th * TH, tw * TW
                patch = load_patch(x, b, oh0, ow0)
                for wave in range(4):
                    Bm = np.zeros((32, KP))
                    for t in range(11):
                        for half in range(2):
                            r = 2 * t + half
                            if r >= 21: r = 0
                            ic, kh = divmod(r, 7)
                            for p in range(32):
                                q = (ic * PR + 2 * wave + kh) * PD + p          # dword index
                                Bm[p, t * 16 + half * 8: t * 16 + half * 8 + 8] = patch[2 * q: 2 * q + 8]
                    out = Bm @ wp.T                                               # [pixel][oc]
                    oh = oh0 + wave
                    if oh < OH:
                        n = min(32, OW - ow0)
                        y[b, oh, ow0:ow0 + n] = out[:n]
    return y

def plane_base(q, sg):
    return (q * 2 + sg) * PLANE + sg + 4 * q

def wrw(x, dy):                                    # dy: [B, OH, OW, 64]
    B, _, H, W = x.shape
    OH, OW = dy.shape[1:3]
    acc = np.zeros((64, 192))
    for b in range(B):
        for th in range((OH + TH - 1) // TH):
            for tw in range((OW + TW - 1) // TW):
                oh0, ow0 = th * TH, tw * TW
                dyT = np.zeros(64 * DS)
                for tid in range(256):
                    spp, strow, spart = tid & 15, (tid >> 4) & 1, tid >> 5
                    for rs in range(2):
                        t = strow + 2 * rs
                        for px in range(2):
                            oh, ow = oh0 + t, ow0 + 2 * spp + px
                            v = dy[b, oh, ow, spart * 8: spart * 8 + 8] if (oh < OH and ow < OW) else np.zeros(8)
                            for e in range(8):
                                dyT[(spart * 8 + e) * DS + 2 * (16 * t + spp) + px] = v[e]
                pl = np.full(4 * PLANE * 2, np.nan)
                patch = load_patch(x, b, oh0, ow0)
                for idx in range(3 * PR * PD):
                    pr, dc = divmod(idx, PD); ic, rr = divmod(pr, PR)
                    row = (ic * RIC + rr) * RS * 2
                    e0, e1 = patch[2 * idx], patch[2 * idx + 1]
                    pl[plane_base(0, 0) * 2 + row + dc] = e0
                    pl[plane_base(0, 1) * 2 + row + dc + 1] = e0
                    pl[plane_base(1, 0) * 2 + row + dc] = e1
                    pl[plane_base(1, 1) * 2 + row + dc + 1] = e1
                for wm in range(2):
                    for wn in range(2):
                        for j in range(3):
                            for n in range(32):
                                k = 32 * (3 * wn + j) + n
                                r, s = k >> 3, k & 7
                                if r >= 21: r = 20
                                ic, kh = divmod(r, 7); q, sh = s & 1, s >> 1; sg = sh & 1
                                bbase = plane_base(q, sg) + (ic * RIC + kh) * RS + ((sh + sg) >> 1)
                                for ks in range(8):
                                    for half in range(2):
                                        off = (2 * (ks >> 1)) * RS + (ks & 1) * 8 + 4 * half
                                        fb = pl[2 * (bbase + off): 2 * (bbase + off) + 8]
                                        if k < 168 and (k & 7):
                                            assert not np.isnan(fb).any(), (k, ks, half)
                                        fb = np.nan_to_num(fb)
                                        for oc in range(32 * wm, 32 * wm + 32):
                                            fa = dyT[oc * DS + ks * 16 + 8 * half: oc * DS + ks * 16 + 8 * half + 8]
                                            acc[oc, k] += fa @ fb
    dw = np.zeros((64, 147))
    for k in range(KP):
        r, s = k >> 3, k & 7
        if r < 21 and s >= 1:
            dw[:, r * 7 + s - 1] = acc[:, k]
    return dw.reshape(64, 3, 7, 7)

def banks():
    """LDS bank check of the wgrad B-fragment reads (MI355X_MICROARCH.md: a ds_read_b32 is serviced in two groups
    of 32 lanes, bank = dword address mod 32): distinct addresses per bank within a group."""
    worst = 0
    for wn in range(2):
        for j in range(3):
            for ks in range(8):
                for i in range(4):
                  for half in range(2):
                    seen = {}
                    for n in range(32):
                        k = 32 * (3 * wn + j) + n
                        r, s = k >> 3, k & 7
                        if r >= 21: r = 20
                        ic, kh = divmod(r, 7); q, sh = s & 1, s >> 1; sg = sh & 1
                        a = plane_base(q, sg) + (ic * RIC + kh) * RS + ((sh + sg) >> 1) + (2 * (ks >> 1)) * RS + (ks & 1) * 8 + 4 * half + i
                        seen.setdefault(a % 32, set()).add(a)
                    worst = max(worst, max(len(v) for v in seen.values()))
    return worst

if __name__ == "__main__":
    torch.manual_seed(0)
    for (B, H, W) in [(1, 64, 64), (2, 22, 70)]:
        x = torch.randn(B, 3, H, W, dtype=torch.float64)
        w = torch.randn(64, 3, 7, 7, dtype=torch.float64, requires_grad=True)
        y0 = F.conv2d(x, w, None, 2, 3)
        dy = torch.randn_like(y0)
        y0.backward(dy)
        y = fwd(x.numpy(), w.detach().numpy())
        print((B, H, W), "fwd max|d|", np.abs(y - y0.detach().permute(0, 2, 3, 1).numpy()).max())
        dw = wrw(x.numpy(), dy.permute(0, 2, 3, 1).contiguous().numpy())
        print((B, H, W), "wrw max|d|", np.abs(dw - w.grad.numpy()).max())
    print("worst distinct-address bank multiplicity of a B-fragment read:", banks())
