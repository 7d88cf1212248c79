"""CPU restatements of the index arithmetic of the round-4 convolution kernels (csrc/conv3g.hip), so that the `-m "not gpu"`
tier covers their mathematics:
  * conv3s2d_k: dx of a 3x3 / stride 2 / padding 1 convolution by output parity (1 / 2 / 2 / 4 taps of the SAME dy patch, the
    tap -> (parity class, patch offset) table of the kernel and the mode-1 filter slot 8 - (3 kh + kw)) equals autograd;
  * conv3h_fwd_k: the LDS-DMA piece -> (patch pixel, 16-byte part) mapping with out-of-range offsets as zero padding
    reproduces the zero-padded input patch of every tile, including ragged edges;
  * the statistics on the matrix cores: ones x Y and the diagonal of Y^T x Y are the channel sums / square sums, and the
    (lane, register) of the diagonal element is what the kernel extracts."""
import numpy as np
import torch
import torch.nn.functional as F


def _dx_by_parity(dy, w, H, W):
    """dy [B,Co,OH,OW], w [Co,Ci,3,3] -> dx [B,Ci,H,W], following conv3s2d_k: for every dy pixel (a, b) and tap (kh, kw):
    class pa = kh != 1, pb = kw != 1; patch offset dr = kh == 0, dc = kw == 0; dx[2a+pa][2b+pb] += dy[a+dr][b+dc] . w[kh][kw]"""
    B, Co, OH, OW = dy.shape
    Ci = w.shape[1]
    dyp = F.pad(dy, (0, 1, 0, 1))                           # the (+1, +1) halo: zeros beyond the image (out-of-range offsets)
    dx = torch.zeros(B, Ci, 2 * OH, 2 * OW, dtype=dy.dtype)
    # the prepared mode-1 filter holds original tap t = 3 kh + kw in slot 8 - t; the kernel reads slot 8 - t for tap t
    wf = torch.stack([w[:, :, (8 - s) // 3, (8 - s) % 3] for s in range(9)])      # [slot][Co][Ci]
    for t in range(9):
        kh, kw = t // 3, t % 3
        pa, dr = (0, 0) if kh == 1 else (1, 1 if kh == 0 else 0)
        pb, dc = (0, 0) if kw == 1 else (1, 1 if kw == 0 else 0)
        src = dyp[:, :, dr:dr + OH, dc:dc + OW]             # dy[a + dr][b + dc]
        dx[:, :, pa::2, pb::2] += torch.einsum("bohw,oi->bihw", src, wf[8 - t])
    return dx[:, :, :H, :W]


def test_stride2_data_gradient_by_output_parity_equals_autograd():
    g = torch.Generator().manual_seed(0)
    for (B, Ci, Co, H, W) in [(1, 3, 4, 8, 8), (2, 5, 2, 7, 9), (1, 2, 3, 1, 1), (1, 4, 4, 16, 5)]:
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        w = torch.randn(Co, Ci, 3, 3, generator=g, dtype=torch.float64)
        dy = torch.randn(B, Co, OH, OW, generator=g, dtype=torch.float64)
        x = torch.zeros(B, Ci, H, W, dtype=torch.float64, requires_grad=True)
        F.conv2d(x, w, None, 2, 1).backward(dy)
        got = _dx_by_parity(dy, w, H, W)
        assert torch.allclose(got, x.grad, atol=1e-12), (B, Ci, Co, H, W)


def test_conv3h_dma_pieces_reproduce_the_zero_padded_patch():
    """20 pieces of 64 lanes x 16 B: vector v = piece * 64 + lane is part v & 1 of patch pixel q = v >> 1 = (row q / 34, column
    q % 34) = input pixel (oh0 - 1 + row, ow0 - 1 + col); q >= 612 or a pixel outside the image gets the sentinel offset (the
    buffer load then returns zeros).  The B fragment of (patch row r, shift kw) for lane (p, half) sits at byte
    ((4 wave + r) 34 + kw + p) 32 + half 16 of that image."""
    PW, NPX, PIECES = 34, 18 * 34, 20
    rng = np.random.default_rng(1)
    for (H, W, oh0, ow0) in [(40, 70, 0, 0), (40, 70, 32, 64), (16, 32, 0, 0), (5, 3, 0, 0), (33, 65, 32, 64)]:
        Cin, chunk = 32, 1
        x = rng.standard_normal((H, W, Cin)).astype(np.float32)
        xb = x.reshape(-1)                                  # flat element index = (ih W + iw) Cin + c
        lds = np.full((PIECES * 64, 8), np.nan, np.float32)  # [vector][8 channels]
        for v in range(PIECES * 64):
            q, part = v >> 1, v & 1
            ih, iw = oh0 - 1 + q // PW, ow0 - 1 + q % PW
            ok = q < NPX and 0 <= ih < H and 0 <= iw < W
            if ok:
                off = (ih * W + iw) * Cin + chunk * 16 + part * 8
                lds[v] = xb[off:off + 8]
            else:
                lds[v] = 0.0                                # offset 0x80000000 >= num_records
        xp = np.zeros((H + 2 + 32, W + 2 + 64, Cin), np.float32)
        xp[1:H + 1, 1:W + 1] = x                            # zero padding 1 (and beyond, for ragged tiles)
        img = lds.reshape(-1)                               # bf16 elements: pixel q at q * 16
        for wave in range(4):
            for r in range(6):
                for kw in range(3):
                    for p in (0, 13, 31):
                        for half in (0, 1):
                            e0 = ((4 * wave + r) * PW + kw + p) * 16 + half * 8
                            want = xp[oh0 + 4 * wave + r, ow0 + kw + p, chunk * 16 + half * 8: chunk * 16 + half * 8 + 8]
                            assert np.array_equal(img[e0:e0 + 8], want), (H, W, oh0, ow0, wave, r, kw, p, half)


def test_matrix_core_statistics_layout():
    """D = A x B with A = ones or Y^T and B = Y for a 16-pixel K step: every row of ones x Y holds the channel sums, the
    diagonal of Y^T x Y the square sums; in the 32x32 accumulator layout (row = (r & 3) + 8 (r >> 2) + 4 half, column =
    lane & 31) channel c's diagonal element is held by the lane with half == (c >> 2 & 1) in register (c & 3) + 4 (c >> 3)."""
    rng = np.random.default_rng(2)
    Y = rng.standard_normal((128, 32))                      # the wave's 128 pixels x one 32-channel block
    d1 = np.ones((32, 128)) @ Y
    d2 = Y.T @ Y
    assert np.allclose(d1[0], Y.sum(0)) and np.allclose(np.diag(d2), (Y * Y).sum(0))
    for c in range(32):
        half, r = (c >> 2) & 1, (c & 3) + 4 * (c >> 3)
        row = (r & 3) + 8 * (r >> 2) + 4 * half
        assert row == c, (c, half, r, row)                  # lane (c, half) register r is D[c][c]


def _mfma_32x32x16(a_frag, b_frag, acc):
    """v_mfma_f32_32x32x16 by its register layout: a_frag / b_frag [64 lanes][8]: lane l holds row (A) / column (B) l & 31 at
    k = 8 (l >> 5) + e; acc [64 lanes][16]: lane l holds column l & 31 at rows (r & 3) + 8 (r >> 2) + 4 (l >> 5)."""
    A = np.zeros((32, 16)); Bm = np.zeros((16, 32))
    for l in range(64):
        A[l & 31, 8 * (l >> 5): 8 * (l >> 5) + 8] = a_frag[l]
        Bm[8 * (l >> 5): 8 * (l >> 5) + 8, l & 31] = b_frag[l]
    D = A @ Bm
    for l in range(64):
        for r in range(16):
            acc[l, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]
    return acc


def test_pooled_1x1_kernels_by_their_lane_arithmetic():
    """vec1x1_fwd_k / vec1x1_bwd_k (csrc/vecconv.hip) restated lane by lane: forward M = batch, N = output channel, K = input
    channel; weight gradient M = output channel, N = input channel, K = batch; data gradient M = batch, N = input channel,
    K = output channel — each against the plain matrix products."""
    rng = np.random.default_rng(3)
    B, Cin, Cout = 5, 48, 80
    x, w, dy = rng.standard_normal((B, Cin)), rng.standard_normal((Cout, Cin)), rng.standard_normal((B, Cout))
    # ---- forward: block = 32 output channels
    y = np.zeros((B, Cout))
    for blk in range((Cout + 31) // 32):
        acc = np.zeros((64, 16))
        for k0 in range(0, Cin, 16):
            a = np.zeros((64, 8)); b = np.zeros((64, 8))
            for l in range(64):
                n, half = l & 31, l >> 5
                o = blk * 32 + n
                if n < B: a[l] = x[n, k0 + 8 * half: k0 + 8 * half + 8]
                if o < Cout: b[l] = w[o, k0 + 8 * half: k0 + 8 * half + 8]
            _mfma_32x32x16(a, b, acc)
        for l in range(64):
            o = blk * 32 + (l & 31)
            for r in range(16):
                bb = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)
                if bb < B and o < Cout: y[bb, o] = acc[l, r]
    assert np.allclose(y, x @ w.T)
    # ---- weight gradient: 32 x 32 tiles, K = batch (one step of 16, rows >= B zero)
    dw = np.zeros((Cout, Cin))
    cit = (Cin + 31) // 32
    for blk in range(((Cout + 31) // 32) * cit):
        ot, ct = blk // cit, blk % cit
        a = np.zeros((64, 8)); b = np.zeros((64, 8))
        for l in range(64):
            n, half = l & 31, l >> 5
            o, ci = ot * 32 + n, ct * 32 + n
            for e in range(8):
                bb = 8 * half + e
                if bb < B and o < Cout: a[l, e] = dy[bb, o]
                if bb < B and ci < Cin: b[l, e] = x[bb, ci]
        acc = _mfma_32x32x16(a, b, np.zeros((64, 16)))
        for l in range(64):
            ci = ct * 32 + (l & 31)
            for r in range(16):
                oo = ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)
                if oo < Cout and ci < Cin: dw[oo, ci] = acc[l, r]
    assert np.allclose(dw, dy.T @ x)
    # ---- data gradient: 32 input channels per block, K = output channels
    dx = np.zeros((B, Cin))
    for blk in range(cit):
        acc = np.zeros((64, 16))
        for k0 in range(0, Cout, 16):
            a = np.zeros((64, 8)); b = np.zeros((64, 8))
            for l in range(64):
                n, half = l & 31, l >> 5
                ci = blk * 32 + n
                if n < B: a[l] = dy[n, k0 + 8 * half: k0 + 8 * half + 8]
                if ci < Cin: b[l] = w[k0 + 8 * half: k0 + 8 * half + 8, ci]
            _mfma_32x32x16(a, b, acc)
        for l in range(64):
            ci = blk * 32 + (l & 31)
            for r in range(16):
                bb = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)
                if bb < B and ci < Cin: dx[bb, ci] = acc[l, r]
    assert np.allclose(dx, dy @ w)
