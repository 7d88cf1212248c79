"""CPU emulator of csrc/psa.hip::psa_mm's INDEXING: the chunk -> LDS staging maps, the NT (ds_read_b128) and TR
(ds_read_b64_tr_b16, semantics measured in tools/probes/) fragment addressing, the 32x32x16 MFMA lane layouts, the
LDS epilogue image and the XCD tile remap, for the three instantiations (forward, dX, dA).  Arithmetic is fp64 on
exact inputs, so any mismatch against the dense reference is an addressing bug.   python tools/emulate_psa_mm.py"""
import numpy as np

BN, BK, T = 64, 64, 256
NT_ROW, B_TR_ROW = 72, BN + 32
B_ELEMS = BK * B_TR_ROW
EPI_ROW = 68
BM = 256                       # rebound by main() for every tile configuration
A_TR_ROW = A_ELEMS = WROWS = MI = None


def configure(bm):
    global BM, A_TR_ROW, A_ELEMS, WROWS, MI
    BM = bm
    A_TR_ROW = BM + 32
    A_ELEMS = max(BM * NT_ROW, BK * A_TR_ROW)
    WROWS, MI = BM // 4, BM // 128


def tr_read(lds, addr):
    """addr[64] element offsets (lane) -> out[64][4]; 16-lane groups: lane l gets S[4j + (l >> 2)][l & 3], j = 0..3."""
    out = np.zeros((64, 4))
    for g in range(4):
        S = np.stack([lds[addr[16 * g + i]: addr[16 * g + i] + 4] for i in range(16)])      # [16][4]
        for l in range(16):
            for j in range(4):
                out[16 * g + l, j] = S[4 * j + (l >> 2), l & 3]
    return out


def run(Aop, Bop, A_TR, B_TR, M, N, K, btrans=None, epi=None):
    """Aop: NT -> [M][K], TR -> [K][M]; Bop: NT -> [N][K], TR -> [K][N].  Returns C [M][N]."""
    tiles_m, tiles_n = -(-M // BM), -(-N // BN)
    tiles = tiles_m * tiles_n
    per = -(-tiles // 8)
    C = np.full((M, N), np.nan)
    seen = set()
    tid = np.arange(T)
    lane, wave = tid & 63, tid >> 6
    half, sub, i16 = lane >> 5, (lane >> 4) & 1, lane & 15
    for bid in range(8 * per):
        t = (bid & 7) * per + (bid >> 3)
        if t >= tiles:
            continue
        assert t not in seen
        seen.add(t)
        tm, tn = t % tiles_m, (t // tiles_m) % tiles_n
        m0, n0 = tm * BM, tn * BN
        acc = np.zeros((T, MI, 2, 16))
        for k0 in range(0, K, BK):
            sa, sb = np.zeros(A_ELEMS + 64), np.zeros(B_ELEMS + 64)
            for q in range(BM // 32):
                for th in range(T):
                    c = th + T * q
                    if A_TR:
                        k, m = k0 + c // (BM // 8), m0 + (c % (BM // 8)) * 8
                        v = Aop[k, m:m + 8] if (k < K and m < M) else np.zeros(8)
                        off = (c // (BM // 8)) * A_TR_ROW + (c % (BM // 8)) * 8
                    else:
                        m, k = m0 + (c >> 3), k0 + (c & 7) * 8
                        v = Aop[m, k:k + 8] if (k < K and m < M) else np.zeros(8)
                        off = (c >> 3) * NT_ROW + (c & 7) * 8
                    sa[off:off + 8] = v
            for q in range(2):
                for th in range(T):
                    c = th + T * q
                    if B_TR:
                        k, n = k0 + (c >> 3), n0 + (c & 7) * 8
                        ok = k < K and n < N
                        v = btrans(Bop[k, n:n + 8], np.arange(n, n + 8), None) if ok else np.zeros(8)
                        off = (c >> 3) * B_TR_ROW + (c & 7) * 8
                    else:
                        n, k = n0 + (c >> 3), k0 + (c & 7) * 8
                        ok = k < K and n < N
                        v = btrans(Bop[n, k:k + 8], None, np.arange(k, k + 8)) if ok else np.zeros(8)
                        off = (c >> 3) * NT_ROW + (c & 7) * 8
                    sb[off:off + 8] = v
            for w in range(4):
                L = np.arange(64)
                hf, sb_, i6 = L >> 5, (L >> 4) & 1, L & 15
                for ks in range(BK // 16):
                    fa, fb = [], []
                    for i in range(MI):
                        if A_TR:
                            base = (8 * hf + (i6 >> 2)) * A_TR_ROW + w * WROWS + 16 * sb_ + 4 * (i6 & 3) + ks * 16 * A_TR_ROW + i * 32
                            fa.append(np.concatenate([tr_read(sa, base), tr_read(sa, base + 4 * A_TR_ROW)], 1))
                        else:
                            base = (w * WROWS + (L & 31)) * NT_ROW + hf * 8 + i * 32 * NT_ROW + ks * 16
                            fa.append(np.stack([sa[b:b + 8] for b in base]))
                    for i in range(2):
                        if B_TR:
                            base = (8 * hf + (i6 >> 2)) * B_TR_ROW + 16 * sb_ + 4 * (i6 & 3) + ks * 16 * B_TR_ROW + i * 32
                            fb.append(np.concatenate([tr_read(sb, base), tr_read(sb, base + 4 * B_TR_ROW)], 1))
                        else:
                            base = (L & 31) * NT_ROW + hf * 8 + i * 32 * NT_ROW + ks * 16
                            fb.append(np.stack([sb[b:b + 8] for b in base]))
                    for i in range(MI):
                        for j in range(2):
                            Am = np.zeros((32, 16)); Bm = np.zeros((16, 32))
                            for l in range(64):
                                Am[l & 31, 8 * (l >> 5): 8 * (l >> 5) + 8] = fa[i][l]
                                Bm[8 * (l >> 5): 8 * (l >> 5) + 8, l & 31] = fb[j][l]
                            D = Am @ Bm
                            for l in range(64):
                                for r in range(16):
                                    acc[w * 64 + l, i, j, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]
        ep = np.zeros(BM * EPI_ROW)
        for th in range(T):
            for i in range(MI):
                for j in range(2):
                    for r in range(16):
                        row = wave[th] * WROWS + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half[th]
                        ep[row * EPI_ROW + j * 32 + (lane[th] & 31)] = acc[th, i, j, r]
        for th in range(T):
            cch = th & 7
            n = n0 + cch * 8
            for q in range(BM // 32):
                row = (th >> 3) + 32 * q
                m = m0 + row
                if m >= M or n >= N:
                    continue
                v = ep[row * EPI_ROW + cch * 8: row * EPI_ROW + cch * 8 + 8]
                C[m, n:n + 8] = epi(v, m, np.arange(n, n + 8)) if epi else v
    assert len(seen) == tiles
    return C


def main():
    for bm in (256, 128):
        configure(bm)
        check()
    print("psa_mm indexing (BM 256 and 128): forward, dX, dA and the multi-tile remap agree with the dense reference")


def check():
    rng = np.random.default_rng(0)
    ident = lambda v, n, k: v
    # forward: out[c][j] = sum_i X[c][i] exp(A[i][j] - lse[j]);  M = Cx, N = N, K = K
    Cx, K, N = 264, 88, 72
    X, A = rng.standard_normal((Cx, K)), rng.standard_normal((K, N))
    lse = np.log(np.exp(A).sum(0))
    P = np.exp(A - lse)
    out = run(X, A, False, True, Cx, N, K, btrans=lambda v, n, k: np.exp(v - lse[n]))
    assert np.allclose(out, X @ P), "forward"
    dO = rng.standard_normal((Cx, N))
    # dX[c][i] = sum_j dO[c][j] P[i][j]:  A = dO NT [M = Cx][K = N], B = A NT [n = i][k = j], lse by k
    dX = run(dO, A, False, False, Cx, K, N, btrans=lambda v, n, k: np.exp(v - lse[k]))
    assert np.allclose(dX, dO @ P.T), "dX"
    # dA[i][j] = P[i][j] (sum_c X[c][i] dO[c][j] - delta[j]):  A = X TR [k = c][m = i], B = dO TR [k = c][n = j]
    delta = ((X @ P) * dO).sum(0)
    dA = run(X, dO, True, True, K, N, Cx, btrans=ident, epi=lambda v, m, n: P[m, n] * (v - delta[n]))
    assert np.allclose(dA, P * (X.T @ dO - delta)), "dA"
    # a shape with several M tiles and N tiles and the XCD remap exercised
    M2, N2, K2 = 520, 136, 64
    A2, B2 = rng.standard_normal((K2, M2)), rng.standard_normal((K2, N2))
    C2 = run(A2, B2, True, True, M2, N2, K2, btrans=ident)
    assert np.allclose(C2, A2.T @ B2), "TR x TR, multi-tile"


if __name__ == "__main__":
    main()
