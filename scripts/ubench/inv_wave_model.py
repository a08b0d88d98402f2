"""Lane-level numpy model of the ONE-WAVE register-resident 32x32 SPD inverse (chol_tiles.h: ct_spd_inverse_wave)
next to a model of the four-wave LDS-panel form it replaces (ct_spd_inverse).

Both eliminate the bordered matrix [[T, I], [I, 0]] with 4x4 pivot blocks; the one-wave form keeps every fragment as the
TRANSPOSED view of the four-wave form's fragment, so that every operand of a pivot block is a register the lane already
holds (raw panel rows), a v_readlane (the 4x4 pivot block) or the result of one MFMA (the rows of P D^-1):

    F[a][b]  (a <= b, blocks of 16 of the 64x64 bordered matrix), lane (lr, lc), register r  =  M[16 b + lc][16 a + lr + 4 r]

Run:  python scripts/ubench/inv_wave_model.py     (checks both against numpy's inverse and against each other, bit for bit)
"""
import numpy as np

LR = np.arange(64) >> 4
LC = np.arange(64) & 15


def mfma(a, b, c):
    """v_mfma_f64_16x16x4_f64: lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15]; register r of lane l is
    C[(l >> 4) + 4 r][l & 15].  k = 0..3 accumulated in order on top of C."""
    A = np.zeros((16, 4)); B = np.zeros((4, 16))
    A[LC, LR] = a
    B[LR, LC] = b
    C = np.zeros((16, 16))
    for r in range(4):
        C[LR + 4 * r, LC] = c[r]
    for k in range(4):
        C = C + np.outer(A[:, k], B[k, :])
    out = np.zeros((4, 64))
    for r in range(4):
        out[r] = C[LR + 4 * r, LC]
    return out


def ldl_column(D, e):
    """the 4x4 pivot block D = L diag(d) L^T (lower triangle read), then column `e` (vector of 4 per lane) of D^-1;
    the operation order of chol_tiles.h"""
    c00 = D[0][0]; q1x, q1y = D[1][0], D[1][1]; q2ax, q2ay, q2bx = D[2][0], D[2][1], D[2][2]
    q3ax, q3ay, q3bx, q3by = D[3][0], D[3][1], D[3][2], D[3][3]
    e0, e1, e2, e3 = e
    r0 = 1.0 / c00
    l10 = q1x * r0; l20 = q2ax * r0; l30 = q3ax * r0
    d1 = -l10 * q1x + q1y
    c21 = -l20 * q1x + q2ay; c31 = -l30 * q1x + q3ay
    r1 = 1.0 / d1
    l21 = c21 * r1; l31 = c31 * r1
    d2 = -l21 * c21 + (-l20 * q2ax + q2bx)
    c32 = -l31 * c21 + (-l30 * q2ax + q3bx)
    r2 = 1.0 / d2
    l32 = c32 * r2
    d3 = -l32 * c32 + (-l31 * c31 + (-l30 * q3ax + q3by))
    r3 = 1.0 / d3
    y1 = -l10 * e0 + e1
    y2 = -l21 * y1 + (-l20 * e0 + e2)
    y3 = -l32 * y2 + (-l31 * y1 + (-l30 * e0 + e3))
    x3 = y3 * r3
    x2 = -l32 * x3 + y2 * r2
    x1 = -l31 * x3 + (-l21 * x2 + y1 * r1)
    x0 = -l30 * x3 + (-l20 * x2 + (-l10 * x1 + e0 * r0))
    return x0, x1, x2, x3


def frag_of(Mat, bi, bj):
    """C-layout fragment of block (bi, bj): register r of lane (lr, lc) = Mat[16 bi + lr + 4 r][16 bj + lc]"""
    return np.stack([Mat[16 * bi + LR + 4 * r, 16 * bj + LC] for r in range(4)])


def inverse_four_waves(T):
    """model of ct_spd_inverse: wave w = 2 bi + bj, LDS panel, returns the lower fragments of T^-1 {(bi, bj): frag}"""
    top = {(bi, bj): frag_of(T, bi, bj) for bi in range(2) for bj in range(2)}
    top[(0, 1)] = np.zeros((4, 64))
    g = {(bi, bj): np.stack([((bi == bj) & (LR + 4 * r == LC)).astype(float) for r in range(4)]) for bi in range(2) for bj in range(2)}
    ti = {(bi, bj): np.zeros((4, 64)) for bi in range(2) for bj in range(2)}
    e = [(LR == k).astype(float) for k in range(4)]
    for kb in range(8):
        cb = 4 * kb; pbj = cb >> 4; cin = cb & 15
        pan = np.zeros((64, 4))
        for (bi, bj) in top:
            if bj != pbj:
                continue
            sel = (LC >= cin) & (LC < cin + 4)
            for r in range(4):
                if bi >= bj:
                    pan[(16 * bi + LR + 4 * r)[sel], (LC - cin)[sel]] = top[(bi, bj)][r][sel]
                if bi <= bj:
                    pan[(32 + 16 * bi + LR + 4 * r)[sel], (LC - cin)[sel]] = g[(bi, bj)][r][sel]
        D = pan[cb:cb + 4, :]
        x = ldl_column(D, e)
        for (bi, bj) in list(top):
            n_top = bi >= bj and cb + 4 < 16 * (bj + 1)
            n_g = bi <= bj and cb + 4 < 16 * (bj + 1) and 16 * bi <= cb + 3
            n_ti = bi >= bj and 16 * bi <= cb + 3
            prt = pan[16 * bi + LC]; prb = pan[32 + 16 * bi + LC]
            at = prt[:, 3] * x[3] + (prt[:, 2] * x[2] + (prt[:, 1] * x[1] + prt[:, 0] * x[0]))
            ab = prb[:, 3] * x[3] + (prb[:, 2] * x[2] + (prb[:, 1] * x[1] + prb[:, 0] * x[0]))
            bt = pan[16 * bj + LC, LR]; bb = pan[32 + 16 * bj + LC, LR]
            if n_top:
                top[(bi, bj)] = mfma(-at, bt, top[(bi, bj)])
            if n_g:
                g[(bi, bj)] = mfma(-ab, bt, g[(bi, bj)])
            if n_ti:
                ti[(bi, bj)] = mfma(ab, bb, ti[(bi, bj)])
    return {k: v for k, v in ti.items() if k[0] >= k[1]}


def readlane(reg, lane):
    return reg[lane]


def inverse_one_wave(T):
    """model of ct_spd_inverse_wave: ONE wave, transposed-view fragments F[(a, b)], a <= b over the blocks 0, 1 (T) and 2, 3
    (border); no LDS, no barrier.  Returns {(bi, bj): lower fragment of T^-1} in the four-wave form's layout."""
    def tfrag(Mat, a, b):   # register r of lane (lr, lc) = Mat[16 b + lc][16 a + lr + 4 r]
        return np.stack([Mat[16 * b + LC, 16 * a + LR + 4 * r] for r in range(4)])
    ident = np.stack([(LC == LR + 4 * r).astype(float) for r in range(4)])
    zero = np.zeros((4, 64))
    F = {(0, 0): tfrag(T, 0, 0), (0, 1): tfrag(T, 0, 1), (1, 1): tfrag(T, 1, 1),
         (0, 2): ident.copy(), (1, 2): zero.copy(), (1, 3): ident.copy(), (0, 3): zero.copy(),
         (2, 2): zero.copy(), (2, 3): zero.copy(), (3, 3): zero.copy()}
    # lanes lc < 4 solve for column lc of the pivot block's inverse and supply its element lr as the A operand of the Y MFMA
    ecol = [((LC == k) | ((LC >= 4) & (k == 0))).astype(float) for k in range(4)]
    n_mfma = 0
    for kb in range(8):
        cb = 4 * kb; p = cb >> 4; rk = kb & 3; cin = cb & 15
        piv = F[(p, p)][rk]
        D = [[readlane(piv, 16 * j + cin + i) if j <= i else None for j in range(4)] for i in range(4)]
        x = ldl_column(D, ecol)
        xs = np.where(LR == 0, x[0], np.where(LR == 1, x[1], np.where(LR == 2, x[2], x[3])))
        dsel = np.where(LC < 4, xs, 0.0)
        # live fragments, from the conditions of the four-wave form (lower (bi, bj) <-> F[(bj, bi)])
        live = []
        for (a, b) in [(0, 0), (0, 1), (1, 1)]:
            if a >= p and cb + 4 < 16 * (a + 1):
                live.append((a, b, -1.0))
        for (a, b) in [(0, 2), (1, 2), (1, 3)]:
            bi, bj = b - 2, a
            if a >= p and cb + 4 < 16 * (bj + 1) and 16 * bi <= cb + 3:
                live.append((a, b, -1.0))
        for (a, b) in [(2, 2), (2, 3), (3, 3)]:
            bi = b - 2
            if 16 * bi <= cb + 3:
                live.append((a, b, 1.0))
        Y = {}
        for b in sorted({b for (_, b, _) in live}):
            Y[b] = mfma(dsel, F[(p, b)][rk], zero)[0]      # register 0: Y(16 b + lc, lr)
            n_mfma += 1
        rawP = {a: F[(p, a)][rk].copy() for a in sorted({a for (a, _, _) in live})}
        for (a, b, sgn) in live:
            F[(a, b)] = mfma(rawP[a], sgn * Y[b], F[(a, b)])
            n_mfma += 1
    # back to the four-wave form's lower fragments: ti(bi, bj)[i][j] = F[(2 + bj, 2 + bi)][j][i]
    def untr(Fr):   # Fr: register r of lane (lr, lc) = X[lc][lr + 4 r]  ->  C-layout of X
        X = np.zeros((16, 16))
        for r in range(4):
            X[LC, LR + 4 * r] = Fr[r]
        return np.stack([X[LR + 4 * r, LC] for r in range(4)])
    return {(0, 0): untr(F[(2, 2)]), (1, 0): untr(F[(2, 3)]), (1, 1): untr(F[(3, 3)])}, n_mfma


def assemble(frags):
    X = np.zeros((32, 32))
    for (bi, bj), f in frags.items():
        for r in range(4):
            X[16 * bi + LR + 4 * r, 16 * bj + LC] = f[r]
    return np.tril(X) + np.tril(X, -1).T


if __name__ == "__main__":
    rng = np.random.default_rng(3)
    worst = 0.0
    for trial in range(20):
        Q = rng.standard_normal((32, 40))
        T = Q @ Q.T + 0.5 * np.eye(32)
        # the four-wave form only ever reads the lower triangle of T's diagonal blocks + block (1, 0); scramble the rest
        Tl = np.tril(T) + np.triu(rng.standard_normal((32, 32)), 1) * 0
        Tl = np.tril(T) + np.tril(T, -1).T
        f4 = inverse_four_waves(Tl)
        f1, n_mfma = inverse_one_wave(Tl)
        X4 = assemble(f4); X1 = assemble(f1)
        ref = np.linalg.inv(T)
        e4 = np.abs(X4 - ref).max() / np.abs(ref).max(); e1 = np.abs(X1 - ref).max() / np.abs(ref).max()
        worst = max(worst, e4, e1)
        same = all(np.array_equal(f4[k], f1[k]) for k in f4)
        assert e4 < 1e-10 and e1 < 1e-10, (e4, e1)
        assert same, "one-wave form is not bit-identical to the four-wave form"
    print(f"20 tiles: both forms within {worst:.2e} of numpy's inverse, one-wave == four-wave bit for bit; {n_mfma} MFMAs per inverse")
