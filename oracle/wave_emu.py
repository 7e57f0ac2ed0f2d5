"""Lane-level numpy emulation of the gfx950 wave64 primitives the LinearAttention kernels are built from.
TEST INFRASTRUCTURE ONLY (see oracle/dq_oracle.py header): it exists so that the register/lane index algebra of
``csrc/k_linattn.hip`` can be checked on the CPU, against the oracle, before a GPU is spent on it.  ``la_fwd_reassoc_row``
mirrors the shipped forward kernel statement by statement (tests/test_wave_emu.py); the emulations of the round-1 kernels that
preceded the re-association were removed with those kernels.

v_mfma_f32_32x32x2_f32 (guide section 3): D = A*B + C with A 32x2, B 2x32.
  A operand: lane l supplies A[i = l & 31][k = l >> 5]        (one f32 per lane)
  B operand: lane l supplies B[k = l >> 5][j = l & 31]
  C/D      : register r of lane l holds D[row = (r & 3) + 8*(r >> 2) + 4*(l >> 5)][col = l & 31], r in [0,16)
"""
import numpy as np

LANES = np.arange(64)
COL = LANES & 31
HALF = LANES >> 5
F = np.float32


def rowmap(r, half):
    """row index of accumulator register r in lane-half ``half``"""
    return (r & 3) + 8 * (r >> 2) + 4 * half


def mfma(a, b, c):
    """a, b: (64,) f32 ; c: (16, 64) f32 accumulator -> new accumulator"""
    A = np.stack([a[:32], a[32:]], axis=1)  # [i][k]
    Bm = np.stack([b[:32], b[32:]], axis=0)  # [k][j]
    D = (A.astype(np.float64) @ Bm.astype(np.float64)).astype(F)  # [i][j]
    out = c.copy()
    for r in range(16):
        rows = rowmap(r, HALF)
        out[r] += D[rows, COL]
    return out


def shfl_xor32(v):
    return np.concatenate([v[32:], v[:32]])


def shfl(v, src_lane):
    return v[src_lane]


def acc_zero():
    return np.zeros((16, 64), F)


def chan_of(j, half):
    """channel held by x-register j in lane-half ``half`` (same map as accumulator rows)"""
    return rowmap(j, half)


# ======================================================================================================================
# Re-associated LinearAttention (the algorithm csrc/k_linattn.hip / k_la_bwd.hip / k_la_long.hip ship): the C-row products on
# v_mfma_f32_4x4x1_16b_f32.  Lane mapping measured on gfx950 by tools/probe/mfma4x4.hip:
#   block = lane >> 2 ; a lane supplies A_blk[i = lane & 3] and B_blk[j = lane & 3] ;
#   register i of lane (blk, j) += A_blk[i] * B_blk[j]
# ======================================================================================================================
def mfma4(a, b, c):
    """a, b: (64,) f32 ; c: (4, 64) f32 accumulator -> new accumulator"""
    out = c.copy()
    blk = LANES >> 2
    for i in range(4):
        out[i] += (a[blk * 4 + i].astype(np.float64) * b.astype(np.float64)).astype(F)
    return out


def chain4(stage, pitch, off, g, tile):
    """sum_r mfma4(A = stage[(4*g + (lane&3)) * pitch + off + rowmap(r, half)], B = tile[r]) -- the kernels' chain4 lambda"""
    acc = np.zeros((4, 64), F)
    for r in range(16):
        a = stage[(4 * g + (LANES & 3)) * pitch + off + rowmap(r, HALF)]
        acc = mfma4(a.astype(F), tile[r], acc)
    return acc


def la_fwd_reassoc_row(x, Wqkv, Wo, bo, g_pre, g_out):
    """One wave = one row of n in {32, 64} positions, C in {4, 8}: statement-by-statement mirror of k_linattn_fwd's n >= 32 path
    (M^T = xh k^T and P = M^T q on the 4x4x1 form, W2 = Wo Wv per head, normalisations applied to the C-row results)."""
    C, n = x.shape
    NB, NJ, CG, NP = n // 32, 4, C // 4, n
    assert n in (32, 64) and C in (4, 8)
    sqC = F(np.sqrt(F(C)))
    scale = F(0.17677669529663687)
    X = np.zeros((NB, NJ, 64), F)
    Xh = np.zeros((NB, NJ, 64), F)
    xs = np.zeros(C * NP, F)  # LDS image [c][n]
    for b in range(NB):
        pos = b * 32 + COL
        for j in range(NJ):
            c = rowmap(j, HALF)
            X[b, j] = np.where(c < C, x[np.minimum(c, C - 1), pos], 0).astype(F)
        ssq = (X[b] * X[b]).sum(0)
        ssq = ssq + shfl_xor32(ssq)
        inv = sqC / np.maximum(np.sqrt(ssq), F(1e-12))
        for j in range(NJ):
            c = rowmap(j, HALF)
            Xh[b, j] = X[b, j] * inv * np.where(c < C, g_pre[np.minimum(c, C - 1)], 0)
            ok = c < C
            xs[(c * NP + b * 32 + COL)[ok]] = Xh[b, j][ok]
    W2 = np.zeros((4, C, C), F)
    for hd in range(4):
        W2[hd] = (Wo[:, hd * 32:(hd + 1) * 32].astype(np.float64) @ Wqkv[256 + hd * 32:256 + (hd + 1) * 32].astype(np.float64)).astype(F)
    yown = np.zeros((NB, NJ, 64), F)
    for hd in range(4):
        def wfrag(o_base, j):
            c = rowmap(j, HALF)
            return np.where(c < C, Wqkv[o_base + hd * 32 + COL, np.minimum(c, C - 1)], 0).astype(F)
        kT = []
        for b in range(NB):
            ak = acc_zero()
            for j in range(NJ):
                ak = mfma(Xh[b, j], wfrag(128, j), ak)  # rows n, col d
            kT.append(ak)
        m = np.max(np.stack([t.max(0) for t in kT]), axis=0)
        m = np.maximum(m, shfl_xor32(m))
        kT = [np.exp(t - m).astype(F) for t in kT]
        ssum = sum(t.sum(0) for t in kT)
        ssum = ssum + shfl_xor32(ssum)
        krs = (F(1) / ssum).astype(F)
        ms = np.zeros(C * 32, F)  # LDS image [c][d]
        for g in range(CG):
            mt = np.zeros((4, 64), F)
            for b in range(NB):
                mt = mt + chain4(xs, NP, b * 32, g, kT[b])
            for i in range(4):
                v = (mt[i] + shfl_xor32(mt[i])) * krs
                ms[((g * 4 + i) * 32 + COL)[HALF == 0]] = v[HALF == 0]
        for b in range(NB):
            q = acc_zero()
            for j in range(NJ):
                q = mfma(wfrag(0, j), Xh[b, j], q)  # rows d, col n
            mq = q.max(0)
            mq = np.maximum(mq, shfl_xor32(mq))
            q = np.exp(q - mq).astype(F)
            qsum = q.sum(0)
            qsum = qsum + shfl_xor32(qsum)
            qs = scale / qsum
            P = np.zeros((C, 64), F)
            for g in range(CG):
                pp = chain4(ms, 32, 0, g, q)
                for i in range(4):
                    P[g * 4 + i] = pp[i] + shfl_xor32(pp[i])
            for j in range(NJ):
                cp = rowmap(j, HALF)
                s = np.zeros(64, F)
                for c in range(C):
                    s = s + np.where(cp < C, W2[hd][np.minimum(cp, C - 1), c], 0) * P[c]
                yown[b, j] += qs * s
    y = np.zeros((C, n), F)
    for b in range(NB):
        yv = np.zeros((NJ, 64), F)
        for j in range(NJ):
            c = rowmap(j, HALF)
            yv[j] = np.where(c < C, yown[b, j] + bo[np.minimum(c, C - 1)], 0)
        ssq = (yv * yv).sum(0)
        ssq = ssq + shfl_xor32(ssq)
        inv = sqC / np.maximum(np.sqrt(ssq), F(1e-12))
        for j in range(NJ):
            c = rowmap(j, HALF)
            ok = c < C
            out = yv[j] * np.where(ok, g_out[np.minimum(c, C - 1)], 0) * inv + X[b, j]
            y[c[ok], (b * 32 + COL)[ok]] = out[ok]
    return y
