"""Lane-level numpy emulation of the gfx950 wave64 primitives the LinearAttention kernels are built from.
TEST INFRASTRUCTURE ONLY (see oracle/dq_oracle.py header): it exists so that the register/lane index algebra of
``csrc/k_linattn.hip`` can be checked on the CPU, against the oracle, before a GPU is spent on it.  The functions
below mirror the kernel's structure statement by statement.

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


def la_fwd_block_rows(x, Wqkv, Wo, bo, g_pre, g_out):
    """One wave: x is (rows_in_wave, C, n) with rows_in_wave*min(n,32)... handles n <= 64.
    Returns y of the same shape.  Mirrors k_linattn_fwd<C, N>."""
    RW, C, n = x.shape
    NB = max(1, n // 32)  # 32-position blocks of a row
    assert n in (1, 2, 4, 8, 16, 32, 64)
    assert RW == (1 if n >= 32 else 32 // n)
    NJ = 4 * ((C + 7) // 8) if C > 4 else 4
    NJ = 4 if C <= 8 else 8
    sqC = F(np.sqrt(F(C)))
    scale = F(32 ** -0.5)

    def pos_of(blk):
        """(row_local, pos) handled by each lane's column in block blk"""
        if n >= 32:
            return np.zeros(64, int), blk * 32 + COL
        return COL // n, COL % n

    # ---- load x and pre-norm: X[blk][j] lane (col, half) holds channel chan_of(j, half)
    X = np.zeros((NB, NJ, 64), F)
    Xh = np.zeros((NB, NJ, 64), F)
    for blk in range(NB):
        rl, pp = pos_of(blk)
        for j in range(NJ):
            c = chan_of(j, HALF)
            ok = c < C
            X[blk, j] = np.where(ok, x[rl, np.minimum(c, C - 1), pp], 0)
        ssq = (X[blk] ** 2).sum(0)
        ssq = ssq + shfl_xor32(ssq)
        inv = sqC / np.maximum(np.sqrt(ssq), F(1e-12))
        for j in range(NJ):
            c = chan_of(j, HALF)
            Xh[blk, j] = X[blk, j] * inv * np.where(c < C, g_pre[np.minimum(c, C - 1)], 0)

    def wfrag(o_base, j):
        """weight operand: lane (i = col, half) supplies Wqkv[o_base + col][chan_of(j, half)] (0 beyond C)"""
        c = chan_of(j, HALF)
        return np.where(c < C, Wqkv[o_base + COL, np.minimum(c, C - 1)], 0).astype(F)

    # segments of the k-softmax (positions of one row inside a lane's 16 registers)
    if n >= 32:
        SEG, PARTNER = 16, True
    elif n >= 8:
        SEG, PARTNER = n // 2, True
    else:
        SEG, PARTNER = max(n, 1), False
    ypart = np.zeros((NB, C, 64), F)

    for hd in range(4):
        # ---------------- phase 1: K^T, V^T per block, softmax over n, ctx per row
        kT = np.zeros((NB, 16, 64), F)
        vT = np.zeros((NB, 16, 64), F)
        for blk in range(NB):
            ak, av = acc_zero(), acc_zero()
            for j in range(NJ):
                ak = mfma(Xh[blk, j], wfrag(128 + hd * 32, j), ak)  # kT[n][d]
                av = mfma(Xh[blk, j], wfrag(256 + hd * 32, j), av)  # vT[n][e]
            kT[blk], vT[blk] = ak, av
        if n == 1:
            pass  # softmax over a single position is 1
        for s0 in range(0, 16, SEG):
            regs = range(s0, s0 + SEG)
            m = np.full(64, -np.inf, F)
            for blk in range(NB):
                for r in regs:
                    m = np.maximum(m, kT[blk, r])
            if PARTNER:
                m = np.maximum(m, shfl_xor32(m))
            ssum = np.zeros(64, F)
            for blk in range(NB):
                for r in regs:
                    kT[blk, r] = np.exp(kT[blk, r] - m)
                    ssum = ssum + kT[blk, r]
            if PARTNER:
                ssum = ssum + shfl_xor32(ssum)
            for blk in range(NB):
                for r in regs:
                    kT[blk, r] = kT[blk, r] / ssum

        # ---------------- phase 2: per block q, per row ctx -> out, y accumulation
        nrows = RW
        for blk in range(NB):
            q = acc_zero()
            for j in range(NJ):
                q = mfma(wfrag(hd * 32, j), Xh[blk, j], q)  # q[d][n]
            m = q.max(0)
            m = np.maximum(m, shfl_xor32(m))
            q = np.exp(q - m)
            ssum = q.sum(0)
            ssum = ssum + shfl_xor32(ssum)
            q = q * (scale / ssum)
            out = acc_zero()
            for rho in range(nrows):
                # ctx of row rho
                ctx = acc_zero()
                if n >= 32:
                    for b2 in range(NB):
                        for r in range(16):
                            ctx = mfma(kT[b2, r], vT[b2, r], ctx)
                elif n >= 8:
                    for r in range(rho * SEG, (rho + 1) * SEG):
                        ctx = mfma(kT[0, r], vT[0, r], ctx)
                else:
                    # n in {4, 2, 1}: a register's two lane-halves belong to different rows -> mask one operand
                    for r in range(16):
                        row_of = rowmap(r, HALF) // n  # per lane-half
                        a = np.where(row_of == rho, kT[0, r], 0).astype(F)
                        if not (rowmap(r, 0) // n == rho or rowmap(r, 1) // n == rho):
                            continue
                        ctx = mfma(a, vT[0, r], ctx)
                o = acc_zero()
                for r in range(16):
                    o = mfma(ctx[r], q[r], o)  # out[e][n]
                if nrows == 1:
                    out = o
                else:
                    sel = (COL // n) == rho
                    out = np.where(sel[None, :], o, out)
            # to_out on the VALU: lane holds out[e = rowmap(r, half)][n]
            for c in range(C):
                acc = ypart[blk, c]
                for r in range(16):
                    e = rowmap(r, HALF)
                    acc = acc + Wo[c, hd * 32 + e] * out[r]
                ypart[blk, c] = acc

    y = np.zeros_like(x)
    for blk in range(NB):
        rl, pp = pos_of(blk)
        yv = np.zeros((C, 64), F)
        for c in range(C):
            yv[c] = ypart[blk, c] + shfl_xor32(ypart[blk, c]) + bo[c]
        ssq = (yv ** 2).sum(0)
        inv = sqC / np.maximum(np.sqrt(ssq), F(1e-12))
        for j in range(NJ):
            c = chan_of(j, HALF)
            ok = c < C
            val = yv[np.minimum(c, C - 1), LANES] * inv * g_out[np.minimum(c, C - 1)] + X[blk, j]
            for l in range(64):
                if ok[l]:
                    y[rl[l], c[l], pp[l]] = val[l]
    return y


def la_fwd(x, Wqkv, Wo, bo, g_pre, g_out):
    """Whole tensor (R, C, n): split rows over emulated waves."""
    R, C, n = x.shape
    rw = 1 if n >= 32 else 32 // n
    y = np.zeros_like(x)
    for r0 in range(0, R, rw):
        xs = x[r0:r0 + rw]
        pad = rw - xs.shape[0]
        if pad:
            xs = np.concatenate([xs, np.zeros((pad, C, n), F)])
        y[r0:r0 + rw] = la_fwd_block_rows(xs, Wqkv, Wo, bo, g_pre, g_out)[: rw - pad]
    return y
