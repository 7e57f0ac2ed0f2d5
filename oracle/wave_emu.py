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


# ======================================================================================================================
# backward (mirrors k_linattn_bwd in csrc/k_linattn_bwd.hip)
# ======================================================================================================================
def to_mat(acc):
    """accumulator (16, 64) -> logical 32x32 [row][col]"""
    M = np.zeros((32, 32), F)
    for r in range(16):
        M[rowmap(r, HALF), COL] = acc[r]
    return M


def from_mat(M):
    acc = np.zeros((16, 64), F)
    for r in range(16):
        acc[r] = M[rowmap(r, HALF), COL]
    return acc


def T(acc):
    """32x32 transpose of an accumulator (done through a wave-private LDS tile in the kernel)"""
    return from_mat(to_mat(acc).T.copy())


def la_bwd_unit(x, dyp, Wqkv, Wo, g_pre, hd, dW):
    """One wave, one head, one unit (1 row if n >= 32 else 32/n rows).  x, dyp: (RW, C, n); dyp = d loss / d y_pre.
    Accumulates this head's weight gradients into dW (dict of logical matrices) and returns this head's contribution
    to d loss / d xh (RW, C, n) where xh = rmsnorm(x)*g_pre."""
    RW, C, n = x.shape
    NB = max(1, n // 32)
    NJ = 4 if C <= 8 else 8
    sqC = F(np.sqrt(F(C)))
    scale = F(32 ** -0.5)

    def pos_of(blk):
        if n >= 32:
            return np.zeros(64, int), blk * 32 + COL
        return COL // n, COL % n

    def xload(t, blk):
        rl, pp = pos_of(blk)
        out = np.zeros((NJ, 64), F)
        for j in range(NJ):
            c = chan_of(j, HALF)
            out[j] = np.where(c < C, t[rl, np.minimum(c, C - 1), pp], 0)
        return out

    def as_acc(xr):
        a = acc_zero()
        a[:NJ] = xr
        return a

    X = [xload(x, b) for b in range(NB)]
    DYP = [xload(dyp, b) for b in range(NB)]
    Xh = []
    for b in range(NB):
        ssq = (X[b] ** 2).sum(0)
        ssq = ssq + shfl_xor32(ssq)
        inv = sqC / np.maximum(np.sqrt(ssq), F(1e-12))
        g = np.stack([np.where(chan_of(j, HALF) < C, g_pre[np.minimum(chan_of(j, HALF), C - 1)], 0) for j in range(NJ)])
        Xh.append((X[b] * inv * g).astype(F))

    def wfrag(o_base, j):
        c = chan_of(j, HALF)
        return np.where(c < C, Wqkv[o_base + COL, np.minimum(c, C - 1)], 0).astype(F)

    def wofrag(j):
        c = chan_of(j, HALF)
        return np.where(c < C, Wo[np.minimum(c, C - 1), hd * 32 + COL], 0).astype(F)

    if n >= 32:
        SEG, PARTNER = 16, True
    elif n >= 8:
        SEG, PARTNER = n // 2, True
    else:
        SEG, PARTNER = max(n, 1), False

    # ---------------- stage 0: recompute forward pieces
    kT, vT, Kd, q, qT, do, doT, v = [], [], [], [], [], [], [], []
    for b in range(NB):
        ak, av, aq, avn, ado, adoT = (acc_zero() for _ in range(6))
        for j in range(NJ):
            ak = mfma(Xh[b][j], wfrag(128 + hd * 32, j), ak)
            av = mfma(Xh[b][j], wfrag(256 + hd * 32, j), av)
            aq = mfma(wfrag(hd * 32, j), Xh[b][j], aq)
            avn = mfma(wfrag(256 + hd * 32, j), Xh[b][j], avn)
            ado = mfma(wofrag(j), DYP[b][j], ado)
            adoT = mfma(DYP[b][j], wofrag(j), adoT)
        kT.append(ak), vT.append(av), v.append(avn), do.append(ado), doT.append(adoT)
        m = aq.max(0)
        m = np.maximum(m, shfl_xor32(m))
        aq = np.exp(aq - m)
        s = aq.sum(0)
        s = s + shfl_xor32(s)
        aq = aq * (scale / s)
        q.append(aq), qT.append(T(aq))
    for s0 in range(0, 16, SEG):
        regs = range(s0, s0 + SEG)
        m = np.full(64, -np.inf, F)
        for b in range(NB):
            for r in regs:
                m = np.maximum(m, kT[b][r])
        if PARTNER:
            m = np.maximum(m, shfl_xor32(m))
        ssum = np.zeros(64, F)
        for b in range(NB):
            for r in regs:
                kT[b][r] = np.exp(kT[b][r] - m)
                ssum = ssum + kT[b][r]
        if PARTNER:
            ssum = ssum + shfl_xor32(ssum)
        for b in range(NB):
            for r in regs:
                kT[b][r] = kT[b][r] / ssum
    Kd = [T(kT[b]) for b in range(NB)]

    # ---------------- per row: ctx / dctx and their consumers
    outT = [acc_zero() for _ in range(NB)]
    dq = [acc_zero() for _ in range(NB)]
    dkT = [acc_zero() for _ in range(NB)]
    dv = [acc_zero() for _ in range(NB)]
    delta = np.zeros((RW, 64), F)
    for rho in range(RW):
        ctx, dctx = acc_zero(), acc_zero()
        for b in range(NB):
            for r in range(16):
                if n >= 32:
                    mine0 = mine1 = True
                else:
                    mine0, mine1 = rowmap(r, 0) // n == rho, rowmap(r, 1) // n == rho
                if not (mine0 or mine1):
                    continue
                msk = np.where(HALF == 0, mine0, mine1)
                ctx = mfma(np.where(msk, kT[b][r], 0).astype(F), vT[b][r], ctx)
                dctx = mfma(np.where(msk, qT[b][r], 0).astype(F), doT[b][r], dctx)
        ctxT, dctxT = T(ctx), T(dctx)
        dl = (dctxT * ctxT).sum(0)
        delta[rho] = dl + shfl_xor32(dl)
        for b in range(NB):
            sel = np.ones(64, bool) if n >= 32 else (COL // n) == rho
            for r in range(16):
                outT[b] = mfma(np.where(sel, q[b][r], 0).astype(F), ctx[r], outT[b])
                dq[b] = mfma(ctxT[r], np.where(sel, do[b][r], 0).astype(F), dq[b])
                dkT[b] = mfma(np.where(sel, v[b][r], 0).astype(F), dctxT[r], dkT[b])
                dv[b] = mfma(dctx[r], np.where(sel, Kd[b][r], 0).astype(F), dv[b])

    # ---------------- softmax backward, weight gradients, d xh
    dxh = np.zeros_like(x)
    for b in range(NB):
        t = (q[b] * dq[b]).sum(0)
        t = (t + shfl_xor32(t)) / scale
        dq_raw = q[b] * (dq[b] - t)
        dk_rawT = acc_zero()
        for r in range(16):
            rho_r = np.zeros(64, int) if n >= 32 else rowmap(r, HALF) // n
            dk_rawT[r] = kT[b][r] * (dkT[b][r] - delta[rho_r, LANES])
        dvT = T(dv[b])
        dq_rawT, dk_raw = T(dq_raw), T(dk_rawT)
        XhT, DYPT = T(as_acc(Xh[b])), T(as_acc(DYP[b]))
        aq, ak, av, ao = acc_zero(), acc_zero(), acc_zero(), acc_zero()
        for r in range(16):
            aq = mfma(XhT[r], dq_rawT[r], aq)   # rows c, col d
            ak = mfma(XhT[r], dk_rawT[r], ak)
            av = mfma(XhT[r], dvT[r], av)
            ao = mfma(DYPT[r], outT[b][r], ao)  # rows c, col e
        for nm, a in (("q", aq), ("k", ak), ("v", av), ("o", ao)):
            dW[nm] += to_mat(a)[:C, :]
        # d xh on the VALU: lane (n, half) holds rows rowmap(r, half) of dq_raw / dk_raw / dv
        rl, pp = pos_of(b)
        part = np.zeros((C, 64), F)
        for c in range(C):
            for r in range(16):
                o = rowmap(r, HALF)
                part[c] += Wqkv[hd * 32 + o, c] * dq_raw[r] + Wqkv[128 + hd * 32 + o, c] * dk_raw[r] + Wqkv[256 + hd * 32 + o, c] * dv[b][r]
        full = part + np.stack([shfl_xor32(part[c]) for c in range(C)])
        for l in range(32):  # half 0 lanes write (both halves hold the same sums)
            dxh[rl[l], :, pp[l]] = full[:, l]
    return dxh


def la_bwd(x, dy, ypre, Wqkv, Wo, bo, g_pre, g_out):
    """Full backward on (R, C, n) given the saved pre-norm output ypre.  Returns dict of gradients."""
    R, C, n = x.shape
    sqC = np.sqrt(F(C))
    # (1) norm2 backward (k_block_bwd): y = ypre/max(||ypre||,eps)*g_out*sqrt(C)
    nrm = np.sqrt((ypre ** 2).sum(1, keepdims=True))
    inv = 1.0 / np.maximum(nrm, 1e-12)
    uh = ypre * inv
    gd = dy * g_out[None, :, None] * sqC
    dyp = (inv * (gd - uh * (gd * uh).sum(1, keepdims=True))).astype(F)
    out = {"g_out": (dy * uh * sqC).sum((0, 2)), "b_out": dyp.sum((0, 2))}
    # (2) main kernel: head outer, units inner
    rw = 1 if n >= 32 else 32 // n
    dxh = np.zeros_like(x)
    dWq, dWk, dWv, dWo = (np.zeros((128, C), F) for _ in range(3)), None, None, None
    dWqkv = np.zeros((384, C), F)
    dWo = np.zeros((C, 128), F)
    for hd in range(4):
        dW = {k: np.zeros((C, 32), F) for k in "qkvo"}
        for r0 in range(0, R, rw):
            xs, ds = x[r0:r0 + rw], dyp[r0:r0 + rw]
            pad = rw - xs.shape[0]
            if pad:
                xs = np.concatenate([xs, np.zeros((pad, C, n), F)])
                ds = np.concatenate([ds, np.zeros((pad, C, n), F)])
            dxh[r0:r0 + rw] += la_bwd_unit(xs, ds, Wqkv, Wo, g_pre, hd, dW)[: rw - pad]
        dWqkv[hd * 32:(hd + 1) * 32] += dW["q"].T
        dWqkv[128 + hd * 32:128 + (hd + 1) * 32] += dW["k"].T
        dWqkv[256 + hd * 32:256 + (hd + 1) * 32] += dW["v"].T
        dWo[:, hd * 32:(hd + 1) * 32] += dW["o"]
    # (3) norm1 backward + residual
    nrm = np.sqrt((x ** 2).sum(1, keepdims=True))
    inv = 1.0 / np.maximum(nrm, 1e-12)
    uh = x * inv
    gd = dxh * g_pre[None, :, None] * sqC
    dx = inv * (gd - uh * (gd * uh).sum(1, keepdims=True)) + dy
    out.update({"x": dx.astype(F), "g_pre": (dxh * uh * sqC).sum((0, 2)), "w_qkv": dWqkv, "w_out": dWo})
    return out


# ======================================================================================================================
# "quadratic" form for short rows (n <= 32): S[n][n'] = sum_d q[d][n] k[d][n'] masked to pairs of the same m/z row,
# out = v S^T.  One 32-position block holds 32/n rows; no per-row loop, no wasted MFMAs.  Mirrors the N < 32 paths of
# k_linattn_fwd / k_linattn_bwd.
# ======================================================================================================================
def _common(x, Wqkv, g_pre, hd, C, n):
    NJ = 4 if C <= 8 else 8
    sqC = F(np.sqrt(F(C)))
    rl, pp = COL // n, COL % n

    def xload(t):
        out = np.zeros((NJ, 64), F)
        for j in range(NJ):
            c = chan_of(j, HALF)
            out[j] = np.where(c < C, t[rl, np.minimum(c, C - 1), pp], 0)
        return out

    X = xload(x)
    ssq = (X ** 2).sum(0)
    ssq = ssq + shfl_xor32(ssq)
    inv = sqC / np.maximum(np.sqrt(ssq), F(1e-12))
    g = np.stack([np.where(chan_of(j, HALF) < C, g_pre[np.minimum(chan_of(j, HALF), C - 1)], 0) for j in range(NJ)])
    Xh = (X * inv * g).astype(F)

    def wfrag(o_base, j):
        c = chan_of(j, HALF)
        return np.where(c < C, Wqkv[o_base + COL, np.minimum(c, C - 1)], 0).astype(F)

    if n >= 8:
        SEG, PARTNER = n // 2, True
    else:
        SEG, PARTNER = max(n, 1), False
    kT, vT, q, v = acc_zero(), acc_zero(), acc_zero(), acc_zero()
    for j in range(NJ):
        kT = mfma(Xh[j], wfrag(128 + hd * 32, j), kT)
        vT = mfma(Xh[j], wfrag(256 + hd * 32, j), vT)
        q = mfma(wfrag(hd * 32, j), Xh[j], q)
        v = mfma(wfrag(256 + hd * 32, j), Xh[j], v)
    for s0 in range(0, 16, SEG):
        regs = range(s0, s0 + SEG)
        m = np.full(64, -np.inf, F)
        for r in regs:
            m = np.maximum(m, kT[r])
        if PARTNER:
            m = np.maximum(m, shfl_xor32(m))
        ssum = np.zeros(64, F)
        for r in regs:
            kT[r] = np.exp(kT[r] - m)
            ssum = ssum + kT[r]
        if PARTNER:
            ssum = ssum + shfl_xor32(ssum)
        for r in regs:
            kT[r] = kT[r] / ssum
    m = q.max(0)
    m = np.maximum(m, shfl_xor32(m))
    q = np.exp(q - m)
    s = q.sum(0)
    s = s + shfl_xor32(s)
    q = q * (F(32 ** -0.5) / s)
    return X, Xh, xload, kT, vT, q, v, NJ, SEG, PARTNER


def _mask_rows_vs_col(acc, n):
    """keep element (row i in regs, col j on lane) iff i // n == j // n"""
    out = acc.copy()
    for r in range(16):
        out[r] = np.where(rowmap(r, HALF) // n == COL // n, acc[r], 0)
    return out


def la_fwd_quad_head(x, Wqkv, g_pre, hd):
    """out[e][n] (accumulator: rows e, col n) of one head for a 32-position block of 32/n rows"""
    RW, C, n = x.shape
    X, Xh, xload, kT, vT, q, v, NJ, SEG, PARTNER = _common(x, Wqkv, g_pre, hd, C, n)
    Kd = T(kT)
    ST = acc_zero()
    for r in range(16):
        ST = mfma(Kd[r], q[r], ST)  # rows n', col n
    STm = _mask_rows_vs_col(ST, n)
    out = acc_zero()
    for r in range(16):
        out = mfma(vT[r], STm[r], out)  # rows e, col n
    return out


def la_bwd_unit_quad(x, dyp, Wqkv, Wo, g_pre, hd, dW):
    RW, C, n = x.shape
    X, Xh, xload, kT, vT, q, v, NJ, SEG, PARTNER = _common(x, Wqkv, g_pre, hd, C, n)
    scale = F(32 ** -0.5)
    DYP = xload(dyp)

    def wofrag(j):
        c = chan_of(j, HALF)
        return np.where(c < C, Wo[np.minimum(c, C - 1), hd * 32 + COL], 0).astype(F)

    def as_acc(xr):
        a = acc_zero()
        a[:NJ] = xr
        return a

    do, doT = acc_zero(), acc_zero()
    for j in range(NJ):
        do = mfma(wofrag(j), DYP[j], do)
        doT = mfma(DYP[j], wofrag(j), doT)
    Kd, qT = T(kT), T(q)
    ST, S, dST, dS = acc_zero(), acc_zero(), acc_zero(), acc_zero()
    for r in range(16):
        ST = mfma(Kd[r], q[r], ST)   # rows n', col n
        S = mfma(q[r], Kd[r], S)     # rows n, col n'
        dST = mfma(v[r], do[r], dST)  # rows n', col n
        dS = mfma(do[r], v[r], dS)    # rows n, col n'
    STm, Sm, dSTm, dSm = (_mask_rows_vs_col(a, n) for a in (ST, S, dST, dS))
    outT, dvT, dv, dq, dkT = (acc_zero() for _ in range(5))
    for r in range(16):
        outT = mfma(STm[r], vT[r], outT)  # rows n, col e
        dvT = mfma(Sm[r], doT[r], dvT)    # rows n', col e
        dv = mfma(doT[r], Sm[r], dv)      # rows e, col n'
        dq = mfma(kT[r], dSTm[r], dq)     # rows d, col n
        dkT = mfma(dSm[r], qT[r], dkT)    # rows n', col d
    t = (q * dq).sum(0)
    t = (t + shfl_xor32(t)) / scale
    dq_raw = q * (dq - t)
    dk_rawT = acc_zero()
    for s0 in range(0, 16, SEG):
        regs = range(s0, s0 + SEG)
        dl = np.zeros(64, F)
        for r in regs:
            dl = dl + dkT[r] * kT[r]
        if PARTNER:
            dl = dl + shfl_xor32(dl)
        for r in regs:
            dk_rawT[r] = kT[r] * (dkT[r] - dl)
    dq_rawT, dk_raw = T(dq_raw), T(dk_rawT)
    XhT, DYPT = T(as_acc(Xh)), T(as_acc(DYP))
    aq, ak, av, ao = acc_zero(), acc_zero(), acc_zero(), acc_zero()
    for r in range(16):
        aq = mfma(XhT[r], dq_rawT[r], aq)
        ak = mfma(XhT[r], dk_rawT[r], ak)
        av = mfma(XhT[r], dvT[r], av)
        ao = mfma(DYPT[r], outT[r], ao)
    for nm, a in (("q", aq), ("k", ak), ("v", av), ("o", ao)):
        dW[nm] += to_mat(a)[:C, :]
    rl, pp = COL // n, COL % n
    part = np.zeros((C, 64), F)
    for c in range(C):
        for r in range(16):
            o = rowmap(r, HALF)
            part[c] += Wqkv[hd * 32 + o, c] * dq_raw[r] + Wqkv[128 + hd * 32 + o, c] * dk_raw[r] + Wqkv[256 + hd * 32 + o, c] * dv[r]
    full = part + np.stack([shfl_xor32(part[c]) for c in range(C)])
    dxh = np.zeros_like(x)
    for l in range(32):
        dxh[rl[l], :, pp[l]] = full[:, l]
    return dxh


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
