"""GPU parity per building block, through the stand-alone C-ABI entry points (include/dq_hip.h "building blocks"): the fixtures
captured from the REFERENCE's own modules (tests/golden/blocks.npz: RMSNorm with the eps-clamp column, SinusoidalPosEmb + time_mlp,
ConditionalScaleShift, ResnetBlock with / without res_conv, Downsample / Upsample) go straight into the HIP kernels; the
backward of the ResnetBlock and the bottleneck attention (RT = 400 / 2000 and ragged tails) are checked against the oracle's
autograd; RoPE (parity unpinned: third-party rotary_embedding_torch) gets property tests."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def rel(a, b):
    a = a.detach().float().cpu()
    b = (b if torch.is_tensor(b) else torch.as_tensor(np.asarray(b))).detach().float().cpu().reshape(a.shape)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.fixture(scope="module")
def N():
    from dquartic import _native

    _native.lib()
    return _native


def dev(a):
    return (a if torch.is_tensor(a) else T(np.ascontiguousarray(a))).detach().float().contiguous().cuda()


# ------------------------------------------------------------------------------------------------ RMSNorm (unet1d.py:113-140)
@pytest.mark.parametrize("C", [4, 12])
def test_rmsnorm_fixture_incl_eps_clamp(N, golden, C):
    g = golden("blocks.npz")
    x, gain, y = g[f"rmsnorm{C}/x"], g[f"rmsnorm{C}/g"], g[f"rmsnorm{C}/y"]
    assert np.all(x[0, :, 2] == 0.0)  # the all-zero column: ||x|| = 0 -> the norm is clamped to 1e-12, output 0 (not NaN)
    xd, gd = dev(x), dev(gain.reshape(-1))
    out = torch.empty_like(xd)
    N.check(N.lib().dq_rmsnorm_fwd(N.ptr(xd), N.ptr(gd), N.ptr(out), C, x.shape[0], x.shape[2], N.stream_ptr()), "dq_rmsnorm_fwd")
    assert rel(out, y) < 2e-6 and bool(torch.isfinite(out).all()) and float(out[0, :, 2].abs().max()) == 0.0
    # the same norm fused behind a conv (how every other RMSNorm of the network runs): identity 1x1 conv -> RMSNorm
    w = torch.eye(C).reshape(C, C, 1).contiguous().cuda()
    out2 = torch.empty_like(xd)
    N.check(N.lib().dq_conv_fwd(N.ptr(xd), N.ptr(w), None, N.ptr(gd), 0, N.ptr(out2), C, C, 1, 0, x.shape[0], x.shape[2], x.shape[2],
                                N.stream_ptr()), "dq_conv_fwd")
    assert rel(out2, y) < 2e-6 and float(out2[0, :, 2].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------ time embedding (unet1d.py:196-218, 956-960)
def test_time_mlp_fixture(N, golden):
    g = golden("blocks.npz")
    t = T(g["time/t"]).cuda()
    B = t.numel()
    w1, b1, w2, b2 = (dev(g[f"time/{k}"]) for k in ("1.weight", "1.bias", "3.weight", "3.bias"))
    sinu, temb, scratch = torch.empty(B, 4, device="cuda"), torch.empty(B, 16, device="cuda"), torch.empty(B * 100, device="cuda")
    N.check(N.lib().dq_time_mlp_fwd(N.ptr(w1), N.ptr(b1), N.ptr(w2), N.ptr(b2), N.ptr(t), N.ptr(sinu), N.ptr(temb), N.ptr(scratch), B,
                                    N.stream_ptr()), "dq_time_mlp_fwd")
    # sin / cos of arguments up to 999 rad: the device functions differ from glibc's in the last bits of the reduced argument
    assert float((sinu.cpu() - T(g["time/sinu"])).abs().max()) < 2e-6
    assert rel(temb, g["time/out"]) < 5e-6


# ------------------------------------------------------------------------------------------------ ConditionalScaleShift (unet1d.py:648-678)
def test_conditional_scale_shift_fixture(N, golden):
    g = golden("blocks.npz")
    x, te, y = g["css/x"], g["css/temb"], g["css/y"]  # x (5, 1, 8): 5 rows of ONE sample, temb (1, 16)
    ss = torch.empty(1, 2, device="cuda")
    ted, wd, bd = dev(te), dev(g["css/to_scale_shift.1.weight"]), dev(g["css/to_scale_shift.1.bias"])  # (named: they must outlive the call)
    N.check(N.lib().dq_scale_shift_fwd(N.ptr(ted), N.ptr(wd), N.ptr(bd), N.ptr(ss), 1, 2, N.stream_ptr()), "dq_scale_shift_fwd")
    RT, MZ = x.shape[0], x.shape[2]
    cond = dev(x.reshape(1, RT, MZ))
    xin, ms1 = torch.randn(1, RT, MZ, device="cuda"), torch.rand(1, RT, device="cuda")
    cat0, ms1n = torch.empty(RT, 2, MZ, device="cuda"), torch.empty(1, RT, device="cuda")
    N.check(N.lib().dq_prep_inputs_fwd(N.ptr(xin), N.ptr(cond), N.ptr(ms1), N.ptr(ss), 1.0, 0.0, N.ptr(cat0), N.ptr(ms1n), 1, RT, MZ,
                                       N.stream_ptr()), "dq_prep_inputs_fwd")
    assert rel(cat0[:, 0], y[:, 0]) < 2e-6          # channel 0: cond * (scale + 1) + shift
    assert torch.equal(cat0[:, 1], xin[0])          # channel 1: x (unet1d.py:1115: cat((init_cond, x)))
    assert torch.equal(ms1n, ms1)


# ------------------------------------------------------------------------------------------------ Downsample / Upsample (unet1d.py:82-110)
def test_down_and_up_sample_fixtures(N, golden):
    g = golden("blocks.npz")
    for name, mode, K in (("down", 1, 4), ("up", 2, 3)):
        x, y, w, b = g[f"{name}/x"], g[f"{name}/y"], g[f"{name}/weight"], g[f"{name}/bias"]
        out = torch.empty(y.shape, device="cuda")
        xd, wd, bd = dev(x), dev(w), dev(b)
        N.check(N.lib().dq_conv_fwd(N.ptr(xd), N.ptr(wd), N.ptr(bd), None, 0, N.ptr(out), w.shape[0], w.shape[1], K, mode,
                                    x.shape[0], x.shape[2], y.shape[2], N.stream_ptr()), "dq_conv_fwd")
        assert rel(out, y) < 2e-6, name


# ------------------------------------------------------------------------------------------------ ResnetBlock (unet1d.py:271-323)
RES_KEYS = ("mlp.1.weight", "mlp.1.bias", "block1.proj.weight", "block1.proj.bias", "block1.norm.g", "block2.proj.weight",
            "block2.proj.bias", "block2.norm.g", "res_conv.weight", "res_conv.bias")


def _res_flat(wd):
    keys = [k for k in RES_KEYS if k in wd]
    return torch.cat([torch.as_tensor(np.asarray(wd[k])).float().reshape(-1) for k in keys]), keys


def _run_resblock(N, wd, x, temb, rows_per_sample, split, gy=None):
    """forward (+ backward when gy is given) through dq_resblock_*; the input is handed over as cat(xA, xB) when split > 0"""
    L = N.lib()
    rows, cin, n = x.shape
    cout = np.asarray(wd["block1.proj.weight"]).shape[0]
    flat, keys = _res_flat(wd)
    flat = flat.cuda()
    cinA = split if split else cin
    xA = x[:, :cinA].contiguous().cuda()
    xB = x[:, cinA:].contiguous().cuda() if split else None
    nws = L.dq_resblock_workspace_floats(cin, cout, rows, n, rows_per_sample)
    assert nws > 0
    ws = torch.empty(nws, device="cuda")
    out = torch.empty(rows, cout, n, device="cuda")
    td = dev(temb)
    N.check(L.dq_resblock_fwd(N.ptr(flat), N.ptr(xA), cinA, N.ptr(xB), cin - cinA, N.ptr(td), N.ptr(out), cout, rows, n, rows_per_sample,
                              1, N.ptr(ws), nws, N.stream_ptr()), "dq_resblock_fwd")
    if gy is None:
        return out
    dA, dB = torch.zeros_like(xA), (torch.zeros_like(xB) if split else None)
    grads = torch.zeros_like(flat)
    dss = torch.empty(rows // rows_per_sample, 2 * cout, device="cuda")
    gyd = gy.cuda()
    N.check(L.dq_resblock_bwd(N.ptr(flat), N.ptr(xA), cinA, N.ptr(xB), cin - cinA, N.ptr(gyd), N.ptr(dA), N.ptr(dB), N.ptr(grads),
                              N.ptr(dss), cout, rows, n, rows_per_sample, N.ptr(ws), nws, N.stream_ptr()), "dq_resblock_bwd")
    torch.cuda.synchronize()
    dx = torch.cat([dA, dB], dim=1) if split else dA
    gd, o = {}, 0
    for k in keys:
        m = int(np.prod(np.asarray(wd[k]).shape))
        gd[k] = grads[o:o + m].cpu()
        o += m
    return out, dx.cpu(), gd, dss.cpu()


# (a block whose input is wider than its output always receives cat(x, skip) in the network -- unet1d.py:1151, 1154, 1160 -- and is
# handed over as those two tensors here)
@pytest.mark.parametrize("name,split", [("res_4_4_64", 0), ("res_24_12_4", 12), ("res_32_16_1", 16), ("res_8_4_64", 4)])
def test_resnet_block_fixture_forward(N, golden, name, split):
    g = golden("blocks.npz")
    wd = {k[len(name) + 3:]: v for k, v in g.items() if k.startswith(name + "/w/")}
    x = T(g[f"{name}/x"])
    out = _run_resblock(N, wd, x, g[f"{name}/temb"], rows_per_sample=x.shape[0], split=split)  # the fixture's 3 rows: one sample
    assert rel(out, g[f"{name}/y"]) < 5e-6, name


@pytest.mark.parametrize("cin,cout,n,rows,rps,split", [(4, 4, 64, 70, 35, 0), (8, 8, 32, 40, 20, 0), (16, 8, 16, 36, 12, 8), (24, 12, 4, 150, 50, 12),
                                                       (32, 16, 1, 66, 33, 16), (8, 4, 64, 26, 13, 4), (16, 16, 96, 2, 1, 0), (16, 16, 400, 3, 1, 0),
                                                       (12, 8, 8, 90, 45, 8), (8, 8, 8, 64, 32, 0), (8, 4, 32, 48, 24, 4), (16, 8, 256, 6, 3, 8),
                                                       # 12 / 16 channels with the weight gradients in the same launch (k_res_bwd_wg, wide path):
                                                       # identity residual, narrower skip than the block, several tiles per sample, rows of 1..8
                                                       (12, 12, 8, 66, 33, 0), (16, 16, 2, 130, 65, 0), (28, 16, 2, 160, 80, 16), (16, 12, 8, 70, 35, 12),
                                                       (24, 12, 8, 200, 100, 12), (32, 16, 4, 300, 150, 16), (16, 16, 1, 600, 300, 0),
                                                       # the bottleneck's 16-channel blocks, RT position = lane column (k_res_rt.hip): RT axes shorter
                                                       # than a tile, one short of / one over a tile and a workgroup (14 / 56 forward, 12 / 48 backward),
                                                       # longer than the 512 the thread-per-position kernel took
                                                       (16, 16, 1, 5, 1, 0), (16, 16, 2, 3, 1, 0), (16, 16, 11, 3, 1, 0), (16, 16, 12, 2, 1, 0), (16, 16, 13, 4, 1, 0),
                                                       (16, 16, 14, 2, 1, 0), (16, 16, 15, 2, 1, 0), (16, 16, 47, 2, 1, 0), (16, 16, 49, 2, 1, 0), (16, 16, 57, 2, 1, 0),
                                                       (16, 16, 413, 2, 1, 0), (16, 16, 700, 2, 1, 0), (16, 16, 2000, 1, 1, 0)])
def test_resnet_block_backward_vs_oracle_autograd(N, cin, cout, n, rows, rps, split):
    """every dispatch of the ResnetBlock (fused m/z-row kernels, channel-parallel deep levels, the step-by-step bottleneck path
    with rows_per_sample = 1): forward, dX, all weight gradients and d(scale, shift) against autograd over the oracle"""
    from oracle import dq_oracle as O

    gen = torch.Generator().manual_seed(100 * cin + n)
    B = rows // rps
    r = lambda *s: torch.randn(*s, generator=gen)
    wd = {"mlp.1.weight": r(2 * cout, 16) * 0.3, "mlp.1.bias": r(2 * cout) * 0.1, "block1.proj.weight": r(cout, cin, 3) * 0.3,
          "block1.proj.bias": r(cout) * 0.1, "block1.norm.g": torch.rand(1, cout, 1, generator=gen) + 0.5,
          "block2.proj.weight": r(cout, cout, 3) * 0.3, "block2.proj.bias": r(cout) * 0.1, "block2.norm.g": torch.rand(1, cout, 1, generator=gen) + 0.5}
    if cin != cout:
        wd["res_conv.weight"], wd["res_conv.bias"] = r(cout, cin, 1) * 0.3, r(cout) * 0.1
    x, temb, gy = r(rows, cin, n), r(B, 16), r(rows, cout, n)
    p = {"b." + k: v.clone().requires_grad_() for k, v in wd.items()}
    xo = x.clone().requires_grad_()
    yo = O.resnet_block(p, "b", xo, temb, rps)
    (yo * gy).sum().backward()
    out, dx, gd, dss = _run_resblock(N, wd, x, temb, rps, split, gy)
    assert rel(out, yo) < 1e-5
    assert rel(dx, xo.grad) < 2e-5
    for k in wd:
        if k.startswith("mlp."):
            continue
        assert rel(gd[k], p["b." + k].grad) < 5e-5, k
    # d(scale, shift): d mlp.1.bias = sum_b dss_b ; d mlp.1.weight = sum_b dss_b (x) SiLU(temb_b)
    assert rel(dss.sum(0), p["b.mlp.1.bias"].grad) < 5e-5
    assert rel(dss.t() @ torch.nn.functional.silu(temb), p["b.mlp.1.weight"].grad) < 5e-5


# ------------------------------------------------------------------------------------------------ a level's convolutional part in one launch
@pytest.mark.parametrize("pre,C,cp,cs,n,rows,rps,nblocks", [(0, 4, 4, 0, 64, 70, 35, 2), (1, 8, 4, 0, 16, 90, 45, 2), (1, 12, 12, 0, 2, 66, 33, 2),
                                                            (2, 8, 12, 8, 16, 36, 12, 2), (2, 16, 16, 12, 2, 160, 80, 2), (2, 4, 8, 4, 64, 26, 13, 2),
                                                            (3, 4, 4, 4, 64, 26, 13, 1), (0, 16, 16, 16, 1, 130, 65, 2), (1, 16, 12, 0, 1, 128, 64, 2)])
def test_level_forward_vs_oracle(N, pre, C, cp, cs, n, rows, rps, nblocks):
    """k_level_fwd (dq_level_fwd): [Downsample | Upsample | k3 conv] -> ResnetBlock -> ResnetBlock with the skip concatenations of the up
    path, against the oracle's conv + resnet_block composition (reference unet1d.py:82-110, 271-323, 1134-1163)."""
    import torch.nn.functional as F
    from oracle import dq_oracle as O

    gen = torch.Generator().manual_seed(1000 * pre + 10 * C + n)
    r = lambda *s: torch.randn(*s, generator=gen)
    B, K = rows // rps, 4 if pre == 1 else 3
    n_in = {0: n, 1: 2 * n, 2: n // 2, 3: n}[pre]
    x = r(rows, cp if pre else C, n_in)
    temb = r(B, 16)
    pw, pb = r(C, cp, K) * 0.3, r(C) * 0.1
    blocks, flat = [], ([pw.reshape(-1), pb] if pre else [])
    for _ in range(nblocks):
        cin = C + cs
        wd = {"mlp.1.weight": r(2 * C, 16) * 0.3, "mlp.1.bias": r(2 * C) * 0.1, "block1.proj.weight": r(C, cin, 3) * 0.3,
              "block1.proj.bias": r(C) * 0.1, "block1.norm.g": torch.rand(1, C, 1, generator=gen) + 0.5,
              "block2.proj.weight": r(C, C, 3) * 0.3, "block2.proj.bias": r(C) * 0.1, "block2.norm.g": torch.rand(1, C, 1, generator=gen) + 0.5}
        if cs:
            wd["res_conv.weight"], wd["res_conv.bias"] = r(C, cin, 1) * 0.3, r(C) * 0.1
        blocks.append(wd)
        flat += [wd[k].reshape(-1) for k in RES_KEYS if k in wd]
    skips = [r(rows, cs, n) if cs else None for _ in range(2)]
    # oracle
    if pre == 1:
        h = F.conv1d(x, pw, pb, stride=2, padding=1)
    elif pre == 2:
        h = F.conv1d(F.interpolate(x, scale_factor=2, mode="nearest"), pw, pb, padding=1)
    elif pre == 3:
        h = F.conv1d(x, pw, pb, padding=1)
    else:
        h = x
    outs = []
    for i, wd in enumerate(blocks):
        xin = torch.cat([h, skips[i]], dim=1) if cs else h
        h = O.resnet_block({"b." + k: v for k, v in wd.items()}, "b", xin, temb, rps)
        outs.append(h)
    # kernel
    L = N.lib()
    params = torch.cat(flat).cuda()
    assert params.numel() == L.dq_level_param_floats(pre, C, cp, cs, nblocks)
    xd, td = x.cuda(), temb.cuda()
    s0, s1 = (sk.cuda() if sk is not None else None for sk in skips)
    o0, o1 = torch.empty(rows, C, n, device="cuda"), torch.empty(rows, C, n, device="cuda")
    ws = torch.empty(2 * B * 2 * C, device="cuda")
    N.check(L.dq_level_fwd(N.ptr(params), pre, N.ptr(xd), cp, N.ptr(s0), N.ptr(s1), cs, N.ptr(td), N.ptr(o0), N.ptr(o1) if nblocks == 2 else None, C,
                           nblocks, rows, n, rps, N.ptr(ws), ws.numel(), N.stream_ptr()), "dq_level_fwd")
    torch.cuda.synchronize()
    assert rel(o0, outs[0]) < 1e-5
    if nblocks == 2:
        assert rel(o1, outs[1]) < 1e-5
    # the same launch fed from a prepared operand image (k_level_images; the network path): the same numbers, bit for bit
    p0, p1 = torch.full_like(o0, float("nan")), torch.full_like(o1, float("nan"))
    ws2 = torch.empty(2 * B * 2 * C + 8256, device="cuda")
    N.check(L.dq_level_fwd(N.ptr(params), pre, N.ptr(xd), cp, N.ptr(s0), N.ptr(s1), cs, N.ptr(td), N.ptr(p0), N.ptr(p1) if nblocks == 2 else None, C,
                           nblocks, rows, n, rps, N.ptr(ws2), ws2.numel(), N.stream_ptr()), "dq_level_fwd")
    torch.cuda.synchronize()
    assert torch.equal(p0, o0)
    if nblocks == 2:
        assert torch.equal(p1, o1)


# ------------------------------------------------------------------------------------------------ bottleneck attention (unet1d.py:428-443)
def _attn_ref(q, k, v):
    B, _, RT = q.shape
    h = lambda t: t.reshape(B, 4, 32, RT).transpose(2, 3)  # b (h c) n -> b h n c
    sim = torch.einsum("bhid,bhjd->bhij", h(q), h(k)) * 32 ** -0.5
    o = torch.einsum("bhij,bhjd->bhid", sim.softmax(dim=-1), h(v))
    return o.transpose(2, 3).reshape(B, 128, RT)


# (164, 397): a grid that fills the SIMDs several times over
@pytest.mark.parametrize("B,RT", [(2, 400), (1, 2000), (2, 413), (1, 1999), (3, 31), (1, 33), (164, 397)])
def test_attention_fwd_bwd_standalone(N, B, RT):
    """softmax(q k^T / sqrt(32)) v over RT at the bench's 13 key blocks (RT = 400), configs[4]'s 63 (RT = 2000) and ragged tails
    (RT % 32 != 0, rows off 16-byte alignment): forward and dQ / dK / dV against autograd over the plain formula"""
    gen = torch.Generator().manual_seed(RT)
    q, k, v = (torch.randn(B, 128, RT, generator=gen).requires_grad_() for _ in range(3))
    go = torch.randn(B, 128, RT, generator=gen)
    o = _attn_ref(q, k, v)
    (o * go).sum().backward()
    L = N.lib()
    qd, kd, vd, god = q.detach().cuda(), k.detach().cuda(), v.detach().cuda(), go.cuda()
    od, lse, delta = torch.empty_like(qd), torch.empty(B * 4 * RT, device="cuda"), torch.empty(B * 4 * RT, device="cuda")
    N.check(L.dq_attn_fwd(N.ptr(qd), N.ptr(kd), N.ptr(vd), N.ptr(od), N.ptr(lse), B, RT, N.stream_ptr()), "dq_attn_fwd")
    dq, dk, dv = (torch.full_like(qd, float("nan")) for _ in range(3))  # plain stores: every element must be written
    N.check(L.dq_attn_bwd(N.ptr(qd), N.ptr(kd), N.ptr(vd), N.ptr(od), N.ptr(god), N.ptr(lse), N.ptr(delta), N.ptr(dq), N.ptr(dk), N.ptr(dv),
                          B, RT, N.stream_ptr()), "dq_attn_bwd")
    torch.cuda.synchronize()
    assert rel(od, o) < 1e-5
    assert rel(dq, q.grad) < 2e-5 and rel(dk, k.grad) < 2e-5 and rel(dv, v.grad) < 2e-5


# ------------------------------------------------------------------------------------------------ RoPE (parity unpinned): properties
def _rope(N, t, freqs, sign=1.0, stride=None):
    out = t.clone()
    B, C, RT = t.shape
    N.check(N.lib().dq_rope(N.ptr(out), N.ptr(freqs), B, stride or C * RT, RT, sign, N.stream_ptr()), "dq_rope")
    return out


def test_rope_properties(N):
    """k_rope forward / backward without a fixture of the third-party package: (1) equals the oracle's restatement, (2) rotations
    preserve the norm of every (position, head) vector, (3) channels 16..31 of each head pass through untouched, (4) position 0 is
    the identity, (5) q.k scores depend only on the position DIFFERENCE (a common shift of both leaves them unchanged),
    (6) sign = -1 is the transpose = inverse (what the backward applies), (7) a sample's channels beyond the first 128 (the v half
    of the q|v projection) are not touched."""
    from oracle import dq_oracle as O

    B, RT = 2, 77
    gen = torch.Generator().manual_seed(4)
    fr = O.rope_freqs().cuda()
    t = torch.randn(B, 256, RT, generator=gen).cuda()
    r = _rope(N, t, fr, 1.0, stride=256 * RT)
    assert torch.equal(r[:, 128:], t[:, 128:])                                                        # (7)
    th = t[:, :128].reshape(B, 4, 32, RT)
    rh = r[:, :128].reshape(B, 4, 32, RT)
    ref = O.rope_rotate(th.transpose(2, 3).cpu(), fr.cpu()).transpose(2, 3)
    assert rel(rh, ref) < 2e-6                                                                        # (1)
    assert float((rh.norm(dim=2) - th.norm(dim=2)).abs().max() / th.norm(dim=2).max()) < 1e-6        # (2)
    assert torch.equal(rh[:, :, 16:], th[:, :, 16:])                                                  # (3)
    assert torch.equal(rh[..., 0], th[..., 0])                                                        # (4)
    back = _rope(N, r, fr, -1.0, stride=256 * RT)
    assert rel(back, t) < 2e-6                                                                        # (6)
    # (5): scores of (q at i, k at j) == scores of the same vectors placed at (i + s, j + s)
    q, k = torch.randn(1, 128, RT, generator=gen).cuda(), torch.randn(1, 128, RT, generator=gen).cuda()
    s = 9
    qs, ks = torch.zeros_like(q), torch.zeros_like(k)
    qs[..., s:], ks[..., s:] = q[..., :-s], k[..., :-s]
    sc = lambda a, b: torch.einsum("hci,hcj->hij", a[0].reshape(4, 32, RT), b[0].reshape(4, 32, RT))
    s0 = sc(_rope(N, q, fr), _rope(N, k, fr))[:, :RT - s, :RT - s]
    s1 = sc(_rope(N, qs, fr), _rope(N, ks, fr))[:, s:, s:]
    assert rel(s1, s0) < 1e-5
    # adjoint: <rope(a), b> == <a, rope^T(b)> -- the backward kernel really is the transpose of the forward
    a, b = torch.randn(B, 128, RT, generator=gen).cuda(), torch.randn(B, 128, RT, generator=gen).cuda()
    lhs, rhs = float((_rope(N, a, fr) * b).sum()), float((a * _rope(N, b, fr, -1.0)).sum())
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs))
