"""GPU parity, forward side: the HIP path (through the C ABI) against the committed golden vectors captured from the
reference and against the oracle on seeded inputs."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import sub

pytestmark = pytest.mark.gpu

T = torch.from_numpy


def rel_err(a, b):
    a, b = a.detach().float().cpu(), torch.as_tensor(b).float()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.fixture(scope="module")
def N():
    from dquartic import _native

    _native.lib()
    return _native


def _dev(x):
    return torch.as_tensor(np.array(x)).cuda().contiguous()


@pytest.mark.parametrize("C,n", [(4, 64), (4, 32), (8, 16), (12, 4), (12, 2), (16, 1)])
def test_linattn_fwd_golden(N, golden, C, n):
    g = golden("blocks.npz")
    pre = f"la_{C}_{n}/"
    x = _dev(g[pre + "x"])
    w = _dev(g[pre + "w/fn.fn.to_qkv.weight"][:, :, 0])
    wo = _dev(g[pre + "w/fn.fn.to_out.0.weight"][:, :, 0])
    bo = _dev(g[pre + "w/fn.fn.to_out.0.bias"])
    g1 = _dev(g[pre + "w/fn.norm.g"].reshape(-1))
    g2 = _dev(g[pre + "w/fn.fn.to_out.1.g"].reshape(-1))
    y = torch.empty_like(x)
    N.check(N.lib().dq_linattn_fwd(N.ptr(x), N.ptr(y), None, N.ptr(w), N.ptr(wo), N.ptr(bo), N.ptr(g1), N.ptr(g2), C, x.shape[0], n,
                                   N.stream_ptr()), "dq_linattn_fwd")
    torch.cuda.synchronize()
    assert rel_err(y, g[pre + "y"]) < 1e-5  # fp32 tolerance: 1e-5 of the output scale


@pytest.mark.parametrize("C,n", [(4, 64), (4, 32), (8, 16), (8, 8), (4, 2), (12, 4), (16, 1), (12, 2), (16, 2), (16, 4), (12, 8)])
def test_linattn_fwd_prepared_equals_standalone(N, C, n, la_form):
    """dq_linattn_prepare + dq_linattn_fwd_prepared (the network's path: derived weights and the split-bf16 / fp32 operand images formed once
    per parameter state) == dq_linattn_fwd (every workgroup derives them itself), bit for bit; and against the oracle.
    Rows of 2 / 4 positions at 12 / 16 channels and of 8 positions at 12: the prepared path is another kernel (k_la_small: one group of registers per position, every
    product on the 32x32x2 matrix pipe) -- same fp32 tolerance against the oracle, and against the stand-alone kernel instead of bit equality."""
    from oracle import dq_oracle as O

    gen = torch.Generator().manual_seed(7 * C + n)
    rows = 203
    x = torch.randn(rows, C, n, generator=gen)
    p = {"la.fn.norm.g": torch.rand(1, C, 1, generator=gen) + 0.5, "la.fn.fn.to_qkv.weight": torch.randn(384, C, 1, generator=gen) * 0.4,
         "la.fn.fn.to_out.0.weight": torch.randn(C, 128, 1, generator=gen) * 0.2, "la.fn.fn.to_out.0.bias": torch.randn(C, generator=gen) * 0.1,
         "la.fn.fn.to_out.1.g": torch.rand(1, C, 1, generator=gen) + 0.5}
    ref = O.linear_attention(p, "la", x)
    d = {k: v.cuda().reshape(v.shape[0] if v.dim() == 1 else -1).contiguous() for k, v in p.items()}
    xd = x.cuda()
    y0, y1 = torch.empty_like(xd), torch.empty_like(xd)
    L = N.lib()
    args = (N.ptr(d["la.fn.fn.to_qkv.weight"]), N.ptr(d["la.fn.fn.to_out.0.weight"]), N.ptr(d["la.fn.fn.to_out.0.bias"]), N.ptr(d["la.fn.norm.g"]),
            N.ptr(d["la.fn.fn.to_out.1.g"]))
    N.check(L.dq_linattn_fwd(N.ptr(xd), N.ptr(y0), None, *args, C, rows, n, N.stream_ptr()), "dq_linattn_fwd")
    prep = torch.zeros(L.dq_linattn_prep_floats(), device="cuda")
    N.check(L.dq_linattn_prepare(args[0], args[1], args[3], C, N.ptr(prep), N.stream_ptr()), "dq_linattn_prepare")
    N.check(L.dq_linattn_fwd_prepared(N.ptr(xd), N.ptr(y1), None, *args, N.ptr(prep), C, rows, n, N.stream_ptr()), "dq_linattn_fwd_prepared")
    torch.cuda.synchronize()
    # the prepared path is ANOTHER kernel than the stand-alone one: k_la_small with the per-row forms forced, k_la_rows_fwd (one m/z row per lane
    # column, rows of 2 / 4 positions below the k_la_small threshold) under the default rule
    if (la_form == "rows" and ((n in (2, 4) and C in (12, 16)) or (n == 8 and C == 12))) or (la_form == "default" and n in (2, 4) and C in (8, 12, 16)):
        assert rel_err(y1, y0.cpu()) < 1e-5
    else:
        assert torch.equal(y0, y1)
    assert rel_err(y1, ref) < 1e-5


@pytest.mark.parametrize("C,n,rows", [(4, 64, 401), (8, 32, 37), (8, 8, 13), (12, 4, 29), (16, 2, 50), (16, 1, 77), (12, 16, 5)])
def test_linattn_fwd_oracle_ragged(N, C, n, rows):
    """row counts that do not fill the last wave / workgroup"""
    from oracle import dq_oracle as O

    gen = torch.Generator().manual_seed(C * 100 + n)
    x = torch.randn(rows, C, n, generator=gen)
    p = {"la.fn.norm.g": torch.rand(1, C, 1, generator=gen) + 0.5,
         "la.fn.fn.to_qkv.weight": torch.randn(384, C, 1, generator=gen) * 0.4,
         "la.fn.fn.to_out.0.weight": torch.randn(C, 128, 1, generator=gen) * 0.2,
         "la.fn.fn.to_out.0.bias": torch.randn(C, generator=gen) * 0.1,
         "la.fn.fn.to_out.1.g": torch.rand(1, C, 1, generator=gen) + 0.5}
    ref = O.linear_attention(p, "la", x)
    xd = x.cuda()
    y = torch.empty_like(xd)
    d = {k: v.cuda().reshape(v.shape[0] if v.dim() == 1 else -1).contiguous() for k, v in p.items()}
    N.check(N.lib().dq_linattn_fwd(N.ptr(xd), N.ptr(y), None, N.ptr(d["la.fn.fn.to_qkv.weight"]), N.ptr(d["la.fn.fn.to_out.0.weight"]),
                                   N.ptr(d["la.fn.fn.to_out.0.bias"]), N.ptr(d["la.fn.norm.g"]), N.ptr(d["la.fn.fn.to_out.1.g"]),
                                   C, rows, n, N.stream_ptr()), "dq_linattn_fwd")
    torch.cuda.synchronize()
    assert rel_err(y, ref) < 1e-5


def _default_net(g):
    from dquartic.model.unet1d import UNet1d

    net = UNet1d(dim=4, channels=1, dim_mults=(1, 2, 2, 3, 3, 4, 4), conditional=True, init_cond_channels=1,
                 attn_cond_channels=1, tfer_dim_mult=620, downsample_dim=64, simple=True)
    net.load_state_dict(sub(g, "w/"))
    return net.cuda()


@pytest.mark.parametrize("tag,use_rope", [("rope", True), ("norope", False)])
def test_whole_net_forward_golden(golden, tag, use_rope, la_form):
    g = golden("unet_default_rt16.npz")
    net = _default_net(g)
    net.use_rope = use_rope
    with torch.no_grad():
        y = net(_dev(g["x"]), _dev(g["t"]), _dev(g["init_cond"]), _dev(g["attn_cond"]))
    torch.cuda.synchronize()
    assert rel_err(y, g[f"{tag}/y"]) < 2e-5


def test_tiny_net_batched_forward_golden(golden):
    """B = 3 == per-sample loop of the B = 1 reference (batched semantics)"""
    from dquartic.model.unet1d import UNet1d

    g = golden("tiny_diffusion.npz")
    net = UNet1d(dim=4, channels=1, dim_mults=(1, 2), conditional=True, init_cond_channels=1, attn_cond_channels=1,
                 downsample_dim=8, simple=True)
    net.load_state_dict(sub(g, "w/"))
    net = net.cuda()
    with torch.no_grad():
        y = net(_dev(g["batch/x"]), _dev(g["batch/t"]), _dev(g["batch/init_cond"]), _dev(g["batch/attn_cond"]))
    assert rel_err(y, g["batch/y"]) < 2e-5


def test_full_size_forward_vs_oracle(golden, la_form):
    """BASELINE config C1 shape: (B=4, RT=400, MZ=64), default network, against the oracle"""
    from oracle import dq_oracle as O

    g = golden("unet_default_rt16.npz")
    net = _default_net(g)
    gen = torch.Generator().manual_seed(3)
    B, RT, MZ = 4, 400, 64
    x, c2, c1 = torch.randn(B, RT, MZ, generator=gen), torch.rand(B, RT, MZ, generator=gen), torch.rand(B, RT, generator=gen)
    t = torch.tensor([0, 17, 500, 999])
    p = sub(g, "w/")
    with torch.no_grad():
        ref = O.unet_forward(p, O.UNetConfig(downsample_dim=64), x, t, c2, c1)
        y = net(x.cuda(), t.cuda(), c2.cuda(), c1.cuda())
    assert rel_err(y, ref) < 5e-5


def test_q_sample_and_ddim_step(N, golden):
    from oracle import dq_oracle as O

    ab = O.make_schedule(1000, "cosine")["alpha_bars"]
    gen = torch.Generator().manual_seed(0)
    B, RT, MZ = 5, 400, 64
    x0, nz = torch.rand(B, RT, MZ, generator=gen), torch.randn(B, RT, MZ, generator=gen)
    t = torch.tensor([0, 1, 500, 998, 999])
    ref = O.q_sample(ab, O.normalize(x0), t, nz)
    abd, x0d, td, nzd = ab.cuda(), x0.cuda(), t.cuda(), nz.cuda()  # keep the device tensors alive across the calls
    out = torch.empty(B, RT, MZ, device="cuda")
    N.check(N.lib().dq_q_sample(N.ptr(abd), N.ptr(x0d), N.ptr(td), N.ptr(nzd), N.ptr(out), B, RT * MZ, 1, N.stream_ptr()), "dq_q_sample")
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), ref) or rel_err(out, ref) < 2e-7
    for tv in (999, 500, 1, 0):
        refp = O.ddim_update(ab, x0, nz, tv)
        a = ab[tv]
        coef = torch.tensor([a.sqrt(), (1 - a).sqrt(), ab[tv - 1].sqrt() if tv > 0 else -1.0, (1 - ab[tv - 1]).sqrt() if tv > 0 else 0.0]).cuda()
        o = torch.empty(B, RT, MZ, device="cuda")
        N.check(N.lib().dq_ddim_step(N.ptr(x0d), N.ptr(nzd), N.ptr(o), N.ptr(coef), B * RT * MZ, N.stream_ptr()), "dq_ddim_step")
        torch.cuda.synchronize()
        assert rel_err(o, refp) < 1e-6
