"""GPU parity, backward side: hand-written HIP backward kernels against torch autograd over the oracle and against the
gradients / optimiser trajectories captured from the reference."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import sub

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def rel_err(a, b):
    def f(v):
        return v.detach().float().cpu() if torch.is_tensor(v) else torch.as_tensor(np.asarray(v)).float()

    a, b = f(a), f(b)
    return float((a - b.reshape(a.shape)).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.fixture(scope="module")
def N():
    from dquartic import _native

    _native.lib()
    return _native


@pytest.mark.parametrize("C,n,rows", [(4, 64, 37), (4, 32, 9), (8, 32, 70), (8, 16, 13), (8, 8, 22), (12, 8, 5), (12, 4, 50), (12, 2, 33),
                                      (16, 2, 16), (16, 1, 77), (4, 64, 1300), (4, 128, 21), (4, 256, 9), (8, 128, 6),
                                      # run-time row lengths (k_la_long.hip with N = 0): ragged last block, rows shorter than a block,
                                      # and the lengths of the reference's shipped downsample_dim 40000 (625 = 19 blocks + 17)
                                      (4, 320, 5), (4, 160, 3), (8, 80, 4), (8, 40, 6), (12, 20, 7), (12, 10, 5), (16, 5, 9), (16, 625, 2),
                                      (4, 1000, 2), (8, 96, 3), (12, 3, 4),
                                      # rows of 16 positions at every width (the per-row M / P form), rows of 8 / 4 at the others, more rows than one
                                      # resident round of four-wave blocks, and one-position rows at every width (the closed form)
                                      (4, 16, 11), (12, 16, 7), (16, 16, 5), (4, 8, 10), (16, 8, 9), (16, 4, 3), (8, 16, 9000), (4, 1, 40), (8, 1, 33),
                                      (12, 1, 100),
                                      # the instantiations no level of the default network uses (the widest hold 110-160 KB of LDS per workgroup)
                                      (8, 64, 5), (12, 32, 4), (16, 32, 3), (12, 64, 3), (16, 64, 2), (8, 4, 9), (8, 2, 17), (4, 4, 6), (4, 2, 40)])
def test_linattn_bwd_vs_autograd(N, C, n, rows):
    _la_bwd_case(N, C, n, rows)


@pytest.mark.parametrize("form", ["default", "rows", "register"])
@pytest.mark.parametrize("C,n,rows,wscale", [(12, 4, 50, 0.4), (12, 2, 33, 0.4), (16, 2, 16, 0.4), (16, 4, 3, 0.4), (8, 4, 9, 0.4), (8, 2, 17, 0.4),
                                             (12, 4, 1, 0.4), (16, 4, 15, 0.4), (16, 4, 17, 0.4), (12, 2, 4100, 0.4), (16, 2, 20000, 0.4), (12, 4, 12800, 0.4),
                                             # logits beyond the bounded-softmax criterion (k_linattn_prepare): the shifted form of both softmaxes
                                             (12, 4, 70, 3.0), (16, 2, 40, 3.0)])
def test_linattn_bwd_forms_vs_autograd(N, C, n, rows, wscale, form):
    """Rows of 2 / 4 positions at 8 / 12 / 16 channels in BOTH backward forms -- one m/z row per lane column on the 16x16x4 matrix pipe
    (k_la_rows_bwd.hip, the product's default) and the register-resident tiles (k_la_bwd.hip) -- against the oracle's autograd: ragged last
    tiles, a single row, more tiles than one resident round of waves (20,000 rows = 1,250 tiles), the train step's own row count (12,800).
    The PREPARED forward of the same case (the network's path) is held against the oracle too: under `default` that is k_la_rows_fwd.hip,
    under `rows` k_la_small, under `register` the register-resident kernel."""
    from conftest import set_la_form

    set_la_form(form)
    try:
        _la_bwd_case(N, C, n, rows, wscale)
    finally:
        set_la_form("default")


def _la_bwd_case(N, C, n, rows, wscale=0.4):
    from oracle import dq_oracle as O

    gen = torch.Generator().manual_seed(1000 * C + n)
    x = torch.randn(rows, C, n, generator=gen).requires_grad_()
    p = {"la.fn.norm.g": (torch.rand(1, C, 1, generator=gen) + 0.5).requires_grad_(),
         "la.fn.fn.to_qkv.weight": (torch.randn(384, C, 1, generator=gen) * wscale).requires_grad_(),
         "la.fn.fn.to_out.0.weight": (torch.randn(C, 128, 1, generator=gen) * 0.2).requires_grad_(),
         "la.fn.fn.to_out.0.bias": (torch.randn(C, generator=gen) * 0.1).requires_grad_(),
         "la.fn.fn.to_out.1.g": (torch.rand(1, C, 1, generator=gen) + 0.5).requires_grad_()}
    y = O.linear_attention(p, "la", x)
    gy = torch.randn(y.shape, generator=gen)
    (y * gy).sum().backward()

    d = {k: v.detach().cuda().reshape(v.shape[0] if v.dim() == 1 else -1).contiguous() for k, v in p.items()}
    xd, gyd = x.detach().cuda(), gy.cuda()
    yd, ypre = torch.empty_like(xd), torch.empty_like(xd)
    L = N.lib()
    w, wo, bo, g1, g2 = (d["la.fn.fn.to_qkv.weight"], d["la.fn.fn.to_out.0.weight"], d["la.fn.fn.to_out.0.bias"], d["la.fn.norm.g"],
                         d["la.fn.fn.to_out.1.g"])
    N.check(L.dq_linattn_fwd(N.ptr(xd), N.ptr(yd), N.ptr(ypre), N.ptr(w), N.ptr(wo), N.ptr(bo), N.ptr(g1), N.ptr(g2), C, rows, n,
                             N.stream_ptr()), "dq_linattn_fwd")
    dx = torch.zeros_like(xd)
    dw, dwo, dbo, dg1, dg2 = (torch.zeros_like(t) for t in (w, wo, bo, g1, g2))
    scratch = torch.empty(2 * xd.numel() + 2048 * 512 * C, device="cuda")
    N.check(L.dq_linattn_bwd(N.ptr(xd), N.ptr(ypre), N.ptr(gyd), N.ptr(dx), N.ptr(w), N.ptr(wo), N.ptr(bo), N.ptr(g1), N.ptr(g2),
                             N.ptr(dw), N.ptr(dwo), N.ptr(dbo), N.ptr(dg1), N.ptr(dg2), N.ptr(scratch), C, rows, n, N.stream_ptr()),
            "dq_linattn_bwd")
    if n <= 64 and (n & (n - 1)) == 0:  # the network's forward: prepared weights / operand images (its kernel depends on the dispatch rule in force)
        prep = torch.zeros(L.dq_linattn_prep_floats(), device="cuda")
        yp, ypre_p = torch.empty_like(xd), torch.empty_like(xd)
        N.check(L.dq_linattn_prepare(N.ptr(w), N.ptr(wo), N.ptr(g1), C, N.ptr(prep), N.stream_ptr()), "dq_linattn_prepare")
        N.check(L.dq_linattn_fwd_prepared(N.ptr(xd), N.ptr(yp), N.ptr(ypre_p), N.ptr(w), N.ptr(wo), N.ptr(bo), N.ptr(g1), N.ptr(g2), N.ptr(prep),
                                          C, rows, n, N.stream_ptr()), "dq_linattn_fwd_prepared")
        torch.cuda.synchronize()
        assert rel_err(yp, y) < 1e-5
        assert rel_err(ypre_p, ypre.cpu()) < 1e-5
    torch.cuda.synchronize()
    assert rel_err(yd, y) < 1e-5
    tol = 2e-5 if (rows < 1000 and wscale < 1.0) else 1e-4  # long fp32 sums (in a fixed order: no atomics) lose a little; so do sharp softmaxes
    assert rel_err(dx, x.grad) < tol
    assert rel_err(dw, p["la.fn.fn.to_qkv.weight"].grad) < tol
    assert rel_err(dwo, p["la.fn.fn.to_out.0.weight"].grad) < tol
    assert rel_err(dbo, p["la.fn.fn.to_out.0.bias"].grad) < tol
    assert rel_err(dg1, p["la.fn.norm.g"].grad) < tol
    assert rel_err(dg2, p["la.fn.fn.to_out.1.g"].grad) < tol


def _default_net(g):
    from dquartic.model.unet1d import UNet1d

    net = UNet1d(dim=4, channels=1, dim_mults=(1, 2, 2, 3, 3, 4, 4), conditional=True, init_cond_channels=1,
                 attn_cond_channels=1, tfer_dim_mult=620, downsample_dim=64, simple=True)
    net.load_state_dict(sub(g, "w/"))
    return net.cuda()


@pytest.mark.parametrize("tag,use_rope", [("norope", False), ("rope", True)])
def test_whole_net_grads_golden(golden, tag, use_rope, la_form):
    """all 395 parameter gradients + d/dx through loss.backward() (autograd bridge -> dq_unet_bwd) vs the reference's"""
    g = golden("unet_default_rt16.npz")
    net = _default_net(g)
    net.use_rope = use_rope
    x = T(g["x"]).cuda().requires_grad_()
    y = net(x, T(g["t"]).cuda(), T(g["init_cond"]).cuda(), T(g["attn_cond"]).cuda())
    assert rel_err(y, g[f"{tag}/y"]) < 2e-5
    (y * T(g["gout"]).cuda()).sum().backward()
    torch.cuda.synchronize()
    assert rel_err(x.grad, g[f"{tag}/dx"]) < 1e-4
    ref = sub(g, f"{tag}/grad/")
    # a gradient that is analytically zero (e.g. the key bias without RoPE: softmax is shift-invariant) is pure
    # round-off in the reference too, so errors are measured against max(|ref|, 1e-4 * the largest gradient)
    floor = 1e-4 * max(float(v.abs().max()) for v in ref.values())
    named = dict(net.named_parameters())
    assert len(ref) == 395
    # The yardstick is the oracle evaluated in float64.  A few of these sums cancel heavily and move by ~1e-4 of their value when one conv of
    # the forward pass rounds differently: the REFERENCE's own fp32 fixture is that far from the float64 value on them.  Those tensors are
    # NAMED below with the reference's fp32-vs-float64 distance measured on this fixture (tools/whole_net_errs.py, round 4); the test
    # re-measures that distance and requires it to be >= 2.5e-5 for every name on the list, so the list cannot grow into a blanket
    # allowance.  Bounds: every tensor within 2e-4 of float64; vs the reference's fp32 fixture 1e-4 for every tensor that is not on the list,
    # 4e-4 for the listed ones (two fp32 errors of ~1e-4 each can add up).
    CANCELLING = {
        "ups.5.2.fn.fn.to_out.1.g": 1.17e-4,    # gain of a LinearAttention output norm
        "ups.4.0.block1.norm.g": 5.7e-5,
        "downs.1.3.bias": 4.3e-5,
        "ups.4.1.block1.norm.g": 3.8e-5,
        "ups.4.1.block1.proj.bias": 3.3e-5,
        "ups.4.1.block1.proj.weight": 3.1e-5,
        "downs.1.2.fn.norm.g": 3.1e-5,
    }
    from oracle import dq_oracle as O
    p64 = {k: v.double().clone().requires_grad_(not k.endswith("freqs")) for k, v in sub(g, "w/").items()}
    y64 = O.unet_forward(p64, O.UNetConfig(downsample_dim=64), T(g["x"]).double(), torch.as_tensor(np.asarray(g["t"])), T(g["init_cond"]).double(),
                         T(g["attn_cond"]).double(), use_rope=use_rope)
    (y64 * T(g["gout"]).double()).sum().backward()
    worst, worst_listed, worst64, ref64 = ("", 0.0), ("", 0.0), ("", 0.0), ("", 0.0)
    for k, v in ref.items():
        mine = named[k].grad.detach().cpu()
        e = float((mine - v).abs().max()) / max(float(v.abs().max()), floor)
        t64 = p64[k].grad
        d = max(float(t64.abs().max()), floor)
        e64 = float((mine.double() - t64).abs().max()) / d
        er = float((v.double() - t64).abs().max()) / d
        if k in CANCELLING:
            if not use_rope:  # (the distances above were measured on the norope fixture: the list is justified by measurement)
                assert er >= 2.5e-5, (k, er)
            worst_listed = max(worst_listed, (k, e), key=lambda q: q[1])
        else:
            worst = max(worst, (k, e), key=lambda q: q[1])
        worst64 = max(worst64, (k, e64), key=lambda q: q[1])
        ref64 = max(ref64, (k, er), key=lambda q: q[1])
    print("whole-net gradients:", tag, "vs the reference's fp32", worst, "| listed", worst_listed, "| vs float64", worst64,
          "| the reference's fp32 vs float64", ref64)
    assert worst64[1] < 2e-4, worst64
    assert worst[1] < 1e-4, worst
    assert worst_listed[1] < 4e-4, worst_listed


def _tiny_dm(g):
    from dquartic.model.model import DDIMDiffusionModel
    from dquartic.model.unet1d import UNet1d

    net = UNet1d(dim=4, channels=1, dim_mults=(1, 2), conditional=True, init_cond_channels=1, attn_cond_channels=1,
                 downsample_dim=8, simple=True)
    net.load_state_dict(sub(g, "w/"))
    return DDIMDiffusionModel(model_class=net.cuda(), device="cuda")


def test_train_step_loss_golden(golden):
    g = golden("tiny_diffusion.npz")
    dm = _tiny_dm(g)
    x0, c2, c1 = (T(g[k]).cuda() for k in ("x0", "ms2_cond", "ms1_cond"))
    # generic path (autograd bridge), explicit t; the reference maps a passed noise 2n-1, so pass (n+1)/2
    loss = dm.train_step(x0, c2, c1, noise=(T(g["train/noise"]).cuda() + 1) / 2, t=T(g["train/t"]).cuda())
    assert loss.dim() == 0 and abs(float(loss) - float(g["train/loss"][0])) < 2e-5 * abs(float(g["train/loss"][0]))
    # fused path
    lf = dm.train_step_fused(x0, c2, c1, t=T(g["train/t"]).cuda(), noise=T(g["train/noise"]).cuda())
    assert abs(float(lf) - float(g["train/loss"][0])) < 2e-5 * abs(float(g["train/loss"][0]))
    # batched semantics: loss of a batch == mean over samples of the B = 1 loss
    lb = dm.train_step_fused(T(g["batch/x"]).cuda(), T(g["batch/init_cond"]).cuda(), T(g["batch/attn_cond"]).cuda(),
                             t=T(g["batch/t"]).cuda(), noise=T(g["batch/noise"]).cuda())
    assert abs(float(lb) - float(g["batch/loss_mean"])) < 2e-5 * abs(float(g["batch/loss_mean"]))


def test_fused_grads_equal_autograd_bridge(golden):
    g = golden("tiny_diffusion.npz")
    dm = _tiny_dm(g)
    net = dm.model
    xb, cb, mb = (T(g[k]).cuda() for k in ("batch/x", "batch/init_cond", "batch/attn_cond"))
    t, nz = T(g["batch/t"]).cuda(), T(g["batch/noise"]).cuda()
    dm.train_step_fused(xb, cb, mb, t=t, noise=nz)
    fused = net.flat_grads().clone()
    net.flat_grads(zero=True)
    loss = dm.train_step(xb, cb, mb, noise=(nz + 1) / 2, t=t)
    loss.backward()
    bridge = torch.cat([p.grad.reshape(-1) for _, p in net.trainable_named()])
    assert rel_err(fused, bridge) < 1e-4


def test_optimizer_trajectory_golden(golden):
    """_train_one_batch x3 (zero_grad, train_step, backward, clip 10, AdamW lr=1e-5) vs the reference's parameters"""
    g = golden("tiny_diffusion.npz")
    dm = _tiny_dm(g)
    lr = float(g["opt/lr"])
    dm._set_optimizer(lr)
    x0, c2, c1 = (T(g[k]).cuda() for k in ("x0", "ms2_cond", "ms1_cond"))
    for step in range(3):
        loss = dm._train_one_batch(x0, ms2_cond=c2, ms1_cond=c1, noise=(T(g["opt/noise"])[step:step + 1].cuda() + 1) / 2,
                                   t=T(g["opt/t"])[step:step + 1].cuda())
        assert abs(loss - g["opt/losses"][step]) <= 5e-5 * abs(g["opt/losses"][step])
        assert abs(float(dm.last_grad_norm) - g["opt/gnorms"][step]) <= 5e-4 * g["opt/gnorms"][step]
        if step in (0, 2):
            sd = dm.model.state_dict()
            for k, v in sub(g, f"opt/after{step + 1}/").items():
                # an AdamW step moves a weight by ~lr: check the displacement, not just the value
                assert float((sd[k].cpu() - v).abs().max()) <= 2e-7 + 0.05 * lr, (step, k)


def test_adamw_clip_matches_torch(N):
    torch.manual_seed(0)
    n = 128847
    p0, g0 = torch.randn(n), torch.randn(n) * 0.3  # norm >> 10 => clipping active
    ref = p0.clone().requires_grad_()
    opt = torch.optim.AdamW([ref], lr=1e-3)
    p, m, v, scratch, gn = p0.cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), torch.empty(1024, device="cuda"), torch.zeros((), device="cuda")
    for step in range(1, 4):
        gr = g0 * step
        ref.grad = gr.clone()
        tn = torch.nn.utils.clip_grad_norm_([ref], 10.0)
        opt.step()
        gd = gr.cuda()
        N.check(N.lib().dq_adamw_clip_step(N.ptr(p), N.ptr(gd), N.ptr(m), N.ptr(v), n, N.ptr(scratch), 1.0, 10.0, 1e-3, 0.9, 0.999, 1e-8,
                                           0.01, step, N.ptr(gn), N.stream_ptr()), "dq_adamw_clip_step")
        torch.cuda.synchronize()
        assert abs(float(gn) - float(tn)) < 1e-4 * float(tn)
        assert float((p.cpu() - ref.detach()).abs().max()) < 2e-6


def test_large_window_config_forward_and_grads_vs_oracle():
    """BASELINE configs[4] family: MZ = 256 (downsample_dim 256 -> LinearAttention rows of 256/128 positions, bottleneck width
    64), reduced RT so that the oracle finishes in seconds.  Loss and all parameter gradients against oracle autograd."""
    from dquartic.model.model import DDIMDiffusionModel
    from dquartic.model.unet1d import UNet1d
    from oracle import dq_oracle as O

    torch.manual_seed(3)
    net = UNet1d(dim=4, channels=1, dim_mults=(1, 2, 2, 3, 3, 4, 4), conditional=True, init_cond_channels=1,
                 attn_cond_channels=1, downsample_dim=256, simple=True)
    params = {k: v.detach().clone() for k, v in net.state_dict().items()}
    dm = DDIMDiffusionModel(model_class=net.cuda(), device="cuda")
    B, RT, MZ = 2, 24, 256
    g = torch.Generator().manual_seed(1)
    x0, c2, c1 = torch.rand(B, RT, MZ, generator=g), torch.rand(B, RT, MZ, generator=g), torch.rand(B, RT, generator=g)
    t, nz = torch.tensor([11, 871]), torch.randn(B, RT, MZ, generator=g)
    loss = dm.train_step_fused(x0.cuda(), c2.cuda(), c1.cuda(), t=t.cuda(), noise=nz.cuda())
    net.flat_grads()
    # (per-parameter views, not the flat buffer: the 64-channel bottleneck runs on the wide path since round 4, whose plan starts every
    # bottleneck tensor on a 16-byte boundary -- the flat buffers hold 2 alignment floats that belong to no parameter)
    grads = torch.cat([p.grad.reshape(-1) for _, p in net.trainable_named()]).clone()
    po = {k: v.clone().requires_grad_(not k.endswith("freqs")) for k, v in params.items()}
    lo, eps_o = O.Diffusion(po, O.UNetConfig(downsample_dim=256)).train_loss(x0, c2, c1, t, nz)
    lo.backward()
    assert abs(float(loss) - float(lo)) < 2e-5 * abs(float(lo))
    ref = torch.cat([po[n].grad.reshape(-1) for n, _ in net.trainable_named()])
    assert rel_err(grads, ref) < 2e-4
    # forward alone (inference path) at the same shape
    with torch.no_grad():
        xt = O.q_sample(O.make_schedule()["alpha_bars"], O.normalize(x0), t, nz)
        y = net(xt.cuda(), t.cuda(), O.normalize(c2).cuda(), O.normalize(c1).cuda())
    assert rel_err(y, eps_o) < 5e-5


@pytest.mark.parametrize("RT", [70, 96])
def test_default_net_grads_multiblock_attention(RT):
    """RT > 32: the bottleneck attention sweeps several 32-position blocks (RT = 70: ragged tail + unaligned rows, scalar
    operand loads; RT = 96: whole blocks, 16-byte operand loads).  Loss, eps and all parameter gradients vs the oracle's autograd."""
    from oracle import dq_oracle as O
    from dquartic.model.model import DDIMDiffusionModel
    from dquartic.model.unet1d import UNet1d

    torch.manual_seed(9)
    net = UNet1d(dim=4, channels=1, dim_mults=(1, 2, 2, 3, 3, 4, 4), conditional=True, init_cond_channels=1,
                 attn_cond_channels=1, downsample_dim=64, simple=True)
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.05 * torch.randn_like(p))
    po = {k: v.detach().clone().cpu() for k, v in net.state_dict().items()}
    dm = DDIMDiffusionModel(model_class=net.cuda(), device="cuda")
    od = O.Diffusion(po, O.UNetConfig(downsample_dim=64))
    B, MZ = 2, 64
    gen = torch.Generator().manual_seed(RT)
    x0, c2, c1 = torch.rand(B, RT, MZ, generator=gen), torch.rand(B, RT, MZ, generator=gen), torch.rand(B, RT, generator=gen)
    t, nz = torch.tensor([40, 700]), torch.randn(B, RT, MZ, generator=gen)
    keys = O.trainable_keys(po)
    for k in keys:
        po[k].requires_grad_(True)
    lo, eps_o = od.train_loss(x0, c2, c1, t, nz)
    lo.backward()
    net.train()
    loss = dm.train_step_fused(x0.cuda(), c2.cuda(), c1.cuda(), t=t.cuda(), noise=nz.cuda())
    assert abs(float(loss) - float(lo)) < 2e-5 * abs(float(lo))
    gmax = max(float(po[k].grad.abs().max()) for k in keys)
    named = dict(net.named_parameters())
    for k in keys:
        ref = po[k].grad
        err = float((named[k].grad.cpu() - ref).abs().max())
        assert err <= 2e-4 * max(float(ref.abs().max()), 1e-4 * gmax), (k, err)


@pytest.mark.parametrize("qk_scale,gtol", [(1.0, 2e-4), (30.0, 1e-2)])
def test_linattn_softmax_forms_through_the_whole_net(qk_scale, gtol):
    """The LinearAttention kernels evaluate softmax without the shift by the row maximum when the layer's logits are bounded for
    every input (log2(e) |W_row| sqrt(C) max|g_pre| <= 64, decided per layer by k_linattn_prepare) and with it otherwise.  Default
    initialisation takes the first form; to_qkv rows scaled 30x push every layer into the second.  Same function: loss and all
    parameter gradients vs the oracle's autograd in both.  (With logits of +-100 the softmaxes are nearly one-hot and the gradient is
    ill-conditioned in fp32: the fp32 oracle itself is 5e-3 away from its float64 evaluation there, hence the 1e-2 for that case.)"""
    from oracle import dq_oracle as O
    from dquartic.model.model import DDIMDiffusionModel
    from dquartic.model.unet1d import UNet1d

    torch.manual_seed(21)
    net = UNet1d(dim=4, channels=1, dim_mults=(1, 2, 2, 3, 3, 4, 4), conditional=True, init_cond_channels=1,
                 attn_cond_channels=1, downsample_dim=64, simple=True)
    with torch.no_grad():
        for k, p in net.named_parameters():
            p.add_(0.05 * torch.randn_like(p))
        sd = net.state_dict()
        n_scaled = 0
        for k in sd:
            if k.endswith("to_qkv.weight") and sd[k].shape[0] == 384:  # LinearAttention: q | k | v rows; scale q and k
                sd[k][:256] *= qk_scale
                n_scaled += 1
        net.load_state_dict(sd)
    assert n_scaled == 14
    # the kernel's criterion, restated: which form does each layer take?
    sd = net.state_dict()
    for k in sd:
        if k.endswith("to_qkv.weight") and sd[k].shape[0] == 384:
            C = sd[k].shape[1]
            g = sd[k.replace("fn.to_qkv.weight", "norm.g")].abs().max()
            bound = 1.4426950408889634 * float(sd[k][:256, :, 0].norm(dim=1).max()) * C ** 0.5 * float(g)
            assert (bound <= 64.0) == (qk_scale == 1.0), (k, bound)
    po = {k: v.detach().clone().cpu() for k, v in net.state_dict().items()}
    dm = DDIMDiffusionModel(model_class=net.cuda(), device="cuda")
    od = O.Diffusion(po, O.UNetConfig(downsample_dim=64))
    B, RT, MZ = 2, 24, 64
    gen = torch.Generator().manual_seed(5)
    x0, c2, c1 = torch.rand(B, RT, MZ, generator=gen), torch.rand(B, RT, MZ, generator=gen), torch.rand(B, RT, generator=gen)
    t, nz = torch.tensor([3, 911]), torch.randn(B, RT, MZ, generator=gen)
    keys = O.trainable_keys(po)
    for k in keys:
        po[k].requires_grad_(True)
    lo, _ = od.train_loss(x0, c2, c1, t, nz)
    lo.backward()
    net.train()
    loss = dm.train_step_fused(x0.cuda(), c2.cuda(), c1.cuda(), t=t.cuda(), noise=nz.cuda())
    assert abs(float(loss.detach()) - float(lo.detach())) < 2e-5 * abs(float(lo.detach()))
    gmax = max(float(po[k].grad.abs().max()) for k in keys)
    named = dict(net.named_parameters())
    for k in keys:
        ref = po[k].grad
        err = float((named[k].grad.cpu() - ref).abs().max())
        assert err <= gtol * max(float(ref.abs().max()), 1e-4 * gmax), (k, err)


def test_two_forwards_before_backward_keep_their_own_activations(golden):
    """two evaluations with grad enabled before one backward (micro-batches whose losses are summed) -- each forward
    owns its training workspace until its backward ran, so the summed gradient equals the sum of the separate gradients."""
    g = golden("tiny_diffusion.npz")
    dm = _tiny_dm(g)
    net = dm.model
    xb, cb, mb, t = (T(g[k]).cuda() for k in ("batch/x", "batch/init_cond", "batch/attn_cond", "batch/t"))

    def grads_of(fn):
        net.flat_grads(zero=True)
        for p in net.parameters():
            p.grad = None
        fn().backward()
        return torch.cat([p.grad.reshape(-1) for _, p in net.trainable_named()]).clone()

    f1 = lambda: (net(xb[:1], t[:1], cb[:1], mb[:1]) ** 2).mean()
    f2 = lambda: (net(xb[1:], t[1:], cb[1:], mb[1:]) ** 3).mean()
    g1, g2 = grads_of(f1), grads_of(f2)
    both = grads_of(lambda: f1() + f2())   # forward 1, forward 2, then ONE backward through both
    assert rel_err(both, g1 + g2) < 1e-5
    # a fused step in between a forward and its backward must not disturb it either
    y = net(xb[:1], t[:1], cb[:1], mb[:1])
    dm.train_step_fused(xb, cb, mb, t=t, noise=T(g["batch/noise"]).cuda())
    for p in net.parameters():
        p.grad = None
    (y ** 2).mean().backward()
    again = torch.cat([p.grad.reshape(-1) for _, p in net.trainable_named()])
    assert rel_err(again, g1) < 1e-5


def test_train_one_batch_under_rccl_process_group_world1():
    """The data-parallel product path on the GPU: an RCCL ("nccl") process group of one rank, then ``_prepare_training`` (replica
    sync) and ``_train_one_batch`` -- the flat-gradient all-reduce line, ``grad_scale = 1 / world`` and the optimiser -- against the
    same step without a process group (world = 1: the sum over ranks is the identity, so the two must agree bit for bit)."""
    import os

    import torch.distributed as dist
    from dquartic.model.model import DDIMDiffusionModel
    from dquartic.model.unet1d import UNet1d

    def run(with_pg):
        torch.manual_seed(21)
        net = UNet1d(dim=4, channels=1, dim_mults=(1, 2, 2, 3, 3, 4, 4), conditional=True, init_cond_channels=1,
                     attn_cond_channels=1, downsample_dim=64, simple=True).cuda()
        dm = DDIMDiffusionModel(model_class=net, device="cuda")
        dm._prepare_training(1e-4)
        gen = torch.Generator().manual_seed(2)
        x0, c2, c1 = torch.rand(4, 40, 64, generator=gen).cuda(), torch.rand(4, 40, 64, generator=gen).cuda(), torch.rand(4, 40, generator=gen).cuda()
        t, nz = torch.tensor([1, 300, 640, 999]).cuda(), torch.rand(4, 40, 64, generator=gen).cuda()
        losses = [dm._train_one_batch(x0, ms2_cond=c2, ms1_cond=c1, noise=nz, t=t) for _ in range(3)]
        return losses, net.flat_params.clone(), float(dm.last_grad_norm), dm._global_mean(losses[-1])

    def side_stream_is_joined_in_front_of_the_allreduce():
        """dq_train_step runs its weight-gradient launches on a side stream and joins it to the caller's stream before it returns; the flat
        gradient all-reduce that follows on the caller's stream relies on exactly that.  The test hook makes the side stream's LAST action a
        store into the gradient buffer, delayed by 5 ms: the all-reduce (and a copy queued behind it, before any host synchronisation) must
        already see the stored value."""
        from dquartic import _native as N

        torch.manual_seed(22)
        net = UNet1d(dim=4, channels=1, dim_mults=(1, 2, 2, 3, 3, 4, 4), conditional=True, init_cond_channels=1,
                     attn_cond_channels=1, downsample_dim=64, simple=True).cuda()
        dm = DDIMDiffusionModel(model_class=net, device="cuda")
        gen = torch.Generator().manual_seed(3)
        x0, c2, c1 = torch.rand(2, 40, 64, generator=gen).cuda(), torch.rand(2, 40, 64, generator=gen).cuda(), torch.rand(2, 40, generator=gen).cuda()
        grads = net.flat_grads()
        dm.train_step_fused(x0, c2, c1)  # (warm-up: workspace, side stream, events)
        torch.cuda.synchronize()
        idx, poison = 7, 12345.0
        N.check(N.lib().dq_debug_side_tail_store(net._plan, ctypes.c_void_p(grads.data_ptr() + 4 * idx), poison, 5000), "dq_debug_side_tail_store")
        try:
            dm.train_step_fused(x0, c2, c1)
            dist.all_reduce(grads)              # the data-parallel exchange of model_interface.py, on the caller's stream
            seen = grads[idx].clone()           # queued behind it, no host synchronisation in between
        finally:
            N.check(N.lib().dq_debug_side_tail_store(net._plan, None, 0.0, 0), "dq_debug_side_tail_store")
        torch.cuda.synchronize()
        return float(seen)

    ref = run(False)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 200))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        got = run(True)
        seen = side_stream_is_joined_in_front_of_the_allreduce()
    finally:
        dist.destroy_process_group()
    assert got[0] == ref[0] and torch.equal(got[1], ref[1]) and got[2] == ref[2] and got[3] == ref[3]
    assert all(np.isfinite(l) for l in got[0])
    assert seen == 12345.0, seen


def test_captured_train_step_equals_eager_bit_for_bit():
    """``enable_train_graph`` replays one captured hipGraph per optimiser step (draws, q_sample, forward, loss, backward,
    clip + AdamW with the step count / lr in device memory).  Same kernels, same order: after five steps from the same seed the parameters,
    both moments and every loss equal the eager run's bit for bit -- and building the graph is not a training step."""
    from dquartic.model.model import DDIMDiffusionModel
    from dquartic.model.unet1d import UNet1d

    def make():
        torch.manual_seed(3)
        net = UNet1d(dim=4, channels=1, dim_mults=(1, 2, 2, 3, 3, 4, 4), conditional=True, init_cond_channels=1, attn_cond_channels=1,
                     downsample_dim=64, simple=True).cuda()
        dm = DDIMDiffusionModel(model_class=net, device="cuda")
        dm._set_optimizer(1e-3)
        return net, dm

    g = torch.Generator().manual_seed(5)
    B, RT, MZ = 4, 48, 64
    data = [(torch.rand(B, RT, MZ, generator=g).cuda(), torch.rand(B, RT, MZ, generator=g).cuda(), torch.rand(B, RT, generator=g).cuda()) for _ in range(5)]
    # eager, with the optimiser step in its device-state form
    net_e, dm_e = make()
    torch.manual_seed(11)
    losses_e = []
    for x0, c2, c1 in data:
        loss = dm_e.train_step_fused(x0, c2, c1, zero_grads=True)
        dm_e.optimizer.step_dev()
        losses_e.append(loss.clone())
    # captured
    net_g, dm_g = make()
    dm_g.enable_train_graph()
    torch.manual_seed(11)
    losses_g = [dm_g._train_one_batch(x0, ms2_cond=c2, ms1_cond=c1, sync=False).clone() for x0, c2, c1 in data]
    torch.cuda.synchronize()
    assert dm_g.optimizer._step == dm_e.optimizer._step == 5
    assert int(dm_g.optimizer._step_dev.item()) == 5
    for a, b in zip(losses_g, losses_e):
        assert torch.equal(a, b), (float(a), float(b))
    assert torch.equal(net_g.flat_params, net_e.flat_params)
    assert torch.equal(dm_g.optimizer._m, dm_e.optimizer._m) and torch.equal(dm_g.optimizer._v, dm_e.optimizer._v)
    # the device-state step against the host-side one: the same update up to the last bit of the bias-correction scalars
    net_h, dm_h = make()
    torch.manual_seed(11)
    for x0, c2, c1 in data:
        dm_h._train_one_batch(x0, ms2_cond=c2, ms1_cond=c1, sync=False)
    # (the bias-correction scalars come from pow() on the device there and on the host here: a parameter may differ in its last bit)
    assert float((net_h.flat_params - net_e.flat_params).abs().max()) <= 4 * 1.1920929e-07 * float(net_e.flat_params.abs().max())
    dm_g.enable_train_graph(False)
