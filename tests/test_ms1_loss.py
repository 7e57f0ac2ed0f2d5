"""``ms1_loss_weight > 0`` (reference model.py:364-371, 379-386, 398-402).  The reference's branch cannot run (``torch.max(x, dim=-1)``
returns a tuple that is then divided), so there is no output to pin against: the semantics are CHOSEN, restated
in the oracle, and the kernels (``dq_train_step`` with ms1_loss_weight, ``dq_ms1_loss_fwd_bwd``) are held to the oracle's autograd.
CPU part: the oracle's term against a literal transcription of the reference's lines for the two funcs that do run there."""
import numpy as np
import pytest
import torch

from conftest import sub

T = torch.from_numpy


def test_oracle_term_equals_the_reference_lines_where_they_run():
    """At B = 1 with the MS1 chromatogram as (1, RT, 1), ``sum`` and ``mean`` of the reference's loop (model.py:366-371) run as
    written; ``max`` needs ``.values``.  The oracle's additional term must equal that transcription."""
    import torch.nn.functional as F
    from oracle import dq_oracle as O

    g = torch.Generator().manual_seed(0)
    d = torch.randn(1, 9, 8, generator=g)            # x_t - eps_pred (or x0_pred)
    ms1 = torch.rand(1, 9, generator=g) * 2 - 1      # normalised chromatogram
    ref = torch.zeros(())
    for func in (torch.sum, torch.mean, lambda x, dim: torch.max(x, dim=dim).values):
        sic = func(d, dim=-1)
        ms1_sic = func(ms1[..., None], dim=-1)
        ref = ref + F.mse_loss(sic / torch.max(sic), ms1_sic / torch.max(ms1_sic))
    tgt = ms1 / ms1.max(dim=-1, keepdim=True).values
    mine = sum((((s / s.max(dim=-1, keepdim=True).values) - tgt) ** 2).mean(dim=-1) for s in (d.sum(-1), d.mean(-1), d.max(-1).values))
    assert abs(float(mine) - float(ref)) < 1e-6 * abs(float(ref))
    assert O.Diffusion.train_loss.__doc__ and "CHOSEN" in O.Diffusion.train_loss.__doc__


def _tiny(g, pred_type):
    from dquartic.model.model import DDIMDiffusionModel
    from dquartic.model.unet1d import UNet1d

    net = UNet1d(dim=4, channels=1, dim_mults=(1, 2), conditional=True, init_cond_channels=1, attn_cond_channels=1,
                 downsample_dim=8, simple=True)
    net.load_state_dict(sub(g, "w/"))
    return DDIMDiffusionModel(model_class=net.cuda(), pred_type=pred_type, device="cuda")


@pytest.mark.gpu
@pytest.mark.parametrize("pred_type,w", [("eps", 0.3), ("x0", 0.3), ("eps", 1.0)])
def test_fused_train_step_with_ms1_term_vs_oracle(golden, pred_type, w):
    from oracle import dq_oracle as O

    g = golden("tiny_diffusion.npz")
    dm = _tiny(g, pred_type)
    net = dm.model
    xb, cb, mb, t, nz = (T(g[k]) for k in ("batch/x", "batch/init_cond", "batch/attn_cond", "batch/t", "batch/noise"))
    po = {k: v.clone().requires_grad_(not k.endswith("freqs")) for k, v in sub(g, "w/").items()}
    lo, _ = O.Diffusion(po, O.UNetConfig(dim_mults=(1, 2), downsample_dim=8), pred_type=pred_type).train_loss(xb, cb, mb, t, nz, ms1_loss_weight=w)
    lo.backward()
    ref = torch.cat([po[n].grad.reshape(-1) for n, _ in net.trainable_named()])
    loss = dm.train_step_fused(xb.cuda(), cb.cuda(), mb.cuda(), t=t.cuda(), noise=nz.cuda(), ms1_loss_weight=w)
    fused = net.flat_grads().clone().cpu()
    assert abs(float(loss) - float(lo)) < 2e-5 * abs(float(lo)), (float(loss), float(lo))
    assert float((fused - ref).abs().max() / ref.abs().max()) < 1e-4
    # the generic path (autograd bridge + tensor expressions) agrees with the fused one
    net.flat_grads(zero=True)
    l2 = dm.train_step(xb.cuda(), cb.cuda(), mb.cuda(), noise=(nz.cuda() + 1) / 2, ms1_loss_weight=w, t=t.cuda())
    l2.backward()
    bridge = torch.cat([p.grad.reshape(-1) for _, p in net.trainable_named()]).cpu()
    assert abs(float(l2) - float(lo)) < 2e-5 * abs(float(lo))
    assert float((bridge - ref).abs().max() / ref.abs().max()) < 1e-4
    # and ms1_loss_weight = 0 is the plain objective, bit for bit
    a = dm.train_step_fused(xb.cuda(), cb.cuda(), mb.cuda(), t=t.cuda(), noise=nz.cuda())
    ga = net.flat_grads().clone()
    b = dm.train_step_fused(xb.cuda(), cb.cuda(), mb.cuda(), t=t.cuda(), noise=nz.cuda(), ms1_loss_weight=0.0)
    assert torch.equal(a, b) and torch.equal(ga, net.flat_grads())


@pytest.mark.gpu
def test_ms1_term_at_bench_window_size_vs_oracle_autograd():
    """the stand-alone entry point at (B, RT, MZ) = (3, 400, 64): loss and d loss / d prediction against autograd"""
    from dquartic import _native as N
    from oracle import dq_oracle as O

    gen = torch.Generator().manual_seed(3)
    B, RT, MZ, w = 3, 400, 64, 0.25
    pred = torch.randn(B, RT, MZ, generator=gen).requires_grad_()
    x_t, noise = torch.randn(B, RT, MZ, generator=gen), torch.randn(B, RT, MZ, generator=gen)
    ms1 = torch.rand(B, RT, generator=gen)
    per = ((pred - noise) ** 2).flatten(1).mean(1)
    d = x_t - pred
    m1 = O.normalize(ms1)
    tgt = m1 / m1.max(dim=-1, keepdim=True).values
    add = sum((((s / s.max(dim=-1, keepdim=True).values) - tgt) ** 2).mean(-1) for s in (d.sum(-1), d.mean(-1), d.max(-1).values))
    loss = ((1 - w) * per + w * add).mean()
    loss.backward()
    L = N.lib()
    pd, xd, nd, md = pred.detach().cuda(), x_t.cuda(), noise.cuda(), ms1.cuda()
    lo, gr, sc, sc2 = torch.empty((), device="cuda"), torch.empty_like(pd), torch.empty(1024, device="cuda"), torch.empty(5 * B * RT + B + 64, device="cuda")
    N.check(L.dq_mse_loss_fwd_bwd(N.ptr(pd), N.ptr(nd), N.ptr(lo), N.ptr(gr), N.ptr(sc), pd.numel(), N.stream_ptr()), "mse")
    N.check(L.dq_ms1_loss_fwd_bwd(N.ptr(pd), N.ptr(xd), N.ptr(md), 2.0, -1.0, None, None, w, N.ptr(lo), N.ptr(gr), N.ptr(sc2), B, RT, MZ,
                                  N.stream_ptr()), "ms1")
    torch.cuda.synchronize()
    assert abs(float(lo) - float(loss)) < 1e-5 * abs(float(loss))
    assert float((gr.cpu() - pred.grad).abs().max() / pred.grad.abs().max()) < 2e-5
