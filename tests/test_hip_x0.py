"""GPU parity, pred_type="x0" (reference model.py:209-210, 274-278, 372-376, 404): sampling and the
train step through the drop-in DDIMDiffusionModel against goldens captured from the reference (tiny network, B = 1 and
the B = 3 per-sample loop) and against the oracle at the default network's shapes."""
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
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _tiny(g):
    from dquartic.model.model import DDIMDiffusionModel
    from dquartic.model.unet1d import UNet1d

    net = UNet1d(dim=4, channels=1, dim_mults=(1, 2), conditional=True, init_cond_channels=1, attn_cond_channels=1,
                 downsample_dim=8, simple=True)
    net.load_state_dict(sub(g, "w/"))
    return DDIMDiffusionModel(model_class=net.cuda(), num_timesteps=1000, beta_schedule_type="cosine", pred_type="x0",
                              auto_normalize=True, ms1_loss_weight=0.0, device="cuda")


def test_loss_weight_and_p_sample_golden(golden):
    g = golden("tiny_x0.npz")
    dm = _tiny(g)
    assert np.array_equal(dm.loss_weight.cpu().numpy(), g["loss_weight"])  # SNR table bit-exact (formed on the host)
    c2, c1 = (T(g[k]).cuda() for k in ("ms2_cond", "ms1_cond"))
    dm.model.eval()
    with torch.no_grad():
        for tv in (999, 500, 1, 0):
            xp, ep = dm.p_sample(T(g["p/x_t"]).cuda(), tv, dm.normalize(c2), dm.normalize(c1))
            assert rel_err(ep, g[f"p/{tv}/eps"]) < 1e-4
            assert rel_err(xp, g[f"p/{tv}/x_prev"]) < 1e-4


@pytest.mark.parametrize("graph", [False, True])
def test_sample_golden(golden, graph):
    g = golden("tiny_x0.npz")
    dm = _tiny(g)
    dm.use_graph = graph
    x_T, c2, c1 = (T(g[k]).cuda() for k in ("p/x_t", "ms2_cond", "ms1_cond"))
    if not graph:
        s, pn, tx, te = dm.sample(x_T, c2, c1, num_steps=5, return_trajectory=True)
        assert rel_err(te, g["s5/traj_eps"]) < 2e-4  # trajectory holds eps_pred derived from the x0 prediction
        assert rel_err(tx, g["s5/traj_x"]) < 5e-4
    else:
        s, pn = dm.sample(x_T, c2, c1, num_steps=5)
    assert rel_err(s, g["s5/sample"]) < 5e-4
    assert rel_err(pn, g["s5/pred_noise"]) < 5e-4
    assert torch.equal(x_T.cpu(), T(g["p/x_t"]))


def test_train_step_golden_fused_and_autograd(golden):
    g = golden("tiny_x0.npz")
    dm = _tiny(g)
    net = dm.model
    net.train()
    x0, c2, c1 = (T(g[k]).cuda() for k in ("x0", "ms2_cond", "ms1_cond"))
    t, nz = T(g["train/t"]).cuda(), T(g["train/noise"]).cuda()
    keys = [k for k, _ in net.named_parameters() if not k.endswith("rotary_emb.freqs")]
    gmax = max(float(np.abs(g["train/grad/" + k]).max()) for k in keys)

    def check_grads(tag):
        for k, p in net.named_parameters():
            if k.endswith("rotary_emb.freqs"):
                continue
            ref = T(g["train/grad/" + k])
            err = float((p.grad.cpu() - ref).abs().max())
            assert err <= 1e-3 * max(float(ref.abs().max()), 1e-4 * gmax), (tag, k, err)

    # fused native step
    loss = dm.train_step_fused(x0, c2, c1, t=t, noise=nz, zero_grads=True)
    assert rel_err(loss.reshape(1), g["train/loss"]) < 2e-5
    check_grads("fused")
    # autograd bridge (train_step maps a passed noise 2n-1 like the reference: feed (n+1)/2)
    net.flat_grads(zero=True)
    loss2 = dm.train_step(x0, c2, c1, noise=(nz + 1) / 2, t=t)
    loss2.backward()
    assert rel_err(loss2.reshape(1), g["train/loss"]) < 2e-5
    check_grads("autograd")
    # B = 3: mean over samples of the reference's B = 1 weighted losses (t = 30 has SNR ~ 2e3)
    lb = dm.train_step_fused(T(g["batch/x"]).cuda(), T(g["batch/init_cond"]).cuda(), T(g["batch/attn_cond"]).cuda(),
                             t=T(g["batch/t"]).cuda(), noise=T(g["batch/noise"]).cuda())
    assert rel_err(lb.reshape(1), np.asarray(g["batch/loss_mean"]).reshape(1)) < 2e-5


def test_default_net_vs_oracle():
    """Default 7-level network at RT = 24, B = 3: weighted x0 loss, all parameter gradients and a 3-step sample vs the oracle."""
    from oracle import dq_oracle as O
    from dquartic.model.model import DDIMDiffusionModel
    from dquartic.model.unet1d import UNet1d

    torch.manual_seed(5)
    net = UNet1d(dim=4, channels=1, dim_mults=(1, 2, 2, 3, 3, 4, 4), conditional=True, init_cond_channels=1,
                 attn_cond_channels=1, downsample_dim=64, simple=True)
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.05 * torch.randn_like(p))
    po = {k: v.detach().clone().cpu() for k, v in net.state_dict().items()}
    dm = DDIMDiffusionModel(model_class=net.cuda(), pred_type="x0", device="cuda")
    od = O.Diffusion(po, O.UNetConfig(downsample_dim=64), pred_type="x0")
    B, RT, MZ = 3, 24, 64
    gen = torch.Generator().manual_seed(1)
    x0, c2, c1 = torch.rand(B, RT, MZ, generator=gen), torch.rand(B, RT, MZ, generator=gen), torch.rand(B, RT, generator=gen)
    t, nz = torch.tensor([7, 500, 950]), torch.randn(B, RT, MZ, generator=gen)
    keys = O.trainable_keys(po)
    for k in keys:
        po[k].requires_grad_(True)
    lo, _ = od.train_loss(x0, c2, c1, t, nz)
    lo.backward()
    net.train()
    loss = dm.train_step_fused(x0.cuda(), c2.cuda(), c1.cuda(), t=t.cuda(), noise=nz.cuda())
    assert rel_err(loss.reshape(1), lo.detach().reshape(1)) < 5e-5
    gmax = max(float(po[k].grad.abs().max()) for k in keys)
    named = dict(net.named_parameters())
    for k in keys:
        ref = po[k].grad
        err = float((named[k].grad.cpu() - ref).abs().max())
        assert err <= 1e-3 * max(float(ref.abs().max()), 1e-4 * gmax), (k, err)
    with torch.no_grad():
        so, no = od.sample(nz, c2, c1, 3)
    s, n = dm.sample(nz.cuda(), c2.cuda(), c1.cuda(), num_steps=3)
    assert rel_err(s, so) < 5e-4 and rel_err(n, no) < 5e-4


def test_unknown_pred_type_rejected():
    from dquartic import _native as N
    from dquartic.model.model import DDIMDiffusionModel
    from dquartic.model.unet1d import UNet1d

    net = UNet1d(dim=4, channels=1, dim_mults=(1, 2), conditional=True, init_cond_channels=1, attn_cond_channels=1,
                 downsample_dim=8, simple=True)
    with pytest.raises(ValueError, match="Unknown pred_type"):
        DDIMDiffusionModel(model_class=net, pred_type="v", device="cuda")
    # the C ABI rejects an unknown enum value too (no silent default)
    rc = N.lib().dq_ddim_sample(None, None, None, None, 1000, None, None, None, 1, 7, None, 1, None, None, None, None, 0, None, 0, 1, 1, None)
    assert rc != 0
    assert isinstance(N.lib().dq_last_error(), (bytes, type(None)))
    assert ctypes.c_int(N.lib().dq_abi_version()).value == N.ABI_VERSION
