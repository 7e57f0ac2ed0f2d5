"""GPU parity: DDIM sampling through the drop-in DDIMDiffusionModel (native loop in libdq_hip.so) against the golden
trajectories captured from the reference (tiny network) and against the oracle at BASELINE shapes."""
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


def _tiny(g, prefix="w/"):
    from dquartic.model.model import DDIMDiffusionModel
    from dquartic.model.unet1d import UNet1d

    net = UNet1d(dim=4, channels=1, dim_mults=(1, 2), conditional=True, init_cond_channels=1, attn_cond_channels=1,
                 downsample_dim=8, simple=True)
    net.load_state_dict(sub(g, prefix))
    net = net.cuda()
    return DDIMDiffusionModel(model_class=net, num_timesteps=1000, beta_schedule_type="cosine", pred_type="eps",
                              auto_normalize=True, ms1_loss_weight=0.0, device="cuda")


def test_schedule_matches_golden(golden):
    g = golden("tiny_diffusion.npz")
    s = golden("schedule.npz")
    dm = _tiny(g)
    assert np.array_equal(dm.alpha_bars.cpu().numpy(), s["cosine/alpha_bars"])
    assert np.array_equal(dm.betas.cpu().numpy(), s["cosine/betas"])
    assert dm.sampler_timesteps(1000, 50).tolist() == s["timesteps50"].tolist()


def test_q_sample_and_p_sample_golden(golden):
    g = golden("tiny_diffusion.npz")
    dm = _tiny(g)
    x0, c2, c1 = (T(g[k]).cuda() for k in ("x0", "ms2_cond", "ms1_cond"))
    xt = dm.q_sample(dm.normalize(x0), T(g["q/t"]).cuda(), T(g["q/noise"]).cuda())
    assert rel_err(xt, g["q/x_t"]) < 1e-6
    dm.model.eval()
    with torch.no_grad():
        for tv in (999, 500, 1, 0):
            xp, ep = dm.p_sample(T(g["p/x_t"]).cuda(), tv, dm.normalize(c2), dm.normalize(c1))
            assert rel_err(ep, g[f"p/{tv}/eps"]) < 1e-4      # fp32 tolerance stated by north_star: 1e-4 rel on eps per step
            assert rel_err(xp, g[f"p/{tv}/x_prev"]) < 1e-4


@pytest.mark.parametrize("ns", [5, 50])
def test_sample_trajectory_golden(golden, ns):
    g = golden("tiny_diffusion.npz")
    dm = _tiny(g)
    x_T, c2, c1 = (T(g[k]).cuda() for k in ("p/x_t", "ms2_cond", "ms1_cond"))
    s, pn, tx, te = dm.sample(x_T, c2, c1, num_steps=ns, return_trajectory=True)
    # per-step eps: 1e-4 relative, the tolerance north_star states (first step amplifies eps error ~31.6x into x, SURVEY 3.2,
    # hence 5e-4 on x)
    assert rel_err(te, g[f"s{ns}/traj_eps"]) < 1e-4
    assert rel_err(tx, g[f"s{ns}/traj_x"]) < 5e-4
    assert rel_err(s, g[f"s{ns}/sample"]) < 5e-4
    assert rel_err(pn, g[f"s{ns}/pred_noise"]) < 5e-4
    mse = float(((s.cpu() - T(g[f"s{ns}/sample"])) ** 2).mean())
    assert mse < 1e-8  # denoised-MS2 MSE vs the reference
    # x_T untouched; second output is mixture - denoised (model.py:321-322)
    assert torch.equal(x_T.cpu(), T(g["p/x_t"]))
    assert rel_err(pn, (c2 - s).cpu()) < 1e-6


def test_predict_one_batch_golden(golden):
    g = golden("harness.npz")
    dm = _tiny(g, "pred/w/")
    x0, c2, c1 = (T(g[k]).cuda() for k in ("pred/x0", "pred/ms2_cond", "pred/ms1_cond"))
    # the reference draws x_T = randn_like(x_0) on ITS device (CPU); feed the captured draw instead of re-drawing on the GPU
    dm.model.eval()
    with torch.no_grad():
        s, pn = dm.sample(T(g["pred/x_T"]).cuda(), ms2_cond=c2, ms1_cond=c1, num_steps=5)
    assert rel_err(s[0], g["pred/sample0"]) < 5e-4 and rel_err(pn[0], g["pred/pred_noise0"]) < 5e-4
    a, b = dm._predict_one_batch(x0, ms2_cond=c2, ms1_cond=c1, num_steps=5)
    assert isinstance(a, np.ndarray) and a.shape == tuple(x0.shape[1:]) and b.shape == a.shape


def test_sample_full_size_vs_oracle(golden, la_form):
    """BASELINE shape (RT=400, MZ=64), default network, B=2, 5 steps: per-step eps and final sample vs the oracle"""
    from dquartic.model.model import DDIMDiffusionModel
    from dquartic.model.unet1d import UNet1d
    from oracle import dq_oracle as O

    g = golden("unet_default_rt16.npz")
    p = sub(g, "w/")
    net = UNet1d(dim=4, channels=1, dim_mults=(1, 2, 2, 3, 3, 4, 4), conditional=True, init_cond_channels=1,
                 attn_cond_channels=1, downsample_dim=64, simple=True)
    net.load_state_dict(p)
    dm = DDIMDiffusionModel(model_class=net.cuda(), device="cuda")
    gen = torch.Generator().manual_seed(9)
    B, RT, MZ = 2, 400, 64
    xT, c2, c1 = torch.randn(B, RT, MZ, generator=gen), torch.rand(B, RT, MZ, generator=gen), torch.rand(B, RT, generator=gen)
    tr = []
    with torch.no_grad():
        so, po = O.Diffusion(p, O.UNetConfig(downsample_dim=64)).sample(xT, c2, c1, 5, trace=tr)
    s, pn, tx, te = dm.sample(xT.cuda(), c2.cuda(), c1.cuda(), num_steps=5, return_trajectory=True)
    assert rel_err(te, torch.stack([e for _, _, e in tr])) < 1e-4
    assert rel_err(s, so) < 1e-3
    assert float(((s.cpu() - so) ** 2).mean()) < 1e-8


def test_graph_replay_equals_eager_loop(golden):
    """sample() replays one hipGraph-captured step; it must equal the eager per-step loop bit for bit, also on a second call
    with different inputs (same captured graph, conditions re-staged) and for another step count"""
    g = golden("tiny_diffusion.npz")
    dm = _tiny(g)
    x_T, c2, c1 = (T(g[k]).cuda() for k in ("p/x_t", "ms2_cond", "ms1_cond"))
    assert dm.use_graph
    for ns, xx, cc in ((50, x_T, c2), (7, x_T * 0.5, 1 - c2), (50, x_T, c2)):
        dm.use_graph = True
        sg, ng = dm.sample(xx, cc, c1, num_steps=ns)
        dm.use_graph = False
        se, ne = dm.sample(xx, cc, c1, num_steps=ns)
        assert torch.equal(sg, se) and torch.equal(ng, ne)
    assert rel_err(sg, g["s50/sample"]) < 5e-4
