"""CPU: the oracle (oracle/dq_oracle.py) against the golden vectors captured from the reference itself
(oracle/make_golden.py).  This is what pins the oracle; the GPU parity tests then compare the HIP path
with the oracle."""
import math

import numpy as np
import pytest
import torch

from conftest import sub
from oracle import dq_oracle as O

T = torch.from_numpy


def close(a, b, rtol=2e-5, atol=2e-6):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    scale = max(1.0, float(b.abs().max()))
    assert a.shape == b.shape, (a.shape, b.shape)
    err = float((a.detach() - b).abs().max())
    assert err <= atol * scale + rtol * scale, f"max abs err {err:.3e} (scale {scale:.3g})"


def test_schedule_bit_exact(golden):
    g = golden("schedule.npz")
    for kind in ("cosine", "linear"):
        s = O.make_schedule(1000, kind)
        for k in ("betas", "alphas", "alpha_bars"):
            assert np.array_equal(s[k].numpy(), g[f"{kind}/{k}"]), (kind, k)
    assert O.sampler_timesteps(1000, 50) == g["timesteps50"].tolist()
    assert O.sampler_timesteps(1000, 5) == g["timesteps5"].tolist()
    # known answers quoted in SURVEY 8a / Appendix A
    ab = O.make_schedule(1000, "cosine")["alpha_bars"]
    assert abs(float(ab[0]) - 0.999958694) < 1e-7 and abs(float(ab[500]) - 0.492285043) < 1e-7
    assert O.sampler_timesteps(1000, 50)[:4] == [999, 978, 958, 937] and O.sampler_timesteps(1000, 50)[-3:] == [40, 20, 0]


@pytest.mark.parametrize("C", [4, 12])
def test_rmsnorm(golden, C):
    g = golden("blocks.npz")
    close(O.rmsnorm(T(g[f"rmsnorm{C}/x"]), T(g[f"rmsnorm{C}/g"])), g[f"rmsnorm{C}/y"])


def test_time_mlp(golden):
    g = golden("blocks.npz")
    t = T(g["time/t"])
    close(O.sinusoidal_emb(t, 4), g["time/sinu"])
    p = {"time_mlp." + k: v for k, v in sub(g, "time/").items() if k[0].isdigit()}
    close(O.time_mlp(p, t, O.UNetConfig()), g["time/out"])


def test_cond_scale_shift(golden):
    g = golden("blocks.npz")
    w = sub(g, "css/")
    ss = torch.nn.functional.linear(torch.nn.functional.silu(w["temb"]), w["to_scale_shift.1.weight"], w["to_scale_shift.1.bias"])
    close(w["x"] * (ss[:, :1] + 1) + ss[:, 1:], w["y"])


@pytest.mark.parametrize("name", ["res_4_4_64", "res_24_12_4", "res_32_16_1", "res_8_4_64"])
def test_resnet_block(golden, name):
    g = golden("blocks.npz")
    p = {"b." + k: v for k, v in sub(g, f"{name}/w/").items()}
    x, temb = T(g[f"{name}/x"]), T(g[f"{name}/temb"])
    close(O.resnet_block(p, "b", x, temb, x.shape[0]), g[f"{name}/y"])


@pytest.mark.parametrize("C,n", [(4, 64), (4, 32), (8, 16), (12, 4), (12, 2), (16, 1)])
def test_linear_attention(golden, C, n):
    g = golden("blocks.npz")
    p = {"la." + k: v for k, v in sub(g, f"la_{C}_{n}/w/").items()}
    close(O.linear_attention(p, "la", T(g[f"la_{C}_{n}/x"])), g[f"la_{C}_{n}/y"])


def test_down_up(golden):
    g = golden("blocks.npz")
    F = torch.nn.functional
    close(F.conv1d(T(g["down/x"]), T(g["down/weight"]), T(g["down/bias"]), stride=2, padding=1), g["down/y"])
    y = F.conv1d(F.interpolate(T(g["up/x"]), scale_factor=2, mode="nearest"), T(g["up/weight"]), T(g["up/bias"]), padding=1)
    close(y, g["up/y"])


@pytest.mark.parametrize("tag,use_rope", [("rope", True), ("norope", False)])
def test_whole_net_forward_and_grads(golden, tag, use_rope):
    g = golden("unet_default_rt16.npz")
    p = {k: v.clone().requires_grad_(not k.endswith("freqs")) for k, v in sub(g, "w/").items()}
    cfg = O.UNetConfig(downsample_dim=64)
    x, c2, c1 = (T(g[k]).clone().requires_grad_(True) for k in ("x", "init_cond", "attn_cond"))
    y = O.unet_forward(p, cfg, x, T(g["t"]), c2, c1, use_rope=use_rope)
    close(y, g[f"{tag}/y"])
    (y * T(g["gout"])).sum().backward()
    close(x.grad, g[f"{tag}/dx"], rtol=1e-4)
    close(c2.grad, g[f"{tag}/dinit_cond"], rtol=1e-4)
    close(c1.grad, g[f"{tag}/dattn_cond"], rtol=1e-4)
    n = 0
    for k, v in sub(g, f"{tag}/grad/").items():
        close(p[k].grad, v, rtol=1e-4)
        n += 1
    assert n == 395  # trainable tensors
    assert sum(v.numel() for k, v in p.items() if not k.endswith("freqs")) == 128847


def _tiny(golden):
    g = golden("tiny_diffusion.npz")
    p = sub(g, "w/")
    cfg = O.UNetConfig(dim_mults=(1, 2), downsample_dim=8)
    return g, O.Diffusion(p, cfg)


def test_q_sample_and_p_sample(golden):
    g, d = _tiny(golden)
    x0, c2, c1 = T(g["x0"]), T(g["ms2_cond"]), T(g["ms1_cond"])
    close(O.q_sample(d.alpha_bars, O.normalize(x0), T(g["q/t"]), T(g["q/noise"])), g["q/x_t"])
    with torch.no_grad():
        for tv in (999, 500, 1, 0):
            xp, ep = d.p_sample(T(g["p/x_t"]), tv, O.normalize(c2), O.normalize(c1))
            close(ep, g[f"p/{tv}/eps"], rtol=1e-4)
            close(xp, g[f"p/{tv}/x_prev"], rtol=1e-4)
        # t = 0 returns x0_pred itself
        xp0, ep0 = d.p_sample(T(g["p/x_t"]), 0, O.normalize(c2), O.normalize(c1))
        ab0 = d.alpha_bars[0]
        close(xp0, (T(g["p/x_t"]) - torch.sqrt(1 - ab0) * ep0) / torch.sqrt(ab0))


@pytest.mark.parametrize("ns", [5, 50])
def test_sample_trajectory(golden, ns):
    g, d = _tiny(golden)
    tr = []
    with torch.no_grad():
        s, pn = d.sample(T(g["p/x_t"]), T(g["ms2_cond"]), T(g["ms1_cond"]), ns, trace=tr)
    # the first step multiplies eps error by ~31.6: tolerance is relative to the trajectory scale
    close(torch.stack([e for _, _, e in tr]), g[f"s{ns}/traj_eps"], rtol=3e-4)
    close(torch.stack([x for _, x, _ in tr]), g[f"s{ns}/traj_x"], rtol=3e-4)
    close(s, g[f"s{ns}/sample"], rtol=3e-4)
    close(pn, g[f"s{ns}/pred_noise"], rtol=3e-4)
    # second output is mixture - denoised (model.py:321-322)
    close(pn, T(g["ms2_cond"]) - s, rtol=1e-6)


def test_train_loss_and_batched_semantics(golden):
    g, d = _tiny(golden)
    with torch.no_grad():
        loss, _ = d.train_loss(T(g["x0"]), T(g["ms2_cond"]), T(g["ms1_cond"]), T(g["train/t"]), T(g["train/noise"]))
        close(loss.reshape(1), g["train/loss"], rtol=1e-5)
        # B > 1 == per-sample loop over the B = 1 reference
        y = d.net(T(g["batch/x"]), T(g["batch/t"]), T(g["batch/init_cond"]), T(g["batch/attn_cond"]))
        close(y, g["batch/y"])
        lb, _ = d.train_loss(T(g["batch/x"]), T(g["batch/init_cond"]), T(g["batch/attn_cond"]), T(g["batch/t"]), T(g["batch/noise"]))
        close(lb, g["batch/loss_mean"], rtol=1e-5)


def test_optimizer_steps(golden):
    """zero_grad -> train_step -> backward -> clip 10 -> AdamW (model_interface.py:1112-1123), 3 steps."""
    g, d = _tiny(golden)
    p = {k: v.clone() for k, v in d.params.items()}
    keys = O.trainable_keys(p)
    m = {k: torch.zeros_like(p[k]) for k in keys}
    v = {k: torch.zeros_like(p[k]) for k in keys}
    lr = float(g["opt/lr"])
    x0, c2, c1 = T(g["x0"]), T(g["ms2_cond"]), T(g["ms1_cond"])
    for step in range(3):
        for k in keys:
            p[k].requires_grad_(True)
            p[k].grad = None
        dd = O.Diffusion(p, d.cfg)
        loss, _ = dd.train_loss(x0, c2, c1, T(g["opt/t"])[step:step + 1], T(g["opt/noise"])[step:step + 1])
        loss.backward()
        grads = [p[k].grad for k in keys]
        gn, coef = O.clip_coef(grads)
        assert abs(float(loss) - g["opt/losses"][step]) <= 2e-5 * max(1, abs(g["opt/losses"][step]))
        assert abs(gn - g["opt/gnorms"][step]) <= 2e-4 * g["opt/gnorms"][step]
        with torch.no_grad():
            for k in keys:
                p[k].requires_grad_(False)
                O.adamw_step(p[k], grads[keys.index(k)] * coef, m[k], v[k], step + 1, lr)
        if step in (0, 2):
            for k in keys:
                ref = T(g[f"opt/after{step + 1}/{k}"])
                # one AdamW step moves a weight by ~lr; compare the displacement, not just the value
                assert float((p[k] - ref).abs().max()) <= 2e-7 + 0.05 * lr, k


def test_harness_contract(golden):
    g = golden("harness.npz")
    a, b = T(g["in/ms2_1"]), T(g["in/ms2_2"])
    close(0.5 * a + 0.5 * b, g["out/ms2_cond"], rtol=0, atol=0)
    assert np.array_equal(g["out/x_0"], g["in/ms2_1"]) and np.array_equal(g["out/ms1_cond"], g["in/ms1_1"])
    p = sub(g, "pred/w/")
    d = O.Diffusion(p, O.UNetConfig(dim_mults=(1, 2), downsample_dim=8))
    with torch.no_grad():
        s, pn = d.sample(T(g["pred/x_T"]), T(g["pred/ms2_cond"]), T(g["pred/ms1_cond"]), 5)
    close(s[0], g["pred/sample0"], rtol=3e-4)
    close(pn[0], g["pred/pred_noise0"], rtol=3e-4)


# ---------------------------------------------------------------------------------------------------------------
# pred_type = "x0" (model.py:209-210, 274-278, 372-376): SURVEY 8(f) row 2
# ---------------------------------------------------------------------------------------------------------------
def _tiny_x0(golden):
    g = golden("tiny_x0.npz")
    return g, O.Diffusion(sub(g, "w/"), O.UNetConfig(dim_mults=(1, 2), downsample_dim=8), pred_type="x0")


def test_x0_loss_weight_and_p_sample(golden):
    g, d = _tiny_x0(golden)
    assert np.array_equal(d.loss_weight.numpy(), g["loss_weight"])  # SNR table, bit-exact
    c2, c1 = O.normalize(T(g["ms2_cond"])), O.normalize(T(g["ms1_cond"]))
    with torch.no_grad():
        for tv in (999, 500, 1, 0):
            xp, ep = d.p_sample(T(g["p/x_t"]), tv, c2, c1)
            close(ep, g[f"p/{tv}/eps"], rtol=1e-4)
            close(xp, g[f"p/{tv}/x_prev"], rtol=1e-4)
        tr = []
        s, pn = d.sample(T(g["p/x_t"]), T(g["ms2_cond"]), T(g["ms1_cond"]), 5, trace=tr)
    close(torch.stack([e for _, _, e in tr]), g["s5/traj_eps"], rtol=3e-4)
    close(torch.stack([x for _, x, _ in tr]), g["s5/traj_x"], rtol=3e-4)
    close(s, g["s5/sample"], rtol=3e-4)
    close(pn, g["s5/pred_noise"], rtol=3e-4)


def test_x0_train_loss_grads_and_batch(golden):
    g, d = _tiny_x0(golden)
    keys = O.trainable_keys(d.params)
    for k in keys:
        d.params[k].requires_grad_(True)
    loss, _ = d.train_loss(T(g["x0"]), T(g["ms2_cond"]), T(g["ms1_cond"]), T(g["train/t"]), T(g["train/noise"]))
    close(loss.reshape(1), g["train/loss"], rtol=1e-5)
    loss.backward()
    gmax = max(float(np.abs(g["train/grad/" + k]).max()) for k in keys)
    for k in keys:
        ref = T(g["train/grad/" + k])
        err = float((d.params[k].grad - ref).abs().max())
        assert err <= 2e-4 * max(float(ref.abs().max()), 1e-4 * gmax), (k, err)
    with torch.no_grad():
        lb, _ = d.train_loss(T(g["batch/x"]), T(g["batch/init_cond"]), T(g["batch/attn_cond"]), T(g["batch/t"]), T(g["batch/noise"]))
    close(lb, g["batch/loss_mean"], rtol=1e-5)
    with pytest.raises(ValueError):
        O.Diffusion(d.params, d.cfg, pred_type="v")
