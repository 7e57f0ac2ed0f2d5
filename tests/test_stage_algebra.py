"""CPU checks of the closed forms the round-4 launch fusions rest on: each restates, in torch on the CPU, what a kernel stage
computes per element and compares it with autograd of the reference formulation (unet1d.py:206-215 Upsample, :1160-1166 final_conv; model.py:349-361).
The kernels themselves are pinned on the GPU (tests/test_tiny_levels.py: against the launches they replace, bit for bit where the arithmetic is the same)."""
import torch
import torch.nn.functional as F

from oracle import dq_oracle as O


def test_upsample_conv_transpose_as_two_dense_layers():
    """TL_UP_N2 / TL_UP_N2_T (k_tiny.hip): nearest x2 of ONE position followed by a k3 conv (padding 1) is out[:, 0] = (w1 + w2) x, out[:, 1] = (w0 + w1) x;
    its transpose is d x = (w1 + w2)^T d out[:, 0] + (w0 + w1)^T d out[:, 1]."""
    torch.manual_seed(0)
    R, C = 37, 16
    w = torch.randn(C, C, 3, dtype=torch.float64)
    b = torch.randn(C, dtype=torch.float64)
    x = torch.randn(R, C, 1, dtype=torch.float64, requires_grad=True)
    y = F.conv1d(F.interpolate(x, scale_factor=2, mode="nearest"), w, b, padding=1)  # unet1d.py:206-215
    w0, w1, w2 = w[:, :, 0], w[:, :, 1], w[:, :, 2]
    assert torch.allclose(y[:, :, 0], x[:, :, 0] @ (w1 + w2).T + b, atol=1e-12)
    assert torch.allclose(y[:, :, 1], x[:, :, 0] @ (w0 + w1).T + b, atol=1e-12)
    dy = torch.randn_like(y)
    (dx,) = torch.autograd.grad(y, x, dy)
    closed = dy[:, :, 0] @ (w1 + w2) + dy[:, :, 1] @ (w0 + w1)
    assert torch.allclose(dx[:, :, 0], closed, atol=1e-12)


def test_training_head_closed_form():
    """k_level_fwd's training head: eps = final_conv(h) (1x1, 4 -> 1), loss = mean((eps - z)^2): d loss / d eps = 2 (eps - z) / N and
    d loss / d h[c] = w[c] * d loss / d eps -- what k_conv_fwd<1,1,0>, k_mse_fwd_bwd and k_conv_bwd_data<4,1,0> computed in three launches."""
    torch.manual_seed(1)
    R, n = 11, 64
    h = torch.randn(R, 4, n, dtype=torch.float64, requires_grad=True)
    w = torch.randn(1, 4, 1, dtype=torch.float64)
    b = torch.randn(1, dtype=torch.float64)
    z = torch.randn(R, 1, n, dtype=torch.float64)
    eps = F.conv1d(h, w, b)
    loss = F.mse_loss(eps, z)  # model.py:361
    (dh,) = torch.autograd.grad(loss, h)
    g = (eps - z) * (2.0 / eps.numel())
    assert torch.allclose(dh, g * w.view(1, 4, 1), atol=1e-14)
    assert torch.allclose(loss, ((eps - z) ** 2).sum() / eps.numel(), atol=1e-14)


def test_q_sample_as_the_init_stage_forms_it():
    """x_t = sqrt(ab[t]) (2 x0 - 1) + sqrt(1 - ab[t]) noise per sample (model.py:349-352), the expression of k_q_sample and of level 0's INIT stage
    in a train step, against the oracle's q_sample on the normalised x0."""
    torch.manual_seed(2)
    sched = O.make_schedule(1000, "cosine")
    ab = sched["alpha_bars"]
    B = 5
    x0 = torch.rand(B, 7, 8)
    noise = torch.randn(B, 7, 8)
    t = torch.tensor([0, 1, 500, 998, 999])
    ref = O.q_sample(ab, O.normalize(x0), t, noise)
    sa = torch.sqrt(ab[t]).view(B, 1, 1).float()
    sb = torch.sqrt(1.0 - ab[t]).view(B, 1, 1).float()
    mine = sa * (x0 * 2.0 - 1.0) + sb * noise
    assert torch.allclose(mine, ref, atol=2e-6, rtol=0)
