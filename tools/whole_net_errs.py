import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/diffusion-deconvolution-dia-msms-data_amd")
import numpy as np, torch
from dquartic.model.unet1d import UNet1d
g = dict(np.load("/root/repo/tests/golden/unet_default_rt16.npz"))
T = lambda a: torch.as_tensor(np.asarray(a)).float()
sub = lambda g, pre: {k[len(pre):]: T(v) for k, v in g.items() if k.startswith(pre)}
net = UNet1d(dim=4, channels=1, dim_mults=(1, 2, 2, 3, 3, 4, 4), conditional=True, init_cond_channels=1, attn_cond_channels=1, tfer_dim_mult=620, downsample_dim=64, simple=True)
net.load_state_dict(sub(g, "w/")); net = net.cuda(); net.use_rope = False
x = T(g["x"]).cuda().requires_grad_()
y = net(x, T(g["t"]).cuda(), T(g["init_cond"]).cuda(), T(g["attn_cond"]).cuda())
print("y err", float((y.detach().cpu() - T(g["norope/y"])).abs().max() / T(g["norope/y"]).abs().max()))
(y * T(g["gout"]).cuda()).sum().backward(); torch.cuda.synchronize()
ref = sub(g, "norope/grad/"); floor = 1e-4 * max(float(v.abs().max()) for v in ref.values())
named = dict(net.named_parameters()); errs = []
for k, v in ref.items():
    errs.append((float((named[k].grad.detach().cpu() - v).abs().max()) / max(float(v.abs().max()), floor), k, float(v.abs().max())))
errs.sort(reverse=True)
for e in errs[:12]: print(e)
# ---- the same gradients against the oracle evaluated in float64 (the reference's own fp32 result is ~1e-4 off it on the worst tensor)
from oracle import dq_oracle as O
_orig = O.sinusoidal_emb
O.sinusoidal_emb = lambda t, dim, theta=10000.0: _orig(t, dim, theta).double()
p = {k: v.double().clone() for k, v in sub(g, "w/").items()}
for k in p:
    if not k.endswith("freqs"): p[k].requires_grad_(True)
y64 = O.unet_forward(p, O.UNetConfig(downsample_dim=64), T(g["x"]).double(), torch.as_tensor(g["t"]), T(g["init_cond"]).double(), T(g["attn_cond"]).double(), use_rope=False)
(y64 * T(g["gout"]).double()).sum().backward()
e2 = sorted(((float((named[k].grad.detach().cpu().double() - p[k].grad).abs().max()) / max(float(p[k].grad.abs().max()), floor), k) for k in ref), reverse=True)
print("GPU vs oracle64:", e2[:8])
e3 = sorted(((float((ref[k].double() - p[k].grad).abs().max()) / max(float(p[k].grad.abs().max()), floor), k) for k in ref), reverse=True)
print("reference fp32 (golden) vs oracle64:", e3[:8])
