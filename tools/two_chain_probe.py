#!/usr/bin/env python3
"""Experiment: the batch-32 train step as TWO half-batch chains on two streams (each its own plan, workspace and gradient buffer over the
SAME parameters), gradients added afterwards -- does one chain's latency-bound deep levels hide under the other's wide levels?
usage: python tools/two_chain_probe.py [B] [steps]"""
import ctypes, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffusion-deconvolution-dia-msms-data_amd"))
import torch
from dquartic import _native as N
from dquartic.model.model import DDIMDiffusionModel
from dquartic.model.unet1d import UNet1d

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
NCH = int(os.environ.get("CHAINS", "2"))
torch.manual_seed(0)
net = UNet1d(dim=4, channels=1, dim_mults=(1, 2, 2, 3, 3, 4, 4), conditional=True, init_cond_channels=1, attn_cond_channels=1, downsample_dim=64, simple=True).cuda()
dm = DDIMDiffusionModel(model_class=net, device="cuda")
dm._set_optimizer(1e-5)
L = N.lib()
RT, MZ = 400, 64
x0 = torch.rand(B, RT, MZ, device="cuda"); c2 = torch.rand(B, RT, MZ, device="cuda"); c1 = torch.rand(B, RT, device="cuda")
hb = B // NCH
mults = (ctypes.c_int * len(net.dim_mults))(*net.dim_mults)
plans = [L.dq_plan_create(net.dim, len(net.dim_mults), mults, net.downsample_dim, 1000) for _ in range(NCH)]
wss = [torch.empty(L.dq_unet_workspace_bytes(plans[i], hb, RT, 1), dtype=torch.uint8, device="cuda") for i in range(NCH)]
flat = net.flat_params
gbuf = [torch.zeros_like(flat) for _ in range(NCH)]
losses = [torch.empty((), device="cuda") for _ in range(NCH)]
streams = [torch.cuda.Stream() for _ in range(NCH)]
ab = dm.alpha_bars.to("cuda"); lw = dm.loss_weight.to(device="cuda", dtype=torch.float32).contiguous()
fr = net.rope_freqs()

def step():
    t = torch.randint(0, dm.num_timesteps, (B,), device="cuda").long()
    noise = torch.randn_like(x0)
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(main)
    for i in range(NCH):
        s = streams[i]
        s.wait_event(ev)
        sl = slice(i * hb, (i + 1) * hb)
        with torch.cuda.stream(s):
            gbuf[i].zero_()
            N.check(L.dq_train_step(plans[i], N.ptr(flat), N.ptr(fr), N.ptr(ab), N.ptr(x0[sl]), N.ptr(c2[sl]), N.ptr(c1[sl]), N.ptr(t[sl]),
                                    N.ptr(noise[sl]), 1, N.PRED_TYPES[dm.pred_type], N.ptr(lw), 0.0, N.ptr(gbuf[i]), N.ptr(losses[i]), N.ptr(wss[i]),
                                    wss[i].numel(), hb, RT, s.cuda_stream), "dq_train_step")
    for s in streams:
        main.wait_stream(s)
    g = net.flat_grads(zero=False)
    torch.add(gbuf[0], gbuf[1], out=g) if NCH == 2 else g.copy_(gbuf[0])
    dm.optimizer.grad_scale = 1.0 / NCH
    dm.optimizer.step()

for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(steps): step()
torch.cuda.synchronize(); dt = time.time() - t0
print(f"{NCH} chain(s) of {hb}: {dt / steps * 1e3:.3f} ms/step  {B * steps / dt:.1f} windows/s")
