"""The reference's shipped configuration (UNet1d, downsample_dim 40000, windows (34, 40000), batch 1) on one GPU: train-step and sampling-step
time.  A correctness row, timed for the record."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffusion-deconvolution-dia-msms-data_amd"))
import torch
from dquartic.model.model import DDIMDiffusionModel
from dquartic.model.unet1d import UNet1d
B, RT, MZ = int(sys.argv[1]) if len(sys.argv) > 1 else 1, 34, 40000
torch.manual_seed(0)
t0 = time.time()
net = UNet1d(dim=4, channels=1, dim_mults=(1, 2, 2, 3, 3, 4, 4), conditional=True, init_cond_channels=1, attn_cond_channels=1, downsample_dim=MZ, simple=True)
t1 = time.time()
net = net.cuda()
dm = DDIMDiffusionModel(model_class=net, device="cuda")
print(f"construct {t1 - t0:.1f} s (default init of {sum(p.numel() for p in net.parameters()) / 1e9:.2f}e9 parameters on the host), to GPU {time.time() - t1:.1f} s")
x = torch.rand(B, RT, MZ, device="cuda"); c2 = torch.rand(B, RT, MZ, device="cuda"); c1 = torch.rand(B, RT, device="cuda")
dm.sample(x, c2, c1, num_steps=1); torch.cuda.synchronize()
t0 = time.time(); dm.sample(x, c2, c1, num_steps=3); torch.cuda.synchronize(); dt = (time.time() - t0) / 3
print(f"shipped config sample step B={B}: {dt*1e3:.2f} ms/step -> {B/(dt*50):.3f} windows/s @50 steps")
dm._set_optimizer(1e-5)
for _ in range(2): dm._train_one_batch(c2, ms2_cond=c2, ms1_cond=c1, sync=False)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(5): l = dm._train_one_batch(c2, ms2_cond=c2, ms1_cond=c1, sync=False)
torch.cuda.synchronize(); dt = (time.time() - t0) / 5
print(f"shipped config train step B={B}: {dt*1e3:.2f} ms/step -> {B/dt:.2f} windows/s loss {float(l):.4f}; HBM in use {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
