"""Profiling driver: B windows, `steps`-step DDIM sampling (and optionally train steps) on the default network."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffusion-deconvolution-dia-msms-data_amd"))
import torch
from dquartic.model.model import DDIMDiffusionModel
from dquartic.model.unet1d import UNet1d

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
mode = sys.argv[3] if len(sys.argv) > 3 else "sample"
torch.manual_seed(0)
net = UNet1d(dim=4, channels=1, dim_mults=(1, 2, 2, 3, 3, 4, 4), conditional=True, init_cond_channels=1, attn_cond_channels=1, downsample_dim=64, simple=True).cuda()
dm = DDIMDiffusionModel(model_class=net, device="cuda")
x = torch.randn(B, 400, 64, device="cuda"); c2 = torch.rand(B, 400, 64, device="cuda"); c1 = torch.rand(B, 400, device="cuda")
if mode == "sample":
    dm.sample(x, c2, c1, num_steps=2); torch.cuda.synchronize()
    t0 = time.time(); dm.sample(x, c2, c1, num_steps=steps); torch.cuda.synchronize(); dt = time.time() - t0
    print(f"sample B={B} steps={steps}: {dt*1e3:.2f} ms  -> {dt/steps*1e3:.3f} ms/step  {B/ (dt/steps*50):.1f} windows/s @50 steps")
else:
    dm._set_optimizer(1e-5)
    if mode == "train_graph":
        dm.enable_train_graph(True)
    for _ in range(2): dm._train_one_batch(c2, ms2_cond=c2, ms1_cond=c1, sync=False)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(steps): dm._train_one_batch(c2, ms2_cond=c2, ms1_cond=c1, sync=False)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"train B={B}: {dt/steps*1e3:.3f} ms/step  {B*steps/dt:.1f} windows/s")
