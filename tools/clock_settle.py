"""How long a fresh box takes to reach its steady train-step time: ms per step over consecutive chunks of 50 steps from a cold start."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffusion-deconvolution-dia-msms-data_amd"))
import torch
from dquartic.model.model import DDIMDiffusionModel
from dquartic.model.unet1d import UNet1d
torch.manual_seed(0)
net = UNet1d(dim=4, channels=1, dim_mults=(1, 2, 2, 3, 3, 4, 4), conditional=True, init_cond_channels=1, attn_cond_channels=1, downsample_dim=64, simple=True).cuda()
dm = DDIMDiffusionModel(model_class=net, device="cuda")
B = 32
c2 = torch.rand(B, 400, 64, device="cuda"); c1 = torch.rand(B, 400, device="cuda")
dm._set_optimizer(1e-5)
for _ in range(20): dm._train_one_batch(c2, ms2_cond=c2, ms1_cond=c1, sync=False)
torch.cuda.synchronize()
t00 = time.time(); out = []
for chunk in range(40):
    t0 = time.time()
    for _ in range(50): dm._train_one_batch(c2, ms2_cond=c2, ms1_cond=c1, sync=False)
    torch.cuda.synchronize()
    out.append(f"{(time.time()-t0)/50*1e3:.3f}")
print("ms/step per 50-step chunk after 20 warm-up steps:", " ".join(out), f"| total {time.time()-t00:.1f} s")
