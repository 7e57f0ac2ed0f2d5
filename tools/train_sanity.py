import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/diffusion-deconvolution-dia-msms-data_amd")
import bench
dev = torch.device("cuda", 0)
net, dm = bench.build_model(dev)
dm._set_optimizer(1e-3)
batches = bench.make_batches(4, 32, 0, 1, dev)
losses = []
for i in range(120):
    x0, c2, c1 = batches[i % 4]
    losses.append(float(dm._train_one_batch(x0, ms2_cond=c2, ms1_cond=c1, sync=True)))
print("unet loss first10 %.4f last10 %.4f finite %s" % (sum(losses[:10]) / 10, sum(losses[-10:]) / 10, all(l == l for l in losses)))
from dquartic.model.building_blocks import CustomTransformer, DDIMTransformerAdapter
from dquartic.model.model import DDIMDiffusionModel
torch.manual_seed(0)
tn = DDIMTransformerAdapter(CustomTransformer(256, 128, 4, 2)).to(dev)
dt = DDIMDiffusionModel(model_class=tn, num_timesteps=1000, beta_schedule_type="cosine", pred_type="eps", auto_normalize=True, ms1_loss_weight=0.0, device=dev)
dt._set_optimizer(1e-3)
x0, c2, c1 = torch.rand(16, 34, 256, device=dev), torch.rand(16, 34, 256, device=dev), torch.rand(16, 34, device=dev)
ls = [float(dt._train_one_batch(x0, ms2_cond=c2, ms1_cond=c1, sync=True)) for _ in range(200)]
print("tfm loss first10 %.4f last10 %.4f finite %s" % (sum(ls[:10]) / 10, sum(ls[-10:]) / 10, all(l == l for l in ls)))
