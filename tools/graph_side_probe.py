"""Batch-1 / batch-4 / batch-32 train step three ways: eager (two queues), one captured hipGraph as a single chain, one captured hipGraph with the
side-stream fork / join inside (two branches).  usage: python tools/graph_side_probe.py"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffusion-deconvolution-dia-msms-data_amd"))
import torch
from dquartic.model.model import DDIMDiffusionModel
from dquartic.model.unet1d import UNet1d
torch.manual_seed(0)
net = UNet1d(dim=4, channels=1, dim_mults=(1, 2, 2, 3, 3, 4, 4), conditional=True, init_cond_channels=1, attn_cond_channels=1, downsample_dim=64, simple=True).cuda()
dm = DDIMDiffusionModel(model_class=net, device="cuda")
dm._prepare_training(1e-4)
def timed(n=100):
    for _ in range(5): dm._train_one_batch(x0, ms2_cond=c2, ms1_cond=c1, sync=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): dm._train_one_batch(x0, ms2_cond=c2, ms1_cond=c1, sync=False)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for B in (1, 4, 32):
    x0, c2, c1 = torch.rand(B, 400, 64).cuda(), torch.rand(B, 400, 64).cuda(), torch.rand(B, 400).cuda()
    dm.enable_train_graph(False); e = timed()
    dm.enable_train_graph(True, keep_side=False); g0 = timed()
    dm.enable_train_graph(True, keep_side=True); g1 = timed()
    dm.enable_train_graph(False)
    print("batch %2d: eager %.3f ms, graph (one chain) %.3f ms, graph (fork / join inside) %.3f ms" % (B, e, g0, g1), flush=True)
