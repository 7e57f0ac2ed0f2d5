#!/usr/bin/env python3
"""In-situ interval probe of ONE kernel instantiation inside the train step (dq_probe.h).
usage: DQ_HIP_LIB=<variant built with -DDQ_KPROBE> tools/probe_step.py <id> [batch]
  id: k_level_fwd<C,PRE,CP>: C*1000 + PRE*100 + CP;  k_res_bwd_wg<C,WR>: 100000 + C*10 + WR;  k_conv_bwd_wg<C,PRE,CP>: 200000 + C*1000 + PRE*100 + CP
Prints the median / max shader-clock interval between consecutive stamps over the workgroups of the last launch(es) of that instantiation."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "diffusion-deconvolution-dia-msms-data_amd"))
from dquartic import _native as N  # noqa: E402
from dquartic.model.model import DDIMDiffusionModel  # noqa: E402
from dquartic.model.unet1d import UNet1d  # noqa: E402

ident = int(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
torch.manual_seed(0)
net = UNet1d(dim=4, channels=1, dim_mults=(1, 2, 2, 3, 3, 4, 4), conditional=True, init_cond_channels=1, attn_cond_channels=1, downsample_dim=64, simple=True).cuda()
dm = DDIMDiffusionModel(model_class=net, device="cuda")
c2, c1 = torch.rand(B, 400, 64, device="cuda"), torch.rand(B, 400, device="cuda")
dm._set_optimizer(1e-5)
step = lambda: dm._train_one_batch(c2, ms2_cond=c2, ms1_cond=c1, sync=False)
lib = ctypes.CDLL(N.LIB_PATH)
for _ in range(3):
    step()
torch.cuda.synchronize()
assert lib.dq_kprobe_clear() == 0 and lib.dq_kprobe_select(ident) == 0
step()
torch.cuda.synchronize()
buf = np.zeros(4096 * 16, dtype=np.uint64)
assert lib.dq_kprobe_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
st = buf.reshape(4096, 16).astype(np.int64)
st = st[st[:, 0] > 0]
print(f"id {ident}: {len(st)} workgroups")
idx = [i for i in range(16) if (st[:, i] > 0).all()]
med = {i: int(np.median(st[:, i] - st[:, 0])) for i in idx}
idx.sort(key=lambda i: med[i] if i > 1 else i - 2)  # (stamps 0, 1: prologue; the rest in time order of the last tile)
for a_, b_ in zip(idx, idx[1:]):
    d = st[:, b_] - st[:, a_]
    print(f"  stamp {a_:2d} -> {b_:2d}: {int(np.median(d)):8d} {int(d.max()):8d}")
print(f"  first -> last stamp: {int(np.median(st[:, idx[-1]] - st[:, idx[0]]))} clocks (median);  first start -> last end {int(st[:, idx[-1]].max() - st[:, idx[0]].min())}")
