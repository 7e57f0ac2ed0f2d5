#!/usr/bin/env python3
"""Interval probe of one k_linattn_fwd instantiation alone on the device (dq_probe.h).
usage: DQ_HIP_LIB=<variant of k_linattn.hip built with -DDQ_KPROBE> tools/probe_la.py C n [B]     rows = B * 400"""
import ctypes, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffusion-deconvolution-dia-msms-data_amd"))
import torch
from dquartic import _native as N
C, n = int(sys.argv[1]), int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 512
L = N.lib(); lib = ctypes.CDLL(N.LIB_PATH)
rows = B * 400
g = torch.Generator().manual_seed(0)
x = torch.randn(rows, C, n, generator=g).cuda(); y = torch.empty_like(x)
w = (torch.randn(384, C, generator=g) * .4).cuda(); wo = (torch.randn(C, 128, generator=g) * .2).cuda()
bo, g1, g2 = torch.zeros(C).cuda(), torch.ones(C).cuda(), torch.ones(C).cuda()
if os.environ.get("NET"):  # in situ: the k_linattn_fwd<C,n> launches of a sampling step of the default network (prepared weights)
    from dquartic.model.model import DDIMDiffusionModel
    from dquartic.model.unet1d import UNet1d
    torch.manual_seed(0)
    net = UNet1d(dim=4, channels=1, dim_mults=(1, 2, 2, 3, 3, 4, 4), conditional=True, init_cond_channels=1, attn_cond_channels=1, downsample_dim=64, simple=True).cuda()
    dm = DDIMDiffusionModel(model_class=net, device="cuda")
    xs_, c2_, c1_ = torch.randn(B, 400, 64, device="cuda"), torch.rand(B, 400, 64, device="cuda"), torch.rand(B, 400, device="cuda")
    f = lambda: dm.sample(xs_, c2_, c1_, num_steps=2)
else:
  f = lambda: N.check(L.dq_linattn_fwd(N.ptr(x), N.ptr(y), None, N.ptr(w), N.ptr(wo), N.ptr(bo), N.ptr(g1), N.ptr(g2), C, rows, n, N.stream_ptr()), "f")
for _ in range(3): f()
torch.cuda.synchronize()
assert lib.dq_kprobe_clear() == 0 and lib.dq_kprobe_select(300000 + C * 100 + n) == 0
f(); torch.cuda.synchronize()
buf = np.zeros(4096 * 16, dtype=np.uint64)
assert lib.dq_kprobe_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
st = buf.reshape(4096, 16).astype(np.int64); st = st[st[:, 0] > 0]
names = ["weights staged (+ barrier)", "x loaded, normalised, staged", "four heads", "post-norm, store"]
print(f"k_linattn_fwd<{C},{n}> rows {rows}: {len(st)} workgroups sampled (the last to use each of 4096 slots)")
for i, nm in enumerate(names):
    d = st[:, i + 1] - st[:, i]
    print(f"  {nm:32s} {int(np.median(d)):8d} {int(d.max()):8d}")
if st[:, 9].max() > 0:
    print(f"  entry -> loads issued {int(np.median(st[:, 9] - st[:, 0]))}, -> images stored {int(np.median(st[:, 10] - st[:, 9]))}, -> barrier passed {int(np.median(st[:, 1] - st[:, 10]))}")
for w in (1, 2, 3):
    d = st[:, 5 + w] - st[:, 0]
    print(f"  wave {w} enters {int(np.median(d)):6d} clocks after wave 0 (min {int(d.min())}, max {int(d.max())})")
print(f"  workgroup life {int(np.median(st[:, 4] - st[:, 0]))} clocks (median)")
