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
print(f"  workgroup life {int(np.median(st[:, 4] - st[:, 0]))} clocks (median)")
