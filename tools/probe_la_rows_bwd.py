#!/usr/bin/env python3
"""Interval probe of one k_la_rows_bwd instantiation alone on the device (dq_probe.h): shader clocks of wave 0 (head 0) of every workgroup,
for the LAST tile a workgroup walks (the stamps of earlier tiles are overwritten).
usage: DQ_HIP_LIB=<variant of k_la_rows_bwd.hip built with -DDQ_KPROBE> tools/probe_la_rows_bwd.py C n [rows]"""
import ctypes, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffusion-deconvolution-dia-msms-data_amd"))
import torch
from dquartic import _native as N
C, n = int(sys.argv[1]), int(sys.argv[2])
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 12800
L = N.lib(); lib = ctypes.CDLL(N.LIB_PATH)
g = torch.Generator().manual_seed(0)
x = torch.randn(rows, C, n, generator=g).cuda(); gy = torch.randn(rows, C, n, generator=g).cuda()
w = (torch.randn(384 * C, generator=g) * .4).cuda(); wo = (torch.randn(C * 128, generator=g) * .2).cuda()
bo, g1, g2 = torch.zeros(C).cuda(), torch.ones(C).cuda(), torch.ones(C).cuda()
y, ypre, dx = torch.empty_like(x), torch.empty_like(x), torch.zeros_like(x)
dw, dwo, dbo, dg1, dg2 = (torch.zeros_like(t) for t in (w, wo, bo, g1, g2))
scratch = torch.empty(2 * x.numel() + 2048 * 512 * C, device="cuda")
N.check(L.dq_linattn_fwd(N.ptr(x), N.ptr(y), N.ptr(ypre), N.ptr(w), N.ptr(wo), N.ptr(bo), N.ptr(g1), N.ptr(g2), C, rows, n, N.stream_ptr()), "fwd")
f = lambda: N.check(L.dq_linattn_bwd(N.ptr(x), N.ptr(ypre), N.ptr(gy), N.ptr(dx), N.ptr(w), N.ptr(wo), N.ptr(bo), N.ptr(g1), N.ptr(g2), N.ptr(dw), N.ptr(dwo),
                                     N.ptr(dbo), N.ptr(dg1), N.ptr(dg2), N.ptr(scratch), C, rows, n, N.stream_ptr()), "bwd")
for _ in range(3): f()
torch.cuda.synchronize()
assert lib.dq_kprobe_clear() == 0 and lib.dq_kprobe_select(500000 + C * 100 + n) == 0
f(); torch.cuda.synchronize()
buf = np.zeros(4096 * 16, dtype=np.uint64)
assert lib.dq_kprobe_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
st = buf.reshape(4096, 16).astype(np.int64); st = st[st[:, 0] > 0]
names = ["image copy + barrier", "(earlier tiles) + loads + norms", "ux/ud + barrier A", "k, q projections + softmaxes", "S / Z / dZ / dS per position",
         "q side", "k side", "exchange + barrier B", "(last wave's finish) -> flush start", "flush"]
print(f"k_la_rows_bwd<{C},{n}> rows {rows}: {len(st)} workgroups sampled; median / max clocks per interval (last tile of a workgroup)")
for i, nm in enumerate(names):
    d = st[:, i + 1] - st[:, i]
    print(f"  {nm:42s} {int(np.median(d)):8d} {int(d.max()):8d}")
print(f"  workgroup life {int(np.median(st[:, 10] - st[:, 0]))} clocks (median), {int((st[:, 10] - st[:, 0]).max())} max; launch span {int(st[:, 10].max() - st[:, 0].min())} clocks")
