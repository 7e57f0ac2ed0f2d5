#!/usr/bin/env python3
"""Times dq_linattn_fwd + dq_linattn_bwd alone (rows = B * 400) at given (C, n).  usage: python tools/time_la_bwd.py C n [B]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffusion-deconvolution-dia-msms-data_amd"))
import torch
from dquartic import _native as N
C, n = int(sys.argv[1]), int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
L = N.lib()
rows = B * 400
g = torch.Generator().manual_seed(0)
x = torch.randn(rows, C, n, generator=g).cuda(); y = torch.empty_like(x); ypre = torch.empty_like(x)
dy = torch.randn(rows, C, n, generator=g).cuda(); dx = torch.zeros_like(x)
w = (torch.randn(384, C, generator=g) * float(os.environ.get("WSCALE", "0.4"))).cuda(); wo = (torch.randn(C, 128, generator=g) * .2).cuda()
bo, g1, g2 = torch.zeros(C).cuda(), torch.ones(C).cuda(), torch.ones(C).cuda()
dw, dwo, dbo, dg1, dg2 = torch.zeros_like(w), torch.zeros_like(wo), torch.zeros(C).cuda(), torch.zeros(C).cuda(), torch.zeros(C).cuda()
scratch = torch.empty(L.dq_linattn_bwd_scratch_floats(C, rows, n) if hasattr(L, "dq_linattn_bwd_scratch_floats") else 64 << 20, device="cuda")
N.check(L.dq_linattn_fwd(N.ptr(x), N.ptr(y), N.ptr(ypre), N.ptr(w), N.ptr(wo), N.ptr(bo), N.ptr(g1), N.ptr(g2), C, rows, n, N.stream_ptr()), "f")
f = lambda: N.check(L.dq_linattn_bwd(N.ptr(x), N.ptr(ypre), N.ptr(dy), N.ptr(dx), N.ptr(w), N.ptr(wo), N.ptr(bo), N.ptr(g1), N.ptr(g2), N.ptr(dw), N.ptr(dwo), N.ptr(dbo),
                                     N.ptr(dg1), N.ptr(dg2), N.ptr(scratch), C, rows, n, N.stream_ptr()), "b")
for _ in range(3): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
print(f"linattn_bwd C {C} n {n} rows {rows} wscale {os.environ.get('WSCALE', '0.4')}: {e0.elapsed_time(e1) * 100:8.1f} us per call (kernel + prepare + reduces)")
