#!/usr/bin/env python3
"""Times the PREPARED LinearAttention forward (the network's path) at rows of 2 / 4 positions under the dispatch rules: k_la_small (forced) against
k_la_rows_fwd (k_la_small off), per row count.  usage: python tools/time_la_forms.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffusion-deconvolution-dia-msms-data_amd"))
import torch
from dquartic import _native as N
L = N.lib()
for C, n in ((12, 4), (12, 2), (16, 2), (16, 4), (8, 4), (12, 8)):  # (rows of 8 positions: k_la_small against the register-resident kernel)
    for B in (32, 128, 512):
        rows = B * 400
        g = torch.Generator().manual_seed(0)
        x = torch.randn(rows, C, n, generator=g).cuda(); y = torch.empty_like(x)
        w = (torch.randn(384, C, generator=g) * .4).cuda(); wo = (torch.randn(C, 128, generator=g) * .2).cuda()
        bo, g1, g2 = torch.zeros(C).cuda(), torch.ones(C).cuda(), torch.ones(C).cuda()
        prep = torch.zeros(L.dq_linattn_prep_floats(), device="cuda")
        N.check(L.dq_linattn_prepare(N.ptr(w), N.ptr(wo), N.ptr(g1), C, N.ptr(prep), N.stream_ptr()), "prep")
        f = lambda: N.check(L.dq_linattn_fwd_prepared(N.ptr(x), N.ptr(y), None, N.ptr(w), N.ptr(wo), N.ptr(bo), N.ptr(g1), N.ptr(g2), N.ptr(prep), C, rows, n,
                                                      N.stream_ptr()), "f")
        out = []
        for name, small in (("k_la_small", 0), ("k_la_rows_fwd", 1 << 40)):
            N.set_option("la_small_min_rows", small)
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            out.append(f"{name} {e0.elapsed_time(e1) * 50:7.1f} us")
        N.set_option("la_small_min_rows", -1)
        print(f"C {C:2d} n {n} rows {rows:6d}: " + "   ".join(out), flush=True)
