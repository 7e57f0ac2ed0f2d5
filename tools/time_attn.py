#!/usr/bin/env python3
"""Times the bottleneck attention alone (dq_attn_fwd, dq_attn_bwd) at (B, 4 heads x 32, RT).  usage: python tools/time_attn.py [B] [RT]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffusion-deconvolution-dia-msms-data_amd"))
import torch
from dquartic import _native as N
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
RT = int(sys.argv[2]) if len(sys.argv) > 2 else 400
L = N.lib()
g = torch.Generator().manual_seed(0)
q, k, v, go = (torch.randn(B, 128, RT, generator=g).cuda() for _ in range(4))
o, dq, dk, dv = (torch.empty_like(q) for _ in range(4))
lse, delta = torch.empty(B * 4 * RT, device="cuda"), torch.empty(B * 4 * RT, device="cuda")
fwd = lambda: N.check(L.dq_attn_fwd(N.ptr(q), N.ptr(k), N.ptr(v), N.ptr(o), N.ptr(lse), B, RT, N.stream_ptr()), "f")
bwd = lambda: N.check(L.dq_attn_bwd(N.ptr(q), N.ptr(k), N.ptr(v), N.ptr(o), N.ptr(go), N.ptr(lse), N.ptr(delta), N.ptr(dq), N.ptr(dk), N.ptr(dv), B, RT, N.stream_ptr()), "b")
for name, f in (("fwd", fwd), ("bwd (dQ + dK/dV)", bwd)):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    nb = (RT + 31) // 32
    flops = B * 4 * nb * nb * 32 * 32 * 32 * 2 * (2 if name == "fwd" else 7)
    us = e0.elapsed_time(e1) * 50
    print(f"attention {name} B {B} RT {RT}: {us:8.1f} us per call, {flops / us / 1e6:6.1f} TFLOP/s of 32-block products")
