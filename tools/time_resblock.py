#!/usr/bin/env python3
"""Times the stand-alone ResnetBlock entry points (dq_resblock_fwd / dq_resblock_bwd) at the network's shapes, alone on the device.
usage: python tools/time_resblock.py [B] [fwd|bwd|both]      (rows = B * 400)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffusion-deconvolution-dia-msms-data_amd"))
import torch
from dquartic import _native as N

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
what = sys.argv[2] if len(sys.argv) > 2 else "both"
RT = 400
L = N.lib()
shapes = [(4, 4, 64), (8, 4, 64), (4, 4, 32), (12, 8, 32), (8, 8, 16), (16, 8, 16), (8, 8, 8), (20, 12, 8), (12, 12, 4), (24, 12, 4), (12, 12, 2), (28, 16, 2),
          (16, 16, 1), (32, 16, 1)]
dev = "cuda"
for cin, cout, n in shapes:
    rows = B * RT
    g = torch.Generator().manual_seed(0)
    nparam = 2 * cout * 16 + 2 * cout + cout * cin * 3 + cout + cout + cout * cout * 3 + cout + cout + (cout * cin + cout if cin != cout else 0)
    flat = (torch.randn(nparam, generator=g) * 0.3).to(dev)
    cinA = cout if cin != cout else cin
    xA = torch.randn(rows, cinA, n, generator=g).to(dev)
    xB = torch.randn(rows, cin - cinA, n, generator=g).to(dev) if cin != cinA else None
    temb = torch.randn(B, 16, generator=g).to(dev)
    nws = L.dq_resblock_workspace_floats(cin, cout, rows, n, RT)
    ws = torch.empty(nws, device=dev)
    out = torch.empty(rows, cout, n, device=dev)
    gy = torch.randn(rows, cout, n, generator=g).to(dev)
    dA, dB = torch.zeros_like(xA), (torch.zeros_like(xB) if xB is not None else None)
    grads = torch.zeros_like(flat)
    dss = torch.empty(B, 2 * cout, device=dev)

    def fwd(save):
        N.check(L.dq_resblock_fwd(N.ptr(flat), N.ptr(xA), cinA, N.ptr(xB), cin - cinA, N.ptr(temb), N.ptr(out), cout, rows, n, RT, save, N.ptr(ws), nws,
                                  N.stream_ptr()), "fwd")

    def bwd():
        N.check(L.dq_resblock_bwd(N.ptr(flat), N.ptr(xA), cinA, N.ptr(xB), cin - cinA, N.ptr(gy), N.ptr(dA), N.ptr(dB), N.ptr(grads), N.ptr(dss), cout, rows,
                                  n, RT, N.ptr(ws), nws, N.stream_ptr()), "bwd")

    def t(fn, iters=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters

    line = f"cin {cin:2d} cout {cout:2d} n {n:2d} rows {rows}:"
    if what in ("fwd", "both"):
        tf0, tf1 = t(lambda: fwd(0)), t(lambda: fwd(1))
        by = 4 * (cin + cout) * n * rows
        line += f"  fwd(inference) {tf0:7.1f} us ({by / tf0 / 1e6:5.2f} TB/s incl. the out copy)  fwd(train) {tf1:7.1f} us"
    if what in ("bwd", "both"):
        fwd(1)
        tb = t(bwd)
        line += f"  bwd(all launches of the entry point) {tb:7.1f} us"
    print(line, flush=True)
