"""Stand-alone timing of the LinearAttention backward (dq_linattn_bwd: prepare + backward + slot reduce) per (C, n) in both forms:
   python tools/time_la_bwd.py [rows]     (default 12,800 = a train batch of 32 windows)
Prints microseconds per call (HIP events over 50 calls after 5 warm-up calls); the two forms share the prepare / reduce launches."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffusion-deconvolution-dia-msms-data_amd"))
from dquartic import _native as N  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 12800
L = N.lib()
for C, n in ((12, 4), (12, 2), (16, 2), (16, 4), (8, 4), (8, 2), (12, 8), (8, 8)):
    g = torch.Generator().manual_seed(C * 10 + n)
    x = torch.randn(rows, C, n, generator=g).cuda()
    gy = torch.randn(rows, C, n, generator=g).cuda()
    w = (torch.randn(384 * C, generator=g) * 0.4).cuda(); wo = (torch.randn(C * 128, generator=g) * 0.2).cuda()
    bo = (torch.randn(C, generator=g) * 0.1).cuda(); g1 = (torch.rand(C, generator=g) + 0.5).cuda(); g2 = (torch.rand(C, generator=g) + 0.5).cuda()
    y, ypre, dx = torch.empty_like(x), torch.empty_like(x), torch.zeros_like(x)
    dw, dwo, dbo, dg1, dg2 = (torch.zeros_like(t) for t in (w, wo, bo, g1, g2))
    scratch = torch.empty(2 * x.numel() + 2048 * 512 * C, device="cuda")
    N.check(L.dq_linattn_fwd(N.ptr(x), N.ptr(y), N.ptr(ypre), N.ptr(w), N.ptr(wo), N.ptr(bo), N.ptr(g1), N.ptr(g2), C, rows, n, N.stream_ptr()), "fwd")
    out = []
    for form, v in (("rows", 0), ("register", 1 << 40)):
        N.set_option("la_rows_bwd_min_rows", v)

        def call():
            N.check(L.dq_linattn_bwd(N.ptr(x), N.ptr(ypre), N.ptr(gy), N.ptr(dx), N.ptr(w), N.ptr(wo), N.ptr(bo), N.ptr(g1), N.ptr(g2), N.ptr(dw),
                                     N.ptr(dwo), N.ptr(dbo), N.ptr(dg1), N.ptr(dg2), N.ptr(scratch), C, rows, n, N.stream_ptr()), "bwd")
        for _ in range(5):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            call()
        e1.record()
        torch.cuda.synchronize()
        out.append("%s %.1f us" % (form, e0.elapsed_time(e1) * 1000 / 50))
    print("C=%d n=%d rows=%d: %s" % (C, n, rows, ", ".join(out)), flush=True)
N.set_option("la_rows_bwd_min_rows", -1)
