"""PMC driver: a few launches of the level-0 LinearAttention forward/backward (C=4, n=64, 12800 rows) and of a plain
device copy of known size (calibration of FETCH_SIZE / WRITE_SIZE for this access pattern)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffusion-deconvolution-dia-msms-data_amd"))
import torch
from dquartic import _native as N
C, n, rows = 4, 64, 12800
SAMPLE = len(sys.argv) > 1 and sys.argv[1] == "sample"   # the sampling leg's launch: 512 x 400 rows, inference (no pre-norm save)
if SAMPLE:
    rows = 512 * 400
g = torch.Generator().manual_seed(0)
x = torch.randn(rows, C, n, generator=g).cuda(); dy = torch.randn(rows, C, n, generator=g).cuda()
w = (torch.randn(384, C, generator=g) * .4).cuda(); wo = (torch.randn(C, 128, generator=g) * .2).cuda()
bo, g1, g2 = torch.zeros(C).cuda(), torch.ones(C).cuda(), torch.ones(C).cuda()
y, ypre, dx = torch.empty_like(x), torch.empty_like(x), torch.zeros_like(x)
dw, dwo, dbo, dg1, dg2 = (torch.zeros_like(t) for t in (w, wo, bo, g1, g2))
scratch = torch.empty(2 * x.numel() + 2048 * 512 * C, device="cuda")
L = N.lib()
prep = torch.zeros(L.dq_linattn_prep_floats(), device="cuda")   # the network's path: weights prepared once (dq_linattn_prepare)
N.check(L.dq_linattn_prepare(N.ptr(w), N.ptr(wo), N.ptr(g1), C, N.ptr(prep), N.stream_ptr()), "p")
if SAMPLE:
    for _ in range(5):
        N.check(L.dq_linattn_fwd_prepared(N.ptr(x), N.ptr(y), None, N.ptr(w), N.ptr(wo), N.ptr(bo), N.ptr(g1), N.ptr(g2), N.ptr(prep), C, rows, n, N.stream_ptr()), "f")
    torch.cuda.synchronize()
    print("bytes per tensor", x.numel() * 4)
    sys.exit(0)
for _ in range(5):
    N.check(L.dq_linattn_fwd_prepared(N.ptr(x), N.ptr(y), N.ptr(ypre), N.ptr(w), N.ptr(wo), N.ptr(bo), N.ptr(g1), N.ptr(g2), N.ptr(prep), C, rows, n, N.stream_ptr()), "f")
    N.check(L.dq_linattn_bwd(N.ptr(x), N.ptr(ypre), N.ptr(dy), N.ptr(dx), N.ptr(w), N.ptr(wo), N.ptr(bo), N.ptr(g1), N.ptr(g2), N.ptr(dw), N.ptr(dwo), N.ptr(dbo), N.ptr(dg1), N.ptr(dg2), N.ptr(scratch), C, rows, n, N.stream_ptr()), "b")
# the deep levels' backward since round 5 (k_la_rows_bwd.hip): 12 channels, rows of 4 positions, the train step's 12,800 rows
C2, n2 = 12, 4
x2 = torch.randn(rows, C2, n2, generator=g).cuda(); dy2 = torch.randn(rows, C2, n2, generator=g).cuda()
w2_ = (torch.randn(384, C2, generator=g) * .4).cuda(); wo2 = (torch.randn(C2, 128, generator=g) * .2).cuda()
bo2, g12, g22 = torch.zeros(C2).cuda(), torch.ones(C2).cuda(), torch.ones(C2).cuda()
y2, ypre2, dx2 = torch.empty_like(x2), torch.empty_like(x2), torch.zeros_like(x2)
dw2_, dwo2, dbo2, dg12, dg22 = (torch.zeros_like(t) for t in (w2_, wo2, bo2, g12, g22))
scratch2 = torch.empty(2 * x2.numel() + 2048 * 512 * C2, device="cuda")
N.check(L.dq_linattn_fwd(N.ptr(x2), N.ptr(y2), N.ptr(ypre2), N.ptr(w2_), N.ptr(wo2), N.ptr(bo2), N.ptr(g12), N.ptr(g22), C2, rows, n2, N.stream_ptr()), "f2")
for _ in range(5):
    N.check(L.dq_linattn_bwd(N.ptr(x2), N.ptr(ypre2), N.ptr(dy2), N.ptr(dx2), N.ptr(w2_), N.ptr(wo2), N.ptr(bo2), N.ptr(g12), N.ptr(g22), N.ptr(dw2_), N.ptr(dwo2), N.ptr(dbo2), N.ptr(dg12), N.ptr(dg22), N.ptr(scratch2), C2, rows, n2, N.stream_ptr()), "b2")
# calibration: q_sample streams 3 tensors of rows*C*n floats (2 reads + 1 write) with 16 B per lane
ab = torch.linspace(0.9, 0.1, 1000).cuda(); t = torch.zeros(50, dtype=torch.long).cuda()
for _ in range(5):
    N.check(L.dq_q_sample(N.ptr(ab), N.ptr(x), N.ptr(t), N.ptr(dy), N.ptr(y), 50, x.numel() // 50, 0, N.stream_ptr()), "q")
# the HBM-bound kernels bench.py reports under roofline_hbm, at the same shapes
RT = 400
cin, cout, B = 8, 4, 32
nparam = 2 * cout * 16 + 2 * cout + cout * cin * 3 + 2 * cout + cout * cout * 3 + 2 * cout + cout * cin + cout
flat = (torch.randn(nparam, generator=g) * 0.3).cuda()
xA, xB = torch.randn(rows, cout, n, generator=g).cuda(), torch.randn(rows, cin - cout, n, generator=g).cuda()
temb = torch.randn(B, 16, generator=g).cuda()
nws = L.dq_resblock_workspace_floats(cin, cout, rows, n, RT)
ws = torch.zeros(nws, device="cuda"); outb = torch.empty(rows, cout, n, device="cuda")
N.check(L.dq_resblock_fwd(N.ptr(flat), N.ptr(xA), cout, N.ptr(xB), cin - cout, N.ptr(temb), N.ptr(outb), cout, rows, n, RT, 1, N.ptr(ws), nws, N.stream_ptr()), "rf")
dA, dB, grads = torch.empty_like(xA), torch.empty_like(xB), torch.zeros_like(flat)
for _ in range(5):
    N.check(L.dq_resblock_bwd(N.ptr(flat), N.ptr(xA), cout, N.ptr(xB), cin - cout, None, N.ptr(dA), N.ptr(dB), N.ptr(grads), None, cout, rows, n, RT, N.ptr(ws), nws, N.stream_ptr()), "rb")
SB = 512
params = (torch.randn(L.dq_level_param_floats(0, 4, 4, 0, 2), generator=g) * 0.3).cuda()
xl = torch.randn(SB * RT, 4, n, generator=g).cuda(); tl = torch.randn(SB, 16, generator=g).cuda()
o0, o1 = torch.empty_like(xl), torch.empty_like(xl); wsl = torch.empty(2 * SB * 8 + 8256, device="cuda")
for _ in range(3):
    N.check(L.dq_level_fwd(N.ptr(params), 0, N.ptr(xl), 4, None, None, 0, N.ptr(tl), N.ptr(o0), N.ptr(o1), 4, 2, SB * RT, n, RT, N.ptr(wsl), wsl.numel(), N.stream_ptr()), "lv")
# calibration of FETCH_SIZE for 4-byte-per-lane reads: k_rmsnorm_fwd reads one tensor and writes one
for _ in range(5):
    N.check(L.dq_rmsnorm_fwd(N.ptr(x), N.ptr(g1), N.ptr(y), C, rows, n, N.stream_ptr()), "rn")
torch.cuda.synchronize()
print("bytes per tensor", x.numel() * 4)
