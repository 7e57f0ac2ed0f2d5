"""PMC driver for the fp32 GEMM (k_gemm.hip): a few launches of the transformer's output projection at batch 32
((1088 x 1024) x (1024 x 40000)) and of the 4096^3 product, plus the q_sample calibration launch of known traffic."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffusion-deconvolution-dia-msms-data_amd"))
import torch
from dquartic import _native as N
L = N.lib()
for (M, Nn, K) in [(1088, 40000, 1024), (4096, 4096, 4096)]:
    A, B, C = torch.randn(M, K, device="cuda"), torch.randn(Nn, K, device="cuda"), torch.empty(M, Nn, device="cuda")
    scr = torch.empty(max(int(L.dq_gemm_scratch_floats(M, Nn, K)), 4), device="cuda")
    for _ in range(5):
        N.check(L.dq_gemm(N.ptr(A), N.ptr(B), N.ptr(C), None, M, Nn, K, K, K, Nn, 1, 1, 0, 0, N.ptr(scr), scr.numel(), N.stream_ptr()), "g")
    print("gemm", M, Nn, K, "A+B+C bytes", 4 * (M * K + Nn * K + M * Nn))
x = torch.randn(50, 262144, device="cuda"); nz = torch.randn_like(x); y = torch.empty_like(x)
ab = torch.linspace(0.9, 0.1, 1000).cuda(); t = torch.zeros(50, dtype=torch.long).cuda()
for _ in range(5):
    N.check(L.dq_q_sample(N.ptr(ab), N.ptr(x), N.ptr(t), N.ptr(nz), N.ptr(y), 50, 262144, 0, N.stream_ptr()), "q")
torch.cuda.synchronize()
print("calibration tensor bytes", x.numel() * 4)
