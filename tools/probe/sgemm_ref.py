"""What does the vendor library reach for fp32 GEMMs on this box?  (context for k_gemm.hip's fraction of the nominal MFMA peak)"""
import torch, time
dev = torch.device("cuda", 0)
torch.backends.cuda.matmul.allow_tf32 = False
for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 1024), (1088, 40000, 1024), (1088, 1024, 40000), (1088, 4096, 1024)]:
    A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    for _ in range(3):
        C = A @ B.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        C = A @ B.t()
    e1.record()
    torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) / 10 * 1e-3
    print(f"torch (hipBLASLt/rocBLAS) fp32 {M}x{N}x{K}: {dt*1e6:9.1f} us {2.0*M*N*K/dt/1e12:7.2f} TFLOP/s", flush=True)
