"""CustomTransformer at the reference's configuration (dquartic_train_config.json: input_dim 40000, hidden 1024, 8 heads, 8 layers;
windows of 34 retention-time rows, `generate-data-slices --window-size 34`): time of the train step (q_sample + forward + MSE +
backward + clipped AdamW, all native), of the forward alone, and of the large GEMMs by themselves (fp32 MFMA roofline)."""
import os, sys, time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "diffusion-deconvolution-dia-msms-data_amd"))
from dquartic import _native as N  # noqa: E402
from dquartic.model.building_blocks import CustomTransformer, DDIMTransformerAdapter  # noqa: E402
from dquartic.model.model import DDIMDiffusionModel  # noqa: E402
from dquartic.model.model_interface import ModelInterface  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
D, H, HEADS, LAYERS, RT = 40000, 1024, 8, 8, 34
PEAK = 157.3e12


def ev_time(fn, iters):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def gemm_bench():
    lib = N.lib()
    for (M, Nn, K, a_k, b_k, what) in [(34, H, D, 1, 1, "input proj fwd, B=1"), (1088, H, D, 1, 1, "input proj fwd, B=32"),
                                       (1088, D, H, 1, 1, "output proj fwd, B=32"), (1088, 4 * H, H, 1, 1, "ff.0 fwd, B=32"),
                                       (1088, H, D, 1, 0, "output proj dX, B=32"), (D, H, 1088, 0, 0, "output proj dW, B=32"),
                                       (H, D, 1088, 0, 0, "input proj dW, B=32"), (4096, 4096, 4096, 1, 1, "square 4096"),
                                       (8192, 8192, 1024, 1, 1, "8192 x 8192 x 1024")]:
        A = torch.randn((M, K) if a_k else (K, M), device=dev)
        B = torch.randn((Nn, K) if b_k else (K, Nn), device=dev)
        C = torch.empty(M, Nn, device=dev)
        scr = torch.empty(max(int(lib.dq_gemm_scratch_floats(M, Nn, K)), 4), device=dev)

        def run():
            N.check(lib.dq_gemm(N.ptr(A), N.ptr(B), N.ptr(C), None, M, Nn, K, A.shape[1], B.shape[1], Nn, a_k, b_k, 0, 0, N.ptr(scr), scr.numel(),
                                N.stream_ptr()), "dq_gemm")
        dt = ev_time(run, 10)
        fl = 2.0 * M * Nn * K
        by = 4.0 * (M * K + Nn * K + M * Nn)
        print(f"gemm {what:28s} M={M:6d} N={Nn:6d} K={K:6d}: {dt * 1e6:9.1f} us  {fl / dt / 1e12:7.2f} TFLOP/s ({fl / dt / PEAK * 100:5.1f} % of fp32 MFMA peak)  "
              f"{by / dt / 1e9:8.1f} GB/s", flush=True)


def train_bench(B, steps):
    torch.manual_seed(0)
    net = DDIMTransformerAdapter(CustomTransformer(D, H, HEADS, LAYERS)).to(dev)
    dm = DDIMDiffusionModel(model_class=net, num_timesteps=1000, beta_schedule_type="cosine", pred_type="eps", auto_normalize=True,
                            ms1_loss_weight=0.0, device=dev)
    dm._set_optimizer(1e-5)
    x0, c2, c1 = torch.rand(B, RT, D, device=dev), torch.rand(B, RT, D, device=dev), torch.rand(B, RT, device=dev)

    def step():
        dm._train_one_batch(x0, ms2_cond=c2, ms1_cond=c1, sync=False)
    dt = ev_time(step, steps)
    t = torch.randint(0, 1000, (B,), device=dev)
    tf = net.transformer

    def fwd():
        with torch.no_grad():
            tf(x0, t, c1)
    df = ev_time(fwd, steps)
    n_par = tf.flat_params.numel()
    fl_fwd = 2.0 * B * RT * (2 * D * H + LAYERS * (12 * H * H)) + 2.0 * B * RT * LAYERS * 2 * H * 2 * RT  # dense layers + attention
    print(f"B={B:3d}: train step {dt * 1e3:8.3f} ms = {B / dt:8.1f} windows/s ; forward {df * 1e3:8.3f} ms = {B / df:8.1f} windows/s ; "
          f"{n_par} params ; fwd {fl_fwd / df / 1e12:6.2f} TFLOP/s, step {3 * fl_fwd / dt / 1e12:6.2f} TFLOP/s (dense-layer FLOPs)", flush=True)
    del net, dm
    torch.cuda.empty_cache()


if __name__ == "__main__":
    ModelInterface  # noqa: B018 (imported for the side effect of checking the harness imports)
    if os.environ.get("TFM_B"):  # profiling runs: one batch size, no GEMM sweep
        train_bench(int(os.environ["TFM_B"]), 10)
    else:
        gemm_bench()
        for B in (1, 8, 32):
            train_bench(B, 10)
