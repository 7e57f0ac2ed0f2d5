"""Experiment: is the train step host-bound?  Times train_step_fused (forward + loss + backward, no optimiser) issued eagerly
and replayed from a torch.cuda.CUDAGraph, plus the host time to issue one eager step, for the U-Net at batch 32 and the
CustomTransformer at batch 1.  Run on the GPU box: python tools/graph_train.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "diffusion-deconvolution-dia-msms-data_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def measure(name, dm, x0, c2, c1, steps=30):
    B = x0.shape[0]
    t = torch.randint(0, 1000, (B,), device=x0.device)
    noise = torch.randn_like(x0)

    def one():
        return dm.train_step_fused(x0, c2, c1, t=t, noise=noise, zero_grads=True)

    for _ in range(5):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    host = (time.perf_counter() - t0) / steps
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / steps
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            one()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    try:
        with torch.cuda.graph(g):
            loss = one()
    except Exception as e:  # noqa: BLE001
        print(f"{name}: capture failed: {e}", flush=True)
        return
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / steps
    print(f"{name}: eager {eager * 1e3:.3f} ms/step (host issue {host * 1e3:.3f} ms), graph replay {graph * 1e3:.3f} ms/step, loss {float(loss):.5f}",
          flush=True)


def main():
    import bench
    from dquartic.model.building_blocks import CustomTransformer, DDIMTransformerAdapter
    from dquartic.model.model import DDIMDiffusionModel

    dev = torch.device("cuda:0")
    net, dm = bench.build_model(dev)
    x0, c2, c1 = bench.make_batches(1, 32, 0, 1, dev)[0]
    measure("unet b32", dm, x0, c2, c1)
    del net, dm
    torch.cuda.empty_cache()
    torch.manual_seed(0)
    tn = DDIMTransformerAdapter(CustomTransformer(**bench.TFM_CFG)).to(dev)
    dm = DDIMDiffusionModel(model_class=tn, num_timesteps=1000, beta_schedule_type="cosine", pred_type="eps", auto_normalize=True,
                            ms1_loss_weight=0.0, device=dev)
    D = bench.TFM_CFG["input_dim"]
    for B in (1, 32):
        x0, c2, c1 = torch.rand(B, bench.TFM_RT, D, device=dev), torch.rand(B, bench.TFM_RT, D, device=dev), torch.rand(B, bench.TFM_RT, device=dev)
        measure(f"tfm b{B}", dm, x0, c2, c1, steps=10)


if __name__ == "__main__":
    main()
