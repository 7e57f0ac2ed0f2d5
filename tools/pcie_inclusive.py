"""Train-step rate when every batch starts in (pinned) host memory: the three tensors of a batch are copied host -> device on the
step's stream before the native step (bench.py's `value` keeps inputs resident; this is the PCIe-inclusive figure of DESIGN 6)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "diffusion-deconvolution-dia-msms-data_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import bench

    dev = torch.device("cuda:0")
    net, dm = bench.build_model(dev)
    dm._set_optimizer(1e-5)
    dev_batches = bench.make_batches(8, bench.TRAIN_BATCH, 0, 1, dev)
    host = [tuple(t.cpu().pin_memory() for t in b) for b in dev_batches]
    nbytes = sum(t.numel() * 4 for t in host[0])

    def run(resident, steps=40):
        for i in range(10):
            x0, c2, c1 = dev_batches[i % 8] if resident else (t.to(dev, non_blocking=True) for t in host[i % 8])
            dm._train_one_batch(x0, ms2_cond=c2, ms1_cond=c1, sync=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            x0, c2, c1 = dev_batches[i % 8] if resident else (t.to(dev, non_blocking=True) for t in host[i % 8])
            dm._train_one_batch(x0, ms2_cond=c2, ms1_cond=c1, sync=False)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    a, b = run(True), run(False)
    print(f"resident: {a * 1e3:.3f} ms/step = {bench.TRAIN_BATCH / a:.1f} windows/s ; host batches ({nbytes / 1e6:.2f} MB per step over PCIe, "
          f"pinned, same stream): {b * 1e3:.3f} ms/step = {bench.TRAIN_BATCH / b:.1f} windows/s", flush=True)


if __name__ == "__main__":
    main()
