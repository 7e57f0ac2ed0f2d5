"""Would two half-batch lanes on two streams beat one full-batch lane?  The deep U-Net levels are latency-bound (a few hundred
waves per launch), so a second lane could fill the machine while the first one waits.  Two independent plans (each with its own
workspace, gradient buffer and weight-gradient side stream) run train_step_fused at B/2 on two torch streams; compared with one
plan at B on one stream.  Forward + backward only (no optimiser) in both legs."""
import os, sys, time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
B = bench.TRAIN_BATCH
LANES = int(os.environ.get("LANES", "2"))
nets = [bench.build_model(dev) for _ in range(LANES + 1)]
x0, c2, c1 = bench.make_batches(1, B, 0, 1, dev)[0]
t = torch.randint(0, 1000, (B,), device=dev)
noise = torch.randn_like(x0)


def one(steps):
    net, dm = nets[0]
    for _ in range(steps):
        dm.train_step_fused(x0, c2, c1, t=t, noise=noise)


streams = [torch.cuda.Stream(device=dev) for _ in range(LANES)]
h = B // LANES
parts = [(x0[i * h:(i + 1) * h].contiguous(), c2[i * h:(i + 1) * h].contiguous(), c1[i * h:(i + 1) * h].contiguous(),
          t[i * h:(i + 1) * h].contiguous(), noise[i * h:(i + 1) * h].contiguous()) for i in range(LANES)]


def lanes(steps):
    for _ in range(steps):
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                a, b, c, tt, nn = parts[i]
                nets[1 + i][1].train_step_fused(a, b, c, t=tt, noise=nn)


def timeit(tag, fn, steps=20):
    fn(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(steps)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{tag}: {dt / steps * 1e3:.3f} ms per {B} windows (host issue {th / steps * 1e3:.3f} ms)", flush=True)


timeit("one lane ", one)
timeit(f"{LANES} lanes", lanes)
timeit("one lane ", one)
timeit(f"{LANES} lanes", lanes)
