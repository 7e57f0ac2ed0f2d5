"""Where does the train step lose time once a process group exists?  One process, three measurements:
  (1) no process group, (2) RCCL process group initialised but no collective in the step, (3) with the flat all-reduce."""
import os, sys, time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
net, dm = bench.build_model(dev)
dm._set_optimizer(1e-5)
batches = bench.make_batches(4, bench.TRAIN_BATCH, 0, 1, dev)


def run(tag, steps=12):
    for i in range(3):
        dm._train_one_batch(*_b(i), sync=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        dm._train_one_batch(*_b(i), sync=False)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{tag}: {dt / steps * 1e3:.3f} ms/step (host issue time {th / steps * 1e3:.3f} ms)", flush=True)


def _b(i):
    x0, c2, c1 = batches[i % len(batches)]
    return x0, c2, c1


def _kw(f):
    def g(x0, c2, c1, sync=False):
        return f(x0, ms2_cond=c2, ms1_cond=c1, sync=sync)
    return g


dm._train_one_batch = _kw(dm._train_one_batch)
run("no process group")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
real = torch.distributed.all_reduce
torch.distributed.all_reduce = lambda *a, **k: None
run("process group, collective skipped")
torch.distributed.all_reduce = real
run("process group + all_reduce")
torch.distributed.destroy_process_group()
