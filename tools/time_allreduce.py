import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29512"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
g = torch.randn(128847, device="cuda")
for _ in range(5): dist.all_reduce(g)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): dist.all_reduce(g)
torch.cuda.synchronize()
print("all_reduce alone: %.1f us each" % ((time.perf_counter() - t0) / 100 * 1e6))
# interleaved with GPU work on the default stream
a = torch.randn(4096, 4096, device="cuda")
def work():
    for _ in range(20): a.mul_(1.0001)
for _ in range(3): work(); dist.all_reduce(g)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): work()
torch.cuda.synchronize(); tw = (time.perf_counter() - t0) / 50
t0 = time.perf_counter()
for _ in range(50): work(); dist.all_reduce(g)
torch.cuda.synchronize(); twa = (time.perf_counter() - t0) / 50
print("work %.1f us ; work+all_reduce %.1f us" % (tw * 1e6, twa * 1e6))
dist.destroy_process_group()
