#!/usr/bin/env python3
"""Time one k_level_fwd launch (pre = none, two blocks, identity residual) alone on the device.
usage: tools/time_level.py C n B [reps]      rows = B * 400;  env PRE, CP, CS: input stage (0 none, 1 down, 2 up, 3 k3), its input channels,
skip channels per block;  env PROBE=1 with DQ_HIP_LIB=<the -DDQ_LEVEL_PROBE variant>: median shader-clock intervals of the workgroups' first wave
Used to separate a launch's start-up (operand-image staging, instruction fetch) from its per-tile time: same C, growing B."""
import os
import sys
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "diffusion-deconvolution-dia-msms-data_amd"))
from dquartic import _native as N  # noqa: E402

C, n, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
RT = 400
dev = torch.device("cuda:0")
L = N.lib()
g = torch.Generator().manual_seed(0)
rows = B * RT
PRE, CP, CS = int(os.environ.get("PRE", 0)), int(os.environ.get("CP", C)), int(os.environ.get("CS", 0))
npar = L.dq_level_param_floats(PRE, C, CP, CS, 2)
params = (torch.randn(npar, generator=g) * 0.3).to(dev)
n_in = {0: n, 1: 2 * n, 2: max(n // 2, 1), 3: n}[PRE]
x = torch.randn(rows, CP if PRE else C, n_in, generator=g).to(dev)
sk0 = torch.randn(rows, CS, n, generator=g).to(dev) if CS else None
sk1 = torch.randn(rows, CS, n, generator=g).to(dev) if CS else None
temb = torch.randn(B, 16, generator=g).to(dev)
o0, o1 = torch.empty(rows, C, n, device=dev), torch.empty(rows, C, n, device=dev)
ws = torch.empty(2 * B * 2 * C + (0 if os.environ.get("NOIMG") else 8256), device=dev)  # (+ room for the operand image)


def run():
    N.check(L.dq_level_fwd(N.ptr(params), PRE, N.ptr(x), CP if PRE else C, N.ptr(sk0), N.ptr(sk1), CS, N.ptr(temb), N.ptr(o0), N.ptr(o1), C, 2, rows, n,
                           RT, N.ptr(ws), ws.numel(), N.stream_ptr()), "dq_level_fwd")


alt = None
if os.environ.get("ALT"):  # ALT="C,n": another instantiation launched between the timed calls (evicts the instruction cache)
    C2, n2 = map(int, os.environ["ALT"].split(","))
    p2 = (torch.randn(L.dq_level_param_floats(0, C2, C2, 0, 2), generator=g) * 0.3).to(dev)
    x2 = torch.randn(rows, C2, n2, generator=g).to(dev)
    q0, q1 = torch.empty_like(x2), torch.empty_like(x2)
    ws2 = torch.empty(2 * B * 2 * C2, device=dev)

    def alt():
        N.check(L.dq_level_fwd(N.ptr(p2), 0, N.ptr(x2), C2, None, None, 0, N.ptr(temb), N.ptr(q0), N.ptr(q1), C2, 2, rows, n2, RT, N.ptr(ws2),
                               ws2.numel(), N.stream_ptr()), "dq_level_fwd")

    for _ in range(5):
        alt()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        alt()
    e1.record()
    torch.cuda.synchronize()
    t_alt = e0.elapsed_time(e1) / reps * 1e3
    print(f"alt C={C2} n={n2}: {t_alt:.1f} us per call alone")
    _run = run

    def run():
        _run()
        alt()

for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
print(f"level_fwd C={C} n={n} B={B}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per call (incl. two ss-head launches)")

if os.environ.get("PROBE"):
    import ctypes
    import numpy as np
    lib = ctypes.CDLL(N.LIB_PATH)
    buf = np.zeros(4096 * 16, dtype=np.uint64)
    torch.cuda.synchronize()
    _run() if os.environ.get("ALT") else run()
    torch.cuda.synchronize()
    assert lib.dq_level_probe_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    st = buf.reshape(4096, 16).astype(np.int64)
    st = st[st[:, 0] > 0]
    names = ["stage weights", "stage params + barrier", "tile: loads + input stage", "block 0", "block 1", "(loop exit)"]
    print(f"{len(st)} workgroups; shader clocks (median / max):")
    for i, nm in enumerate(names):
        d = st[:, i + 1] - st[:, i]
        print(f"  {nm:28s} {int(np.median(d)):8d} {int(d.max()):8d}")
    if st[:, 7].max() > 0:
        for a_, b_, nm in [(4, 7, "block 1: conv1 over x"), (7, 8, "block 1: conv1 over skip"), (8, 9, "block 1: residual conv"), (9, 10, "block 1: bias, store u1, norm, act"),
                           (10, 11, "block 1: conv2"), (11, 5, "block 1: bias, norm, act, residual, stores")]:
            d = st[:, b_] - st[:, a_]
            print(f"  {nm:44s} {int(np.median(d)):8d} {int(d.max()):8d}")
    print(f"  total {int(np.median(st[:, 6] - st[:, 0]))} clocks; first start -> last end {int(st[:, 6].max() - st[:, 0].min())}")
