#!/usr/bin/env python3
"""Benchmark of the DDIM hot path on MI355X (contract: see the task description / DESIGN.md "Measurement").

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one optimiser step of the reference's ``_train_one_batch`` (zero_grad, train_step incl. drawing t and noise,
backward, clip 10, AdamW) on one batch of 32 synthetic MS2+MS1 windows (RT=400 x MZ=64) per GPU -- BASELINE.json configs[1]
("same model, 10k synthetic windows, batch=32, 1xMI355X"), computed in fp32 (the reference's precision; the bf16 in that
config line is not used).  Inputs are resident in HBM before the timed region.  ``value`` = windows/s over all ranks
(weak scaling: per-GPU batch fixed).  The same JSON line also carries the 50-step DDIM sampling throughput
(configs[3], batch 512/GPU), the roofline of the dominant kernel, and the CPU baseline (oracle on the host cores).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "diffusion-deconvolution-dia-msms-data_amd"))
sys.path.insert(0, REPO)

import numpy as np
import torch

RT, MZ = 400, 64
TRAIN_BATCH = 32
SAMPLE_BATCH = 512
SAMPLE_STEPS = 50
F32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: dense bf16 (v_mfma_f32_32x32x16_bf16 at 32 cycles)
HBM_PEAK_GBS = 8000.0         # same guide: HBM3E 8 TB/s spec (6.3 TB/s achievable with a float4 copy)
# algorithmic work per window at 400 x 64 (torch.utils.flop_counter on the reference, 2*MAC, conv/matmul only)
FLOPS_FWD, FLOPS_TRAIN = 2_549_672_128, 7_648_971_456
BYTES_SAMPLE_STEP = 308_800                                  # read x_t, mixture, MS1; write x_{t-1}
BYTES_TRAIN = lambda batch: 206_400 + 515_388 / batch       # read x0, mixture, MS1 + the batch-amortised gradient
# algorithmic FLOPs (2*MAC) of one LinearAttention forward per m/z row of n positions with C channels:
#   to_qkv 2*384*C*n + ctx 2*4*32*32*n + out 2*4*32*32*n + to_out 2*128*C*n      


def la_flops_fwd(C, n):
    return 2 * 384 * C * n + 2 * 2 * 4 * 32 * 32 * n + 2 * 128 * C * n


# LinearAttention layers of the default network, (C, n) per m/z row: 7 down levels (dim_in, n = 64 .. 1), 7 up levels (dim_out, n = 1 .. 64)
LA_LAYERS = [(4, 64), (4, 32), (8, 16), (8, 8), (12, 4), (12, 2), (16, 1), (16, 1), (16, 2), (12, 4), (12, 8), (8, 16), (8, 32), (4, 64)]


def executed_flops_per_window(mz=MZ, rt=RT):
    """FLOPs (2 x MAC, fp32-equivalent) the kernels EXECUTE per window: everything but LinearAttention runs the reference's association
    (SURVEY 8d counts), LinearAttention runs the re-associated form: per (row, head) 4 products of 2*32*C*n forward,
    12 backward, plus the C x C W2 products (1 forward, 3 backward); rows of one position take the closed form (W2 only).  The backward of
    the non-LinearAttention part is priced at 2 x its forward, as the SURVEY count does.  Returns (forward, train step)."""
    scale = mz // 64
    la_alg = sum(la_flops_fwd(C, n * scale) for C, n in LA_LAYERS) * rt
    la_fwd = sum(4 * ((4 * 2 * 32 * C * n * scale if n * scale > 1 else 0) + 2 * C * C * n * scale) for C, n in LA_LAYERS) * rt
    la_bwd = sum(4 * ((12 * 2 * 32 * C * n * scale if n * scale > 1 else 0) + 3 * 2 * C * C * n * scale) for C, n in LA_LAYERS) * rt
    if (mz, rt) == (MZ, RT):
        fwd_alg = FLOPS_FWD
    elif (mz, rt) == (256, 2000):
        fwd_alg = 51_667_451_072
    else:
        raise ValueError("no algorithmic FLOP count for this window shape")
    rest = fwd_alg - la_alg
    return rest + la_fwd, 3 * rest + la_fwd + la_bwd


def large_window_leg(device):
    """BASELINE configs[4] on one GPU: windows of 2000 RT x 256 m/z (downsample_dim 256: rows of 256 .. 4 positions, 64-channel
    bottleneck over 2000 positions), batch 8 per GPU, fp32, random-init weights, uniform random inputs: train step (the same
    _train_one_batch as the headline) and one DDIM sampling step (hipGraph replay), timed over a handful of steps."""
    from dquartic.model.model import DDIMDiffusionModel
    from dquartic.model.unet1d import UNet1d

    B, rt, mz = 8, 2000, 256
    torch.manual_seed(0)
    net = UNet1d(dim=4, channels=1, dim_mults=(1, 2, 2, 3, 3, 4, 4), conditional=True, init_cond_channels=1, attn_cond_channels=1,
                 tfer_dim_mult=620, downsample_dim=mz, simple=True).to(device)
    dm = DDIMDiffusionModel(model_class=net, num_timesteps=1000, beta_schedule_type="cosine", pred_type="eps", auto_normalize=True,
                            ms1_loss_weight=0.0, device=device)
    dm._set_optimizer(1e-5)
    x0, c2, c1 = torch.rand(B, rt, mz, device=device), torch.rand(B, rt, mz, device=device), torch.rand(B, rt, device=device)
    for _ in range(3):
        dm._train_one_batch(x0, ms2_cond=c2, ms1_cond=c1, sync=False)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        loss = dm._train_one_batch(x0, ms2_cond=c2, ms1_cond=c1, sync=False)
    torch.cuda.synchronize()
    t_train = (time.perf_counter() - t0) / n
    xT = torch.randn(B, rt, mz, device=device)
    dm.sample(xT, c2, c1, num_steps=2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dm.sample(xT, c2, c1, num_steps=n)
    torch.cuda.synchronize()
    t_step = (time.perf_counter() - t0) / n
    ex_f, ex_t = executed_flops_per_window(mz, rt)
    out = {"workload": "BASELINE configs[4]: windows 2000 RT x 256 m/z, batch 8 per GPU, default UNet1d with downsample_dim 256 (197,103 params), fp32",
           "batch": B, "params": int(sum(q.numel() for _, q in net.trainable_named())),
           "train": {"ms_per_step": round(t_train * 1e3, 3), "windows_per_s": round(B / t_train, 2), "loss": round(float(loss), 5),
                     "flop_frac": round(B / t_train * 155_002_129_088 / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                     "executed_flop_frac": round(B / t_train * ex_t / 1e12 / F32_MFMA_PEAK_TFLOPS, 4)},
           "sample": {"ms_per_step": round(t_step * 1e3, 3), "windows_per_s_50_steps": round(B / (t_step * SAMPLE_STEPS), 3),
                      "flop_frac": round(B / t_step * 51_667_451_072 / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                      "executed_flop_frac": round(B / t_step * ex_f / 1e12 / F32_MFMA_PEAK_TFLOPS, 4)}}
    del net, dm, x0, c2, c1, xT
    torch.cuda.empty_cache()
    return out


def build_model(device):
    from dquartic.model.model import DDIMDiffusionModel
    from dquartic.model.unet1d import UNet1d

    net = UNet1d(dim=4, channels=1, dim_mults=(1, 2, 2, 3, 3, 4, 4), conditional=True, init_cond_channels=1,
                 attn_cond_channels=1, tfer_dim_mult=620, downsample_dim=MZ, simple=True).to(device)
    dm = DDIMDiffusionModel(model_class=net, num_timesteps=1000, beta_schedule_type="cosine", pred_type="eps",
                            auto_normalize=True, ms1_loss_weight=0.0, device=device)
    return net, dm


def make_batches(n_batches, batch, rank, world, device):
    """(x0, ms2_cond, ms1_cond) device batches from the synthetic pair dataset."""
    from dquartic.utils.synthetic import SyntheticDIAMSDataset

    ds = SyntheticDIAMSDataset(n_windows=64 * world, RT=RT, MZ=MZ, rank=rank, world=world, seed=0)
    out = []
    for _ in range(n_batches):
        items = [ds[0] for _ in range(batch)]
        a, m1, b, _ = (torch.stack([it[k] for it in items]) for k in range(4))
        out.append((a.to(device), (0.5 * a + 0.5 * b).to(device), m1.to(device)))
    return out


def time_kernel(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters  # seconds per launch


def pmc_traffic(kernel_prefix, section="kernels"):
    """HBM bytes per launch of a kernel from the committed counter passes (profiles/pmc_linattn.json, written by
    tools/pmc_passes.sh on the GPU box) -- only if they were taken on THIS build of the native sources; a stale file gives null."""
    from dquartic import _native as N

    path = os.path.join(REPO, "profiles", "pmc_linattn.json")
    if not os.path.exists(path):
        return None, "no profiles/pmc_linattn.json"
    with open(path) as fh:
        pmc = json.load(fh)
    if pmc.get("build_id") != N.build_id():
        return None, f"profiles/pmc_linattn.json is from build {pmc.get('build_id')}, this build is {N.build_id()}: stale, not quoted"
    for name, v in pmc.get(section, {}).items():
        if name.startswith(kernel_prefix):
            return float(v["hbm_bytes"]), f"profiles/pmc_linattn.json (2 x FETCH_SIZE + WRITE_SIZE, build {pmc['build_id']})"
    return None, f"{kernel_prefix} not in profiles/pmc_linattn.json"


def pmc_sq(kernel_prefix, launch_seconds, section="sq"):
    """Matrix-pipe busy fraction of a kernel from the SQ counter pass of the same file (stale file: null): SQ_VALU_MFMA_BUSY_CYCLES (summed
    over the SIMDs) / (launch duration x 2.4 GHz x 1,024 SIMDs)."""
    from dquartic import _native as N

    path = os.path.join(REPO, "profiles", "pmc_linattn.json")
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        pmc = json.load(fh)
    if pmc.get("build_id") != N.build_id():
        return None
    for name, v in pmc.get(section, {}).items():
        if name.startswith(kernel_prefix) and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            return round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (launch_seconds * 2.4e9 * 1024), 4)
    return None


def roofline_hbm_kernels(device):
    """The top HBM-bound kernel of each leg, alone on the device, timed with HIP events on the launch stream:
      train    -- k_res_bwd_wg<4, true> (+ its slot reduce): backward of an up-path ResnetBlock of level 0 (cat(4, 4) -> 4 channels, rows of 64
                  positions, 12,800 rows) INCLUDING its weight gradients; algorithmic bytes = read d out, u1, u2, x (8 channels), write d x
                  (8 channels) = 28 channel planes;
      sampling -- k_level_fwd<4, 0, 4>: the two ResnetBlocks of down level 0 at batch 512 (204,800 rows of 64 positions, 4 channels) in one
                  launch; algorithmic bytes = read x, write both block outputs (the skip and the LinearAttention input) = 12 channel planes.
    `traffic` = HBM bytes per launch from the counter passes of THIS build (profiles/pmc_linattn.json; FETCH_SIZE scaled by the factor
    measured on a kernel of the same access width, k_rmsnorm_fwd) or null."""
    from dquartic import _native as N

    L = N.lib()
    out = {}
    g = torch.Generator(device="cpu").manual_seed(0)
    # ---- train: ResnetBlock backward with fused weight gradients
    cin, cout, n, rows, B = 8, 4, 64, TRAIN_BATCH * RT, TRAIN_BATCH
    nparam = 2 * cout * 16 + 2 * cout + cout * cin * 3 + 2 * cout + cout * cout * 3 + 2 * cout + cout * cin + cout
    flat = (torch.randn(nparam, generator=g) * 0.3).to(device)
    xA, xB = torch.randn(rows, cout, n, generator=g).to(device), torch.randn(rows, cin - cout, n, generator=g).to(device)
    temb = torch.randn(B, 16, generator=g).to(device)
    nws = L.dq_resblock_workspace_floats(cin, cout, rows, n, RT)
    ws = torch.zeros(nws, device=device)
    outb = torch.empty(rows, cout, n, device=device)
    N.check(L.dq_resblock_fwd(N.ptr(flat), N.ptr(xA), cout, N.ptr(xB), cin - cout, N.ptr(temb), N.ptr(outb), cout, rows, n, RT, 1, N.ptr(ws), nws,
                              N.stream_ptr()), "dq_resblock_fwd")
    off = L.dq_resblock_dout_offset(cin, cout, rows, n, RT)
    ws[off:off + rows * cout * n].copy_(torch.randn(rows * cout * n, generator=g).to(device))
    dA, dB, grads = torch.empty_like(xA), torch.empty_like(xB), torch.zeros_like(flat)

    def bwd():
        N.check(L.dq_resblock_bwd(N.ptr(flat), N.ptr(xA), cout, N.ptr(xB), cin - cout, None, N.ptr(dA), N.ptr(dB), N.ptr(grads), None, cout, rows, n, RT,
                                  N.ptr(ws), nws, N.stream_ptr()), "dq_resblock_bwd")

    t = time_kernel(bwd)
    by = 28 * n * 4 * rows
    tr, src = pmc_traffic("k_res_bwd_wg<4, true>")
    out["train"] = {"bound": "hbm", "kernel": "k_res_bwd_wg<4,true> (+ k_res_wg_reduce): ResnetBlock backward incl. its weight gradients, cat(4,4)->4 channels, n 64",
                    "rows": rows, "launch_us": round(t * 1e6, 2), "achieved": round(by / t / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(by / t / 1e9 / HBM_PEAK_GBS, 4), "bytes_per_launch": by, "traffic": tr, "traffic_source": src}
    del ws, xA, xB, dA, dB, outb
    # ---- sampling: the level kernel on down level 0 at the sampling batch
    C, rows = 4, SAMPLE_BATCH * RT
    npar = L.dq_level_param_floats(0, C, C, 0, 2)
    params = (torch.randn(npar, generator=g) * 0.3).to(device)
    x = torch.randn(rows, C, n, generator=g).to(device)
    temb = torch.randn(SAMPLE_BATCH, 16, generator=g).to(device)
    o0, o1 = torch.empty_like(x), torch.empty_like(x)
    wsl = torch.empty(2 * SAMPLE_BATCH * 2 * C + 8256, device=device)  # (+ room for the prepared operand image: the network path's staging)

    def lvl():
        N.check(L.dq_level_fwd(N.ptr(params), 0, N.ptr(x), C, None, None, 0, N.ptr(temb), N.ptr(o0), N.ptr(o1), C, 2, rows, n, RT, N.ptr(wsl), wsl.numel(),
                               N.stream_ptr()), "dq_level_fwd")

    t = time_kernel(lvl)  # (the two 3-us k_ss_heads launches of the entry point ride along)
    by = 12 * n * 4 * rows
    tr, src = pmc_traffic("k_level_fwd<4, 0, 4,")
    out["sample"] = {"bound": "hbm", "kernel": "k_level_fwd<4,0,4>: both ResnetBlocks of down level 0 in one launch, batch 512", "rows": rows,
                     "launch_us": round(t * 1e6, 2), "achieved": round(by / t / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(by / t / 1e9 / HBM_PEAK_GBS, 4), "bytes_per_launch": by, "traffic": tr, "traffic_source": src}
    return out


def roofline_linattn(device, rows=TRAIN_BATCH * RT, with_bwd=True):
    """Dominant kernels: k_linattn_bwd<4,64> (train step; level-0 LinearAttention backward over 32*400 rows) and
    k_linattn_fwd<4,64> (sampling leg at rows = 512*400).  Timed live with HIP events on the launch stream; `achieved` prices the
    launch at the ALGORITHMIC FLOPs of the reference's association (2x the forward's for the backward), `executed_frac` at the
    FLOPs the re-associated kernels really issue, `hbm_frac` at the algorithmic bytes against the HBM peak."""
    from dquartic import _native as N

    C, n = 4, 64
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(rows, C, n, generator=g).to(device)
    dy = torch.randn(rows, C, n, generator=g).to(device)
    w = (torch.randn(384, C, generator=g) * 0.4).to(device)
    wo = (torch.randn(C, 128, generator=g) * 0.2).to(device)
    bo, g1, g2 = torch.zeros(C, device=device), torch.ones(C, device=device), torch.ones(C, device=device)
    y, ypre, dx = torch.empty_like(x), torch.empty_like(x), torch.zeros_like(x)
    dw, dwo, dbo, dg1, dg2 = (torch.zeros_like(t) for t in (w, wo, bo, g1, g2))
    scratch = torch.empty(2 * x.numel() + 2048 * 512 * C, device=device)
    L = N.lib()
    sampling = rows != TRAIN_BATCH * RT   # the sampling leg's launch (204,800 rows): its own counter passes (tools/pmc_passes.sh, "sample" mode)

    # the network's code path: derived weights + operand images prepared once per parameter state (dq_linattn_prepare), every launch copies them
    prep = torch.zeros(L.dq_linattn_prep_floats(), device=device)
    N.check(L.dq_linattn_prepare(N.ptr(w), N.ptr(wo), N.ptr(g1), C, N.ptr(prep), N.stream_ptr()), "dq_linattn_prepare")

    def fwd():
        N.check(L.dq_linattn_fwd_prepared(N.ptr(x), N.ptr(y), N.ptr(ypre), N.ptr(w), N.ptr(wo), N.ptr(bo), N.ptr(g1), N.ptr(g2), N.ptr(prep), C, rows, n,
                                          N.stream_ptr()), "dq_linattn_fwd_prepared")

    def bwd():
        N.check(L.dq_linattn_bwd(N.ptr(x), N.ptr(ypre), N.ptr(dy), N.ptr(dx), N.ptr(w), N.ptr(wo), N.ptr(bo), N.ptr(g1), N.ptr(g2),
                                 N.ptr(dw), N.ptr(dwo), N.ptr(dbo), N.ptr(dg1), N.ptr(dg2), N.ptr(scratch), C, rows, n,
                                 N.stream_ptr()), "dq_linattn_bwd")

    t_f = time_kernel(fwd)
    fl_f = la_flops_fwd(C, n) * rows
    ex_f = 4 * (4 * 2 * 32 * C * n + 2 * C * C * n) * rows
    by_f = 8 * C * n * rows  # x in, y out (inference; training adds the 4*C*n pre-norm save)
    tr_f, src_f = pmc_traffic("k_linattn_fwd<4, 64,", "sample_kernels" if sampling else "kernels")
    fwd_obj = {"bound": "mfma", "kernel": "k_linattn_fwd<4,64> (K = C projections: split-bf16 on v_mfma_f32_32x32x16_bf16, fp32-exact to ~2^-23)", "rows": rows,
               "launch_us": round(t_f * 1e6, 2),
               "achieved": round(ex_f / t_f / 1e12, 3), "algorithmic_achieved": round(fl_f / t_f / 1e12, 3),
               "algorithmic_frac": round(fl_f / t_f / 1e12 / F32_MFMA_PEAK_TFLOPS, 4), "mfma_busy": pmc_sq("k_linattn_fwd<4, 64,", t_f, "sample_sq" if sampling else "sq"),
               "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
               # the re-association removed work: the algorithmic rate can exceed the pipe's peak, so `frac` is stated on the FLOPs
               # the kernel executes; the algorithmic figure stays in `achieved`
               "frac": round(ex_f / t_f / 1e12 / F32_MFMA_PEAK_TFLOPS, 4), "frac_basis": "executed FLOPs",
               "flops_per_launch": fl_f, "executed_flops_per_launch": ex_f,
               "hbm_frac": round(by_f / t_f / 1e9 / HBM_PEAK_GBS, 4), "bytes_per_launch": by_f,
               "traffic": tr_f, "traffic_source": src_f}
    if not with_bwd:
        return fwd_obj
    t_b = time_kernel(bwd)  # the fused backward launch + the ordered slot reduce (~3 % of it)
    ach_b = 2 * fl_f / t_b / 1e12
    # `achieved` prices the launch at the ALGORITHMIC FLOPs of the reference's formulation.  The kernels
    # re-associate the value path (M = K xh^T, P = M^T Q, W2 = Wo Wv) and EXECUTE fewer: per row and head
    # 4 (forward) resp. 12 (backward) products of 2*32*C*n FLOP plus 1 resp. 3 of 2*C*C*n -- reported next to it.
    ex_b = 4 * (12 * 2 * 32 * C * n + 3 * 2 * C * C * n) * rows
    by_b = 20 * C * n * rows  # read x, ypre, dy, (dx) ; write dx
    tr_b, src_b = pmc_traffic("k_linattn_bwd<4, 64>")
    # ONE basis under `frac` for both LinearAttention kernels: the FLOPs the kernel EXECUTES (the re-association removed work, so the
    # reference-count figure is not a fraction of the pipe: it stays under `algorithmic_frac`); `mfma_busy` is the matrix pipe's measured
    # busy share (SQ pass of the same build)
    return {"bound": "mfma", "kernel": "k_linattn_bwd<4,64>", "achieved": round(ex_b / t_b / 1e12, 3), "peak": F32_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": round(ex_b / t_b / 1e12 / F32_MFMA_PEAK_TFLOPS, 4), "frac_basis": "executed FLOPs",
            "algorithmic_achieved": round(ach_b, 3), "algorithmic_frac": round(ach_b / F32_MFMA_PEAK_TFLOPS, 4),
            "mfma_busy": pmc_sq("k_linattn_bwd<4, 64>", t_b),
            "traffic": tr_b, "traffic_source": src_b,
            "launch_us": round(t_b * 1e6, 2), "flops_per_launch": 2 * fl_f,
            "executed_flops_per_launch": ex_b, "executed_frac": round(ex_b / t_b / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
            "hbm_frac": round(by_b / t_b / 1e9 / HBM_PEAK_GBS, 4), "bytes_per_launch": by_b,
            "fwd_kernel": fwd_obj}


def batch_formation(device):
    """forming one training batch (gather + per-pair min-max + mixture, dq_pair_batch) from a dataset that is
    resident in HBM; HBM-bound.  Algorithmic bytes per pair = read 2 windows + write ms2_1, ms2_2, ms2_cond (5 x RT*MZ*4 B)
    + MS1 (read 2, write 2 rows).  4,096 resident windows (420 MB) so the gathers do not all hit the Infinity Cache."""
    from dquartic import _native as N

    n, B = 4096, TRAIN_BATCH
    g = torch.Generator(device="cpu").manual_seed(0)
    ms2 = torch.rand(n, RT, MZ, device=device)
    ms1 = torch.rand(n, RT, device=device)
    idx = torch.randint(0, n, (64, 2 * B), generator=g).to(device)
    new = lambda *s: torch.empty(s, device=device)
    a, b, c, m1, m2 = new(B, RT, MZ), new(B, RT, MZ), new(B, RT, MZ), new(B, RT), new(B, RT)
    L = N.lib()
    sc = torch.empty(L.dq_pair_batch_scratch_bytes(B) // 4, device=device)
    k = [0]

    def form():
        k[0] = (k[0] + 1) % 64
        N.check(L.dq_pair_batch(N.ptr(ms2), N.ptr(ms1), n, N.ptr(idx[k[0]]), B, RT, MZ, RT, 0.5, 0.5, N.ptr(a), N.ptr(m1), N.ptr(b),
                                N.ptr(m2), N.ptr(c), N.ptr(sc), sc.numel() * 4, N.stream_ptr()), "dq_pair_batch")

    t = time_kernel(form, iters=50)
    nbytes = B * (5 * RT * MZ * 4 + 4 * RT * 4)
    return {"kernel": "k_pair_minmax + k_pair_mix (dq_pair_batch)", "pairs_per_s": round(B / t, 1), "call_us": round(t * 1e6, 2),
            "bound": "hbm", "achieved": round(nbytes / t / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(nbytes / t / 8e12, 4),
            "bytes_per_call": nbytes}


def cpu_baseline(net):
    """The oracle (CPU restatement of the reference, kind 'port') on this box's host cores: per-sample (B = 1) train steps
    -- the only batch size the reference runs at -- incl. clip + AdamW, on a bounded sample of the same workload."""
    from oracle import dq_oracle as O

    # the GPU box gives one GPU a 16-CPU share; os.cpu_count() reports the whole host and oversubscribing it stalls OpenMP
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("DQ_CPU_BASELINE_THREADS", "16"))))
    torch.set_num_threads(cores)
    params = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    keys = O.trainable_keys(params)
    for k in keys:
        params[k].requires_grad_(True)
    opt = torch.optim.AdamW([params[k] for k in keys], lr=1e-5)
    d = O.Diffusion(params, O.UNetConfig(downsample_dim=MZ))
    batches = make_batches(1, 8, 0, 1, "cpu")[0]
    n_steps, t_train = 0, 0.0
    t_begin = time.perf_counter()
    for i in range(200):
        if i >= 8 and time.perf_counter() - t_begin > 12.0:  # bounded sample: ~12 s of CPU work
            break
        x0, c2, c1 = (v[i % 8:i % 8 + 1] for v in batches)
        t0 = time.perf_counter()
        opt.zero_grad()
        t = torch.randint(0, 1000, (1,))
        nz = torch.randn_like(x0)
        loss, _ = d.train_loss(x0, c2, c1, t, nz)
        loss.backward()
        torch.nn.utils.clip_grad_norm_([params[k] for k in keys], 10.0)
        opt.step()
        dt = time.perf_counter() - t0
        if i < 3 or i % 10 == 0:
            log(f"cpu baseline step {i}: {dt:.2f} s")
        if i >= 2:  # 2 warm-up steps
            n_steps += 1
            t_train += dt
    # sampling: 1 window, 5 of the 50 steps timed, extrapolated
    with torch.no_grad():
        x0, c2, c1 = (v[:1] for v in batches)
        t0 = time.perf_counter()
        d.sample(torch.randn_like(x0), c2, c1, 10)
        t_s = (time.perf_counter() - t0) * (SAMPLE_STEPS / 10)
    return {"value": round(n_steps / t_train, 4), "unit": "MS2 windows/s (train step)", "cores": cores, "kind": "port",
            "sample": f"{n_steps} timed B=1 train steps (after 2 warm-up, ~12 s) of the same network/shape; sampling: 1 window x 10 steps, x5",
            "sample_windows_per_s": round(1.0 / t_s, 5)}


# ---------------------------------------------------------------------------------------------- CustomTransformer leg
TFM_CFG = dict(input_dim=40000, hidden_dim=1024, num_heads=8, num_layers=8)  # reference dquartic_train_config.json "CustomTransformer"
TFM_RT = 34  # reference `generate-data-slices --window-size` default: the RT rows of one window


def transformer_leg(device, with_cpu):
    """The reference's other noise predictor at its own configuration (191,126,592 parameters, windows of 34 RT x 40000 m/z): train
    step (q_sample, forward, MSE, backward, clip 10, AdamW -- all native) at batch 1 (the batch size of the reference's wandb
    runs: ~18.4 steps/s on unstated hardware) and batch 32, the matrix-core roofline of the dominant kernel (the fp32
    GEMM, measured alone on the output projection's shape) and the oracle on the host cores."""
    from dquartic import _native as N
    from dquartic.model.building_blocks import CustomTransformer, DDIMTransformerAdapter
    from dquartic.model.model import DDIMDiffusionModel

    torch.manual_seed(0)
    D, H = TFM_CFG["input_dim"], TFM_CFG["hidden_dim"]
    net = DDIMTransformerAdapter(CustomTransformer(**TFM_CFG)).to(device)
    dm = DDIMDiffusionModel(model_class=net, num_timesteps=1000, beta_schedule_type="cosine", pred_type="eps", auto_normalize=True,
                            ms1_loss_weight=0.0, device=device)
    dm._set_optimizer(1e-5)
    out = {"config": {"workload": "CustomTransformer(40000, 1024, 8 heads, 8 layers), windows 34 RT x 40000 m/z, MS1 (34,) as x_cond, fp32",
                      "params": int(net.transformer.flat_params.numel())}}

    def train_rates(tag):
        res = {}
        for B in (1, 32):
            x0, c2, c1 = torch.rand(B, TFM_RT, D, device=device), torch.rand(B, TFM_RT, D, device=device), torch.rand(B, TFM_RT, device=device)
            for _ in range(2):
                dm._train_one_batch(x0, ms2_cond=c2, ms1_cond=c1, sync=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            steps = 10
            for _ in range(steps):
                loss = dm._train_one_batch(x0, ms2_cond=c2, ms1_cond=c1, sync=False)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            res[f"train_b{B}"] = {"value": round(B / dt, 2), "unit": "MS2 windows/s", "ms_per_step": round(dt * 1e3, 3), "loss": round(float(loss), 5)}
            del x0, c2, c1
        return res

    out.update(train_rates("fp32"))
    # the same steps with the dense products in the bf16x3 precision mode (three bf16 matrix-core passes over hi/lo-split operands;
    # NOT the fp32 arithmetic the headline is stated in: its own tolerance, tests/test_tfm.py)
    net.transformer.set_precision("bf16x3")
    out["bf16x3"] = {"dtype": "bf16x3 (split-bf16, 3 MFMA passes, fp32 accumulate)", **train_rates("bf16x3")}
    net.transformer.set_precision("fp32")
    # roofline of the GEMM on the output projection's forward shape at batch 32
    lib = N.lib()
    M, Nn, K = 32 * TFM_RT, D, H
    A, Bm, C = torch.randn(M, K, device=device), torch.randn(Nn, K, device=device), torch.empty(M, Nn, device=device)
    scr = torch.empty(max(int(lib.dq_gemm_scratch_floats(M, Nn, K)), 4), device=device)
    sec = time_kernel(lambda: N.check(lib.dq_gemm(N.ptr(A), N.ptr(Bm), N.ptr(C), None, M, Nn, K, K, K, Nn, 1, 1, 0, 0, N.ptr(scr), scr.numel(),
                                                  N.stream_ptr()), "dq_gemm"), iters=10)
    fl = 2.0 * M * Nn * K
    out["roofline"] = {"bound": "mfma", "kernel": "k_gemm<A k-major, B k-major, 128x128x32> on (1088 x 1024) x (1024 x 40000)",
                       "achieved": round(fl / sec / 1e12, 2), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                       "frac": round(fl / sec / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                       "traffic": None,  # not re-measured on this build (round 1, profiles/r01_pmc_gemm.md: 704.1 MB per launch)
                       "flops_per_launch": fl, "us_per_launch": round(sec * 1e6, 1)}
    sec3 = time_kernel(lambda: N.check(lib.dq_gemm_bf16x3(N.ptr(A), N.ptr(Bm), N.ptr(C), None, M, Nn, K, K, K, Nn, 1, 1, 0, 0, N.ptr(scr), scr.numel(),
                                                          N.stream_ptr()), "dq_gemm_bf16x3"), iters=10)
    out["bf16x3"]["roofline"] = {"bound": "mfma", "kernel": "k_gemm_s3<A k-major, B k-major, 128x128x32> on the same product",
                                 "achieved": round(fl / sec3 / 1e12, 2), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 # three bf16 passes per product: executed FLOPs = 3 x algorithmic
                                 "frac": round(3 * fl / sec3 / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4), "frac_basis": "executed FLOPs (3 passes)",
                                 "flops_per_launch": fl, "us_per_launch": round(sec3 * 1e6, 1), "speedup_vs_fp32": round(sec / sec3, 2)}
    del net, dm, A, Bm, C
    torch.cuda.empty_cache()
    if with_cpu:
        out["cpu_baseline"] = transformer_cpu_baseline()
    return out


def transformer_cpu_baseline():
    """The transformer oracle (kind 'port') on the host cores: B = 1 train steps (forward, MSE, backward, clip, AdamW) at the full
    configuration; bounded to a few steps (each is ~37 GFLOP of dense work plus a 191 M-parameter optimiser update)."""
    from oracle import dq_oracle_tfm as OT

    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("DQ_CPU_BASELINE_THREADS", "16"))))
    torch.set_num_threads(cores)
    D, H = TFM_CFG["input_dim"], TFM_CFG["hidden_dim"]
    params = OT.init_params(D, H, TFM_CFG["num_layers"], seed=0)
    plist = [p.requires_grad_(True) for p in params.values()]
    opt = torch.optim.AdamW(plist, lr=1e-5)
    x0, c1 = torch.rand(1, TFM_RT, D), torch.rand(1, TFM_RT)
    n, tt, t_begin = 0, 0.0, time.perf_counter()
    for i in range(12):
        if i >= 3 and time.perf_counter() - t_begin > 15.0:
            break
        t0 = time.perf_counter()
        opt.zero_grad()
        t = torch.randint(0, 1000, (1,))
        nz = torch.randn_like(x0)
        ab = torch.rand(())  # any alpha_bar: the arithmetic does not depend on its value
        x_t = torch.sqrt(ab) * (2 * x0 - 1) + torch.sqrt(1 - ab) * nz
        loss = torch.nn.functional.mse_loss(OT.forward(params, x_t, t, 2 * c1 - 1, TFM_CFG["num_heads"]), nz)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(plist, 10.0)
        opt.step()
        dt = time.perf_counter() - t0
        log(f"transformer cpu baseline step {i}: {dt:.2f} s")
        if i >= 1:
            n += 1
            tt += dt
    return {"value": round(n / tt, 4), "unit": "MS2 windows/s (train step, B=1)", "cores": cores, "kind": "port",
            "sample": f"{n} timed B=1 train steps (after 1 warm-up) of the same network / window shape"}


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    # stdout carries exactly ONE line, the JSON: anything libraries print there (RCCL's version banner at process-group creation,
    # for one) is sent to stderr by pointing fd 1 at fd 2 for the duration of the run
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)  # ~1.1 s of timed train steps: long enough for an outside observer to see the GPU busy
    ap.add_argument("--warmup", type=int, default=20)  # a fresh box needs ~50 ms of work before its clocks settle
    ap.add_argument("--no-sample", action="store_true", help="skip the sampling leg")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-transformer", action="store_true", help="skip the CustomTransformer leg (rank 0, single-GPU runs only)")
    ap.add_argument("--no-large-window", action="store_true", help="skip the configs[4] leg (2000 x 256 windows, batch 8; single-GPU runs only)")
    ap.add_argument("--train-only", action="store_true", help="only the train leg (clean per-kernel profiles of the train step)")
    ap.add_argument("--sample-batch", type=int, default=SAMPLE_BATCH)
    args = ap.parse_args()
    if args.train_only:
        args.no_sample = args.no_cpu = args.no_transformer = args.no_large_window = True

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)  # under torchrun, also with one rank
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
    elif args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run for --gpus > 1 (one process per GPU)")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    torch.manual_seed(rank)

    net, dm = build_model(device)
    dm._set_optimizer(1e-5)
    dm._sync_replicas()  # data-parallel runs: every rank starts from rank 0's weights (the ranks seed their RNGs differently)
    log("building synthetic batches")
    batches = make_batches(4, TRAIN_BATCH, rank, world, device)
    log("train leg")

    def barrier():
        torch.cuda.synchronize()
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # ------------------------------------------------------------------ train leg
    for i in range(args.warmup):
        x0, c2, c1 = batches[i % len(batches)]
        dm._train_one_batch(x0, ms2_cond=c2, ms1_cond=c1, sync=False)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        x0, c2, c1 = batches[i % len(batches)]
        loss = dm._train_one_batch(x0, ms2_cond=c2, ms1_cond=c1, sync=False)
    # the host's share (all launches queued, nothing waited for); with many steps it includes the back-pressure of a full queue,
    # i.e. it approaches ms_per_step from below -- the host's own cost is what a short run shows (~2.3 ms at --steps 20)
    host_issue_ms = (time.perf_counter() - t0) * 1e3 / max(1, args.steps)
    barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], device=device, dtype=torch.float64)
    if dist_on:
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
    dt = float(tt)
    train_wps = world * TRAIN_BATCH * args.steps / dt
    last_loss = float(loss)

    # ------------------------------------------------------------------ the same step over >= 5 s (clock settle; whatever --steps was)
    sustained = None
    if not args.train_only:
        n_sus = max(args.steps, int(5.0 / (dt / args.steps)) + 1)
        barrier()
        t0 = time.perf_counter()
        for i in range(n_sus):
            x0, c2, c1 = batches[i % len(batches)]
            dm._train_one_batch(x0, ms2_cond=c2, ms1_cond=c1, sync=False)
        barrier()
        ts_ = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
        if dist_on:
            torch.distributed.all_reduce(ts_, op=torch.distributed.ReduceOp.MAX)
        sustained = {"value": round(world * TRAIN_BATCH * n_sus / float(ts_), 2), "unit": "windows/s", "steps": n_sus, "seconds": round(float(ts_), 3),
                     "ms_per_step": round(float(ts_) / n_sus * 1e3, 3)}
    # ------------------------------------------------------------------ small batches: the reference trains at batch_size 1
    # (dquartic_train_config.json:12), BASELINE configs[0] at 4 -- a dependency chain of ~250 short launches there
    # Single-process runs only: with world > 1 every _train_one_batch issues the flat gradient all-reduce, and a collective that rank 0
    # alone issues between two barriers never finds its peers.  The leg describes one GPU anyway.
    small = None
    if rank == 0 and world == 1 and not args.train_only:
        small = {}
        for bsz in (1, 4):
            x0, c2, c1 = (v[:bsz].contiguous() for v in batches[0])
            for _ in range(5):
                dm._train_one_batch(x0, ms2_cond=c2, ms1_cond=c1, sync=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                dm._train_one_batch(x0, ms2_cond=c2, ms1_cond=c1, sync=False)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 50 * 1e3
            small[f"b{bsz}"] = {"ms_per_step": round(ms, 3), "windows_per_s": round(bsz / ms * 1e3, 1)}
            # (The same step replayed from ONE captured hipGraph -- ModelInterface.enable_train_graph, bit-identical to the eager step -- is NOT
            # reported: a graph has to run the step as a single chain (1.48 ms at batch 1 against 1.25 eager on two queues), and with the
            # fork / join captured as a second branch a replay takes 5.9 ms: tools/graph_side_probe.py)
    if dist_on:
        torch.distributed.barrier()

    # ------------------------------------------------------------------ sampling leg (no collective: batch shards)
    log(f"train: {train_wps:.1f} windows/s; sampling leg")
    sample = None
    if not args.no_sample:
        B = args.sample_batch
        g = torch.Generator(device="cpu").manual_seed(rank)
        xT = torch.randn(B, RT, MZ, generator=g).to(device)
        reps = (B + TRAIN_BATCH - 1) // TRAIN_BATCH
        c2 = torch.cat([batches[i % len(batches)][1] for i in range(reps)])[:B].contiguous()
        c1 = torch.cat([batches[i % len(batches)][2] for i in range(reps)])[:B].contiguous()
        dm.sample(xT, c2, c1, num_steps=2)  # warm-up at the timed batch size: the hipGraph of one step is captured (and cached) here
        barrier()
        t0 = time.perf_counter()
        dm.sample(xT, c2, c1, num_steps=SAMPLE_STEPS)
        barrier()
        ds = time.perf_counter() - t0
        ts = torch.tensor([ds], device=device, dtype=torch.float64)
        if dist_on:
            torch.distributed.all_reduce(ts, op=torch.distributed.ReduceOp.MAX)
        swps = world * B / float(ts)
        sample = {"metric": "MS2 windows/s (50-step DDIM sample)", "value": round(swps, 2), "batch_per_gpu": B,
                  "steps": SAMPLE_STEPS, "seconds": round(float(ts), 4),
                  # whole-leg fractions per GPU: algorithmic FLOPs (50 forwards per window) against the f32 matrix peak, compulsory
                  # bytes (x_t, mixture, MS1 in; x_{t-1} out, per step) against the HBM peak
                  "whole_leg": {"flop_frac": round(swps / world * SAMPLE_STEPS * FLOPS_FWD / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                                "executed_flop_frac": round(swps / world * SAMPLE_STEPS * executed_flops_per_window()[0] / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                                "hbm_frac": round(swps / world * SAMPLE_STEPS * BYTES_SAMPLE_STEP / 1e9 / HBM_PEAK_GBS, 6)}}
        if rank == 0:
            log("sampling roofline leg")
            sample["roofline"] = roofline_linattn(device, rows=B * RT, with_bwd=False)

    log("roofline leg")
    roof = roofline_linattn(device) if (rank == 0 and not args.train_only) else None
    form = batch_formation(device) if (rank == 0 and not args.train_only) else None
    hbm = roofline_hbm_kernels(device) if (rank == 0 and not args.train_only) else None
    large = None
    if rank == 0 and world == 1 and not args.no_large_window:
        log("large-window leg (configs[4])")
        large = large_window_leg(device)
    log("cpu baseline leg")
    cpu = cpu_baseline(net) if (rank == 0 and world == 1 and not args.no_cpu) else None
    tfm = None
    if rank == 0 and world == 1 and not args.no_transformer:
        log("transformer leg")
        del net, dm
        torch.cuda.empty_cache()
        tfm = transformer_leg(device, with_cpu=not args.no_cpu)

    if rank == 0:
        out = {
            "metric": "MS2 windows/s (train step: zero_grad + train_step + backward + clip 10 + AdamW)",
            "value": round(train_wps, 2), "unit": "windows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: default UNet1d (dim 4, mults 1,2,2,3,3,4,4; 128,847 params), synthetic MS2+MS1 "
                                   "windows 400 RT x 64 m/z, batch 32 per GPU, fp32", "global_batch": world * TRAIN_BATCH,
                       "window": [RT, MZ], "parallelism": f"dp{world}"},
            "last_loss": round(last_loss, 6),
            "host_issue_ms_per_step": round(host_issue_ms, 3),
            "build_id": __import__("dquartic._native", fromlist=["x"]).build_id(),
            # whole-step fractions per GPU: algorithmic FLOPs of fwd + bwd against the f32 matrix peak; compulsory bytes against HBM
            # (`flop_frac`: the reference's algorithmic count; `executed_flop_frac`: the FLOPs the kernels execute, executed_flops_per_window)
            "whole_step": {"flop_frac": round(train_wps / world * FLOPS_TRAIN / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                           "executed_flop_frac": round(train_wps / world * executed_flops_per_window()[1] / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                           "hbm_frac": round(train_wps / world * BYTES_TRAIN(TRAIN_BATCH) / 1e9 / HBM_PEAK_GBS, 6)},
            "sustained": sustained, "small_batch": small,
            "sample": sample, "roofline": roof, "roofline_hbm": hbm, "cpu_baseline": cpu, "batch_formation": form, "large_window": large,
            "transformer": tfm,
        }
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist_on:
        torch.distributed.barrier()  # rank 0 ran the extra single-rank legs: leave together
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
