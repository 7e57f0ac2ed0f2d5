"""CPU: the NEWEST committed benchmark line (profiles/bench_rNN.json, written by `python bench.py` on an MI355X) carries every field the
benchmark contract names, with consistent values -- including the objects later rounds added (sampling leg with its own roofline,
large-window leg, whole-step executed-FLOP fraction, small-batch legs); and bench.py's argument surface is the contract's."""
import glob
import json
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _newest_bench_line():
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "bench_r[0-9][0-9].json")))
    assert files, "no profiles/bench_rNN.json committed"
    return files[-1], open(files[-1]).read().strip()


def _check_roofline(r, bound=None):
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and (bound is None or r["bound"] == bound)
    assert r["unit"] == ("GB/s" if r["bound"] == "hbm" else "TFLOP/s")
    assert r["peak"] == (8000.0 if r["bound"] == "hbm" else 157.3)
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    assert r["traffic"] is None or r["traffic"] > 0
    if "launch_us" in r and "executed_flops_per_launch" in r:  # achieved = executed FLOPs per launch / HIP-event launch time
        assert abs(r["achieved"] - r["executed_flops_per_launch"] / r["launch_us"] / 1e6) < 0.01 * r["achieved"]
    if "launch_us" in r and r["bound"] == "hbm":
        assert abs(r["achieved"] - r["bytes_per_launch"] / r["launch_us"] / 1e3) < 0.01 * r["achieved"]


def test_committed_bench_line_has_the_contract_fields():
    path, line = _newest_bench_line()
    assert int(re.search(r"bench_r(\d\d)\.json$", path).group(1)) >= 4, path  # (the newest one, not round 1's)
    assert "\n" not in line  # ONE line
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert "configs[1]" in d["config"]["workload"] and d["config"]["global_batch"] == 32 * d["n_gpus"]
    assert abs(d["value"] - d["config"]["global_batch"] * 1e3 / d["ms_per_step"]) < 0.01 * d["value"]  # windows/s = batch / step time
    _check_roofline(d["roofline"], "mfma")
    assert d["roofline"]["traffic"] and d["roofline"]["traffic"] > 0  # the counter passes of THIS build id are committed with the line
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    # sustained leg: the same step over >= 5 s agrees with the K-step figure
    su = d["sustained"]
    assert su["seconds"] >= 4.5 and abs(su["ms_per_step"] - d["ms_per_step"]) < 0.1 * d["ms_per_step"]
    # whole-step fractions: executed <= algorithmic (the re-association removes FLOPs, it adds none)
    ws = d["whole_step"]
    assert 0 < ws["executed_flop_frac"] <= ws["flop_frac"] and 0 < ws["hbm_frac"] < 1
    # sampling leg (configs[3]) with its own roofline object
    sm = d["sample"]
    assert sm["batch_per_gpu"] == 512 and sm["steps"] == 50 and abs(sm["value"] - 512 * d["n_gpus"] / sm["seconds"]) < 0.01 * sm["value"]
    _check_roofline(sm["roofline"], "mfma")
    assert 0 < sm["whole_leg"]["executed_flop_frac"] <= sm["whole_leg"]["flop_frac"]
    # the two HBM-bound kernels
    for leg in ("train", "sample"):
        _check_roofline(d["roofline_hbm"][leg], "hbm")
    # configs[4]
    lw = d["large_window"]
    assert "configs[4]" in lw["workload"] and lw["batch"] == 8
    assert abs(lw["train"]["windows_per_s"] - lw["batch"] * 1e3 / lw["train"]["ms_per_step"]) < 0.01 * lw["train"]["windows_per_s"]
    assert 0 < lw["train"]["executed_flop_frac"] <= lw["train"]["flop_frac"] and lw["sample"]["ms_per_step"] > 0
    # small batches (the reference's own batch_size 1, configs[0]'s 4): a figure per form that is reported
    for b in ("b1", "b4"):
        sb = d["small_batch"][b]
        assert sb["ms_per_step"] > 0 and abs(sb["windows_per_s"] - int(b[1:]) * 1e3 / sb["ms_per_step"]) < 0.01 * sb["windows_per_s"]
        if "graph_ms_per_step" in sb:  # (lines up to round 4 carried the captured-graph replay of the step; round 5 dropped the figure)
            assert abs(sb["graph_speedup"] - sb["ms_per_step"] / sb["graph_ms_per_step"]) < 0.011
    t = d["transformer"]  # the CustomTransformer leg
    assert t["train_b1"]["value"] > 0 and t["train_b32"]["value"] > t["train_b1"]["value"]
    assert t["roofline"]["bound"] == "mfma" and 0 < t["roofline"]["frac"] < 1 and t["cpu_baseline"]["kind"] == "port"
    # provenance: the line names the native build it was measured on
    assert re.fullmatch(r"[0-9a-f]{16}", d["build_id"])


def test_bench_cli_surface():
    src = open(os.path.join(REPO, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert re.search(r'add_argument\("%s", type=int' % flag, src), flag
    assert "init_process_group(\"nccl\"" in src and "127.0.0.1" in src and "ReduceOp.MAX" in src


def test_single_rank_legs_issue_no_collective_alone():
    """every collective between two barriers must be issued by all ranks -- the legs that call _train_one_batch (and with it the
    flat gradient all-reduce) on rank 0 only must be gated on a single-process run."""
    src = open(os.path.join(REPO, "bench.py")).read()
    assert re.search(r"if rank == 0 and world == 1 and not args\.train_only:\s*\n\s*small = \{\}", src)
    assert "cpu_baseline(net) if (rank == 0 and world == 1" in src and "if rank == 0 and world == 1 and not args.no_transformer" in src


def _run_bench_under_torchrun(extra):
    import socket
    import subprocess
    import sys

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(REPO, "bench.py"), "--gpus", "1"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_with_sampling_leg_under_torch_distributed_run_with_one_rank():
    """The same launch mode WITH the sampling leg and the rank-0-only legs that a multi-GPU run keeps (no --train-only): between the train
    leg's barriers and the final gather every rank runs the sampling leg on its shard of the 512 windows and the MAX over ranks of its time is
    taken; the legs that call _train_one_batch on one rank alone (small batch, large window, CPU, transformer) stay gated on world == 1 -- here
    the world IS 1 but the process group is live, so their collectives run too.  Bounded: 3 train steps, a 64-window sampling batch."""
    d = _run_bench_under_torchrun(["--steps", "3", "--warmup", "1", "--no-cpu", "--no-transformer", "--no-large-window", "--sample-batch", "64"])
    assert d["n_gpus"] == 1 and d["config"]["parallelism"] == "dp1" and d["steps"] == 3
    assert d["value"] > 0 and abs(d["last_loss"]) < 1e3
    sm = d["sample"]
    assert sm["batch_per_gpu"] == 64 and sm["steps"] == 50 and sm["value"] > 0 and abs(sm["value"] - 64 / sm["seconds"]) < 0.01 * sm["value"]
    assert d["small_batch"]["b1"]["ms_per_step"] > 0  # (a rank-0-only leg that issues the flat all-reduce inside the live group)


@pytest.mark.gpu
def test_bench_train_leg_under_torch_distributed_run_with_one_rank():
    """The driver's multi-GPU launch mode with the one GPU this box has: `python -m torch.distributed.run --nproc-per-node 1 bench.py
    --train-only` (the launcher is a CHILD process started before it touches the GPU; this test process does not hand its GPU state over).
    Checks that the `dist_on` path of the bench (RCCL process group, replica sync, flat all-reduce inside every step, barriers, MAX over
    ranks) still produces the contract's JSON line."""
    import json as _json
    import socket
    import subprocess
    import sys

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(REPO, "bench.py"), "--gpus", "1", "--train-only", "--steps", "3", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = _json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["parallelism"] == "dp1" and d["steps"] == 3 and d["warmup"] == 1
    assert d["value"] > 0 and d["last_loss"] == d["last_loss"] and abs(d["last_loss"]) < 1e3  # finite
    assert d["small_batch"] is None and d["cpu_baseline"] is None  # --train-only
