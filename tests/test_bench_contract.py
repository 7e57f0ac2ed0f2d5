"""CPU: the committed benchmark line (profiles/bench_r01.json, written by `python bench.py` on an MI355X) carries every field the
benchmark contract names, with consistent values; and bench.py's argument surface is the contract's."""
import json
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    line = open(os.path.join(REPO, "profiles", "bench_r01.json")).read().strip()
    assert "\n" not in line  # ONE line
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["global_batch"] * 1e3 / d["ms_per_step"]) < 0.01 * d["value"]  # windows/s = batch / step time
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] > 0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    t = d["transformer"]  # the CustomTransformer leg
    assert t["train_b1"]["value"] > 0 and t["train_b32"]["value"] > t["train_b1"]["value"]
    assert t["roofline"]["bound"] == "mfma" and 0 < t["roofline"]["frac"] < 1 and t["cpu_baseline"]["kind"] == "port"


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
