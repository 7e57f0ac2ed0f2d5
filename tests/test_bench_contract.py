"""CPU: the committed benchmark line (profiles/bench_r01.json, written by `python bench.py` on an MI355X) carries every field the
benchmark contract names, with consistent values; and bench.py's argument surface is the contract's."""
import json
import os
import re

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
