#!/usr/bin/env python3
"""Print a rocprofv3 *_kernel_stats.csv as a table (per-step figures when --steps is given) and group totals.
usage: tools/kstats.py file.csv [--steps N] [--top K]"""
import csv, sys, re, argparse
ap = argparse.ArgumentParser(); ap.add_argument("csv"); ap.add_argument("--steps", type=int, default=1); ap.add_argument("--top", type=int, default=40)
a = ap.parse_args()
rows = list(csv.DictReader(open(a.csv)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
grp = {}
def group(n):
    for k in ("linattn_bwd", "linattn_fwd", "linattn_dw", "linattn_prepare", "res_bwd", "res_fwd", "res_wg_reduce", "conv_wgrad", "wgrad_reduce", "part_reduce",
              "conv_bwd_data", "conv_fwd", "attn_", "gemm", "block_bwd", "time_", "adamw", "sumsq", "copyBuffer", "level_"):
        if k in n: return k
    return "other"
calls = 0
for r in rows:
    n = re.sub(r"\(.*", "", r["Name"].replace("(anonymous namespace)::", "")).replace("void ", "").replace("dq::", "")
    r["n"] = n
    g = group(n); t = float(r["TotalDurationNs"]); grp[g] = grp.get(g, [0, 0]); grp[g][0] += t; grp[g][1] += int(r["Calls"]); calls += int(r["Calls"])
for r in rows[:a.top]:
    print(f"{r['n'][:60]:60s} {int(r['Calls'])/a.steps:7.1f}/step {float(r['TotalDurationNs'])/1e3/a.steps:9.1f}us/step {float(r['AverageNs'])/1e3:8.1f}us avg {float(r['Percentage']):5.1f}%")
print(f"--- total {tot/1e6/a.steps:.3f} ms/step, {calls/a.steps:.0f} launches/step")
for g, (t, c) in sorted(grp.items(), key=lambda kv: -kv[1][0]):
    print(f"  {g:18s} {t/1e3/a.steps:9.1f} us/step  {c/a.steps:6.1f} launches/step")
