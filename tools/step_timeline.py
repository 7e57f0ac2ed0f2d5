#!/usr/bin/env python3
"""Timeline of the LAST train step in a rocprofv3 kernel trace (csv): every launch with its queue, start, duration and grid, then totals
per kernel family on the main queue.   usage: tools/step_timeline.py trace.csv"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["n"] = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "").replace("dq::", "")
rows.sort(key=lambda r: r["s"])
ad = [i for i, r in enumerate(rows) if "adamw" in r["n"]]
step = rows[ad[-2] + 1:ad[-1] + 1]
t0 = step[0]["s"]
qs = {}
for r in step:
    qs.setdefault(r["Queue_Id"], []).append(r)
main_q = max(qs, key=lambda q: len(qs[q]))
for r in step:
    mark = "" if r["Queue_Id"] == main_q else "      >>"
    print(f"{(r['s'] - t0) / 1e3:8.1f} {(r['e'] - r['s']) / 1e3:7.1f} {mark} {r['n'][:56]} g={int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])}x{r['Grid_Size_Y']}")
print(f"--- step wall {(step[-1]['e'] - t0) / 1e3:.1f} us, {len(step)} launches")
for q, l in qs.items():
    print(f"queue {q}: {len(l)} launches, busy {sum(r['e'] - r['s'] for r in l) / 1e3:.1f} us")
# idle stretches of the main queue (nothing of it running: it waits for the other queue, or for the host)
prev_e, idle = None, 0.0
for r in qs[main_q]:
    if prev_e is not None and r["s"] - prev_e > 3000:
        print(f"  main queue idle {(r['s'] - prev_e) / 1e3:6.1f} us before {r['n'][:50]} at {(r['s'] - t0) / 1e3:.1f}")
    if prev_e is not None and r["s"] > prev_e:
        idle += (r["s"] - prev_e) / 1e3
    prev_e = max(prev_e or 0, r["e"])
print(f"main queue idle in total: {idle:.1f} us")
grp = {}
for r in qs[main_q]:
    k = re.sub(r"<.*", "", r["n"])
    g = grp.setdefault(k, [0.0, 0])
    g[0] += (r["e"] - r["s"]) / 1e3
    g[1] += 1
for k, v in sorted(grp.items(), key=lambda kv: -kv[1][0])[:30]:
    print(f"  {k:44s} {v[0]:8.1f} us {v[1]:3d}")
