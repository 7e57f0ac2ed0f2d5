"""Per-kernel totals of one train step from two rocprofv3 --kernel-trace CSVs: everything on one stream (DQ_NO_SIDE_STREAM=1) against
the default (weight gradients on the side stream).   python tools/trace_compare.py <serial.csv> <side.csv>"""
import collections
import csv
import re
import sys


def step(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "k_adamw_clip" in r["Kernel_Name"]]
    return rows[idx[-2] + 1: idx[-1] + 1]


def short(n):
    return re.sub(r"\(.*", "", n).replace("void dq::", "").replace("dq::", "")


def agg(seg):
    d, c = collections.defaultdict(float), collections.Counter()
    for r in seg:
        k = short(r["Kernel_Name"])
        d[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        c[k] += 1
    return d, c


def main():
    a, b = step(sys.argv[1]), step(sys.argv[2])
    da, ca = agg(a)
    db, _ = agg(b)
    side = lambda k: "wgrad" in k
    red = lambda k: "part_reduce" in k
    for name, d, seg in (("one stream ", da, a), ("side stream", db, b)):
        main_us = sum(v for k, v in d.items() if not side(k) and not red(k))
        print(f"{name}: main-chain kernels {main_us:.0f} us, weight-gradient kernels + their reduces {sum(v for k, v in d.items() if side(k)):.0f} us, "
              f"k_part_reduce {sum(v for k, v in d.items() if red(k)):.0f} us, step span under the profiler "
              f"{(int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e3:.0f} us")
    print(f"{'kernel':44s} {'calls':>5s} {'one stream us':>14s} {'side stream us':>15s}")
    for k in sorted(da, key=lambda k: -da[k]):
        print(f"{k[:44]:44s} {ca[k]:5d} {da[k]:14.1f} {db.get(k, 0.0):15.1f}")


if __name__ == "__main__":
    main()
