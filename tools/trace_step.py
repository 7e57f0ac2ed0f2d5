"""Timeline of ONE train step from a rocprofv3 --kernel-trace CSV: per queue busy time and gaps, the main queue's kernels in order, and
where the side queue (weight gradients) runs relative to it.   python tools/trace_step.py <kernel_trace.csv> [--side]"""
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(.*", "", n)
    return n.replace("void dq::", "").replace("dq::", "")[:44]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "k_adamw_clip" in r["Kernel_Name"]]
    seg = rows[idx[-2] + 1: idx[-1] + 1]
    t0 = int(seg[0]["Start_Timestamp"])
    queues = sorted({r["Queue_Id"] for r in seg}, key=lambda q: -sum(1 for r in seg if r["Queue_Id"] == q))
    for q in queues:
        s = [r for r in seg if r["Queue_Id"] == q]
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in s)
        print(f"queue {q}: {len(s)} kernels, busy {busy / 1e3:.1f} us, first start {(int(s[0]['Start_Timestamp']) - t0) / 1e3:.1f}, "
              f"last end {(int(s[-1]['End_Timestamp']) - t0) / 1e3:.1f}")
    which = queues[1] if "--side" in sys.argv and len(queues) > 1 else queues[0]
    prev = None
    for r in seg:
        if r["Queue_Id"] != which:
            continue
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (st - prev) / 1e3 if prev else 0.0
        print(f"{(st - t0) / 1e3:8.1f} {(en - st) / 1e3:7.1f} gap {gap:6.1f}  {short(r['Kernel_Name'])}  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}")
        prev = en


if __name__ == "__main__":
    main()
