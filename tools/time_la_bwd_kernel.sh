#!/bin/bash
# Kernel-only times of the LinearAttention backward (rocprofv3 kernel trace around tools/time_la_bwd.py):  bash tools/time_la_bwd_kernel.sh [rows]
export TMPDIR=/tmp
D=$(mktemp -d /tmp/lrbprof.XXXX)
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/tools/time_la_bwd.py ${1:-12800} > /dev/null 2>&1)
python3 - $D <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "linattn_bwd<" in r["Name"] or "la_rows_bwd" in r["Name"] or "prepare" in r["Name"] or "reduce" in r["Name"]:
        print("%8.1f us  x%-4s %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], r["Name"][:90]))
PY
rm -rf $D
