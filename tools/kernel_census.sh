#!/bin/bash
# Which kernels of libdq_hip.so does anything launch?  Runs ON THE GPU BOX: the GPU test suite (minus the tests that spawn child processes:
# a profiled parent hands the profiler to them) and the whole bench under rocprofv3 --kernel-trace --stats; writes the union of the
# launched kernel names to gpurun_out/<tag>_census_launched.txt.  tools/kernel_census.py compares it with the compiled set.
#   bash tools/kernel_census.sh r05
set -o pipefail
TAG=${1:-r05}
export TMPDIR=/tmp
R=$PWD
D=/tmp/census
rm -rf $D; mkdir -p $D gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $D/t -- python3 -m pytest $R/tests -m gpu -q -p no:cacheprovider \
  -k "not tiny_levels and not test_dp_gloo and not torch_distributed_run and not rccl_process_group" > $R/gpurun_out/${TAG}_census_pytest.log 2>&1
echo "pytest rc=$?" >> $R/gpurun_out/${TAG}_census_pytest.log
rocprofv3 --kernel-trace --stats --output-format csv -d $D/b -- python3 $R/bench.py --steps 5 --warmup 2 > $R/gpurun_out/${TAG}_census_bench.json 2> $R/gpurun_out/${TAG}_census_bench.err
echo "bench rc=$?" >> $R/gpurun_out/${TAG}_census_pytest.log
python3 - $D $R/gpurun_out/${TAG}_census_launched.txt <<'PY'
import csv, glob, sys
names = {}
for f in glob.glob(sys.argv[1] + "/*/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        names[r["Name"]] = names.get(r["Name"], 0) + int(r["Calls"])
with open(sys.argv[2], "w") as o:
    for n in sorted(names):
        o.write("%d\t%s\n" % (names[n], n))
print(len(names), "distinct kernels launched")
PY
tail -3 $R/gpurun_out/${TAG}_census_pytest.log
rm -rf $D
