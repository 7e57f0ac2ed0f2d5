#!/bin/bash
# rocprofv3 kernel stats of the sampling leg alone (run ON THE GPU BOX):  bash tools/prof_sample_stats.sh <tag>  -> gpurun_out/<tag>_sample_kernel_stats.csv
set -e -o pipefail
TAG=${1:-cur}
export TMPDIR=/tmp
rm -rf gpurun_out/prof_sample
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sample -- python3 tools/prof_sample.py 512 5 > gpurun_out/prof_sample.log 2> gpurun_out/prof_sample.err
cp "$(ls gpurun_out/prof_sample/*/*_kernel_stats.csv | head -1)" gpurun_out/${TAG}_sample_kernel_stats.csv
rm -rf gpurun_out/prof_sample
python3 tools/kstats.py gpurun_out/${TAG}_sample_kernel_stats.csv --steps 7 --top 60
