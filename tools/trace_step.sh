#!/bin/bash
# Kernel trace of a few train steps (run ON THE GPU BOX from the repo root): per-kernel table of the last step -> gpurun_out/step_<tag>.txt
#   bash tools/trace_step.sh <tag> [batch]
set -e -o pipefail
TAG=${1:-cur}; B=${2:-32}
export TMPDIR=/tmp
rm -rf gpurun_out/pt_$TAG
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pt_$TAG -- python3 tools/prof_sample.py $B 12 train > gpurun_out/pt_$TAG.log 2>&1
cp "$(ls gpurun_out/pt_$TAG/*/*_kernel_trace.csv | head -1)" gpurun_out/trace_$TAG.csv
rm -rf gpurun_out/pt_$TAG
python3 tools/step_timeline.py gpurun_out/trace_$TAG.csv > gpurun_out/step_$TAG.txt
tail -45 gpurun_out/step_$TAG.txt
