#!/bin/bash
# The measurement batch behind profiles/ (run ON THE GPU BOX, from the repo root):  bash tools/measure_round.sh <tag>   e.g. r02
#   1. counter passes for roofline.traffic           -> gpurun_out/pmc/pmc_linattn.json
#   2. rocprofv3 kernel stats: 13 train steps         -> gpurun_out/<tag>_train_kernel_stats.csv, <tag>_train_under_rocprof.json
#   3. rocprofv3 kernel stats: sampling leg alone     -> gpurun_out/<tag>_sample_kernel_stats.csv
#   4. python bench.py (defaults)                     -> gpurun_out/bench_<tag>.json
# Copy what is to be judged from gpurun_out/ into profiles/ afterwards.
set -e -o pipefail
TAG=${1:-r02}
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/pmc_passes.sh gpurun_out/pmc
cp gpurun_out/pmc/pmc_linattn.json profiles/pmc_linattn.json   # bench.py quotes roofline.traffic from profiles/ when the build id matches
rm -rf gpurun_out/prof_train gpurun_out/prof_sample
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -- python3 bench.py --train-only --steps 10 --warmup 3 \
  > gpurun_out/${TAG}_train_under_rocprof.json 2> gpurun_out/prof_train.err
cp "$(ls gpurun_out/prof_train/*/*_kernel_stats.csv | head -1)" gpurun_out/${TAG}_train_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sample -- python3 tools/prof_sample.py 512 5 \
  > gpurun_out/prof_sample.log 2> gpurun_out/prof_sample.err
cp "$(ls gpurun_out/prof_sample/*/*_kernel_stats.csv | head -1)" gpurun_out/${TAG}_sample_kernel_stats.csv
rm -rf gpurun_out/prof_train gpurun_out/prof_sample gpurun_out/pmc/FETCH_SIZE gpurun_out/pmc/WRITE_SIZE gpurun_out/pmc/SQ gpurun_out/pmc/S_FETCH_SIZE gpurun_out/pmc/S_WRITE_SIZE gpurun_out/pmc/S_SQ   # (raw traces: not merged back)
python3 bench.py > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err
tail -c 1500 gpurun_out/bench_${TAG}.json
