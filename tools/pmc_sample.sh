#!/bin/bash
set -e -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/pmc_s
rm -rf $OUT; mkdir -p $OUT
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/$ctr -- python3 tools/prof_sample.py 512 2 > $OUT/$ctr.log 2>&1
done
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/SQ -- python3 tools/prof_sample.py 512 2 > $OUT/SQ.log 2>&1 || echo "SQ failed"
ls $OUT/*/*/ | head
