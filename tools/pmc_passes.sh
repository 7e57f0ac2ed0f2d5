#!/bin/bash
# Counter passes behind `roofline.traffic` (run ON THE GPU BOX, from the repo root):
#   bash tools/pmc_passes.sh [outdir]
# Two separate rocprofv3 passes (FETCH_SIZE, WRITE_SIZE; counters only together with --kernel-trace, as
# /opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes) + one SQ pass over tools/pmc_linattn.py, then tools/pmc_parse.py
# turns the CSVs into <outdir>/pmc_linattn.json, stamped with the build id of the native sources that ran.
set -e -o pipefail
OUT=${1:-gpurun_out/pmc}
mkdir -p "$OUT"
export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$OUT/$ctr" -- python3 tools/pmc_linattn.py > "$OUT/$ctr.log" 2>&1
done
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv \
  -d "$OUT/SQ" -- python3 tools/pmc_linattn.py > "$OUT/SQ.log" 2>&1 || echo "SQ pass failed (not needed for roofline.traffic)"
# the sampling leg's dominant launch (k_linattn_fwd<4, 64> over 512 x 400 rows): the same three passes into their own directories
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$OUT/S_$ctr" -- python3 tools/pmc_linattn.py sample > "$OUT/S_$ctr.log" 2>&1
done
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv \
  -d "$OUT/S_SQ" -- python3 tools/pmc_linattn.py sample > "$OUT/S_SQ.log" 2>&1 || echo "sampling SQ pass failed"
python3 tools/pmc_parse.py "$OUT"
