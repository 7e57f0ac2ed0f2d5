#!/bin/bash
# A-B of one development switch with a VALUE on the GPU box (dev build):  bash tools/ab_dev.sh DQ_SIDE_CUMASK 5 [1 ...]
set -e -o pipefail
VAR=${1:?variable}; shift
export DQ_HIP_LIB=$PWD/diffusion-deconvolution-dia-msms-data_amd/build/dev/libdq_hip_dev.so
mkdir -p gpurun_out
for V in "" "$@"; do
  if [ -n "$V" ]; then export $VAR=$V; else unset $VAR; fi
  python3 bench.py --no-cpu --no-transformer --no-large-window --no-sample --steps 200 2> /dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
sb = d.get('small_batch') or {}
print('$VAR=${V:-unset}', 'train ms', d['ms_per_step'], 'sustained', d['sustained']['ms_per_step'], 'b1', (sb.get('b1') or {}).get('ms_per_step'), 'b4', (sb.get('b4') or {}).get('ms_per_step'))"
done
