#!/bin/bash
# A-B of one environment switch on the GPU box:  bash tools/ab_env.sh DQ_NO_TINY  -> gpurun_out/ab_<VAR>_{off,on}.json (bench lines, no CPU / transformer / large-window legs)
set -e -o pipefail
VAR=${1:?variable name}
mkdir -p gpurun_out
# the switches exist in the development build only (csrc/dq_dev.h; make -C diffusion-deconvolution-dia-msms-data_amd dev)
export DQ_HIP_LIB=${DQ_HIP_LIB:-$PWD/diffusion-deconvolution-dia-msms-data_amd/build/dev/libdq_hip_dev.so}
env $VAR=1 python3 bench.py --no-cpu --no-transformer --no-large-window --steps 100 > gpurun_out/ab_${VAR}_on.json 2> gpurun_out/ab_${VAR}_on.err
python3 bench.py --no-cpu --no-transformer --no-large-window --steps 100 > gpurun_out/ab_${VAR}_off.json 2> gpurun_out/ab_${VAR}_off.err
python3 - <<PY
import json
for tag in ("on", "off"):
    d = json.loads(open("gpurun_out/ab_${VAR}_%s.json" % tag).read().strip().splitlines()[-1])
    sb = d.get("small_batch") or {}
    print("${VAR}=%s" % ("1" if tag == "on" else "unset"), "train ms", d["ms_per_step"], "windows/s", d["value"], "sample", d["sample"]["value"],
          "b1", (sb.get("b1") or {}).get("ms_per_step"), "b4", (sb.get("b4") or {}).get("ms_per_step"), "loss", d["last_loss"])
PY
