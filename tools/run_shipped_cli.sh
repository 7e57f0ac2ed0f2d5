#!/bin/bash
# The reference's shipped configuration end to end through the CLI, ON THE GPU BOX from the repo root:
#   bash tools/run_shipped_cli.sh [outdir]
# `dquartic generate-config` writes the reference's defaults (UNet1d, dim_mults [1,2,2,3,3,4,4], downsample_dim 40000, batch 1);
# the only edits are the ones a user without instrument files has to make: synthetic (34, 40000) windows instead of a parquet
# directory, 2 epochs instead of 10000, wandb off.  The epoch loop writes the "latest" and "best" checkpoints (model + AdamW
# state: 14 GB each at 1.2e9 parameters) into a scratch directory that is deleted afterwards.
set -e -o pipefail
OUT=${1:-gpurun_out/shipped_cli}
mkdir -p "$OUT"
SCR=$(mktemp -d /tmp/dq_shipped.XXXX)
export PYTHONPATH=diffusion-deconvolution-dia-msms-data_amd
python3 -m dquartic.cli generate-config "$SCR/c.json" > "$OUT/run.log" 2>&1
python3 - "$SCR" <<'PY'
import json, sys
p = sys.argv[1] + "/c.json"
c = json.load(open(p))
assert c["model"]["UNet1d"]["downsample_dim"] == 40000 and c["model"]["batch_size"] == 1
c["data"]["synthetic"] = {"n_windows": 6, "RT": 34, "MZ": 40000}
c["model"].update(num_epochs=2, warmup_epochs=1, checkpoint_path=sys.argv[1] + "/best_model.ckpt")
c["wandb"]["use_wandb"] = False
c["threads"] = 0
json.dump(c, open(p, "w"), indent=1)
PY
{ time python3 -m dquartic.cli train "$SCR/c.json" ; } >> "$OUT/run.log" 2>&1
ls -la "$SCR" >> "$OUT/run.log"
rm -rf "$SCR"
tail -20 "$OUT/run.log"
