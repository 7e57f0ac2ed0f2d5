"""Turns the counter CSVs of tools/pmc_passes.sh into pmc_linattn.json: per kernel the mean FETCH_SIZE / WRITE_SIZE (KB per dispatch)
and the corrected HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE tallies 128-B requests at 64 B, see
/opt/skills/guides/MI355X_MICROARCH.md "HBM"; the factor is re-checked here on k_q_sample, which reads exactly two tensors and
writes one).  No GPU needed."""
import csv
import glob
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffusion-deconvolution-dia-msms-data_amd"))


def counter_means(outdir, ctr, prefix=""):
    """kernel name -> mean counter value per dispatch (prefix "S_": the sampling-size passes)"""
    acc = {}
    for path in glob.glob(os.path.join(outdir, prefix + (ctr if ctr in ("FETCH_SIZE", "WRITE_SIZE") else "SQ"), "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != ctr:
                    continue
                name = row["Kernel_Name"]
                acc.setdefault(name, []).append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    outdir = sys.argv[1]
    from dquartic import _native as N

    fetch, nf = counter_means(outdir, "FETCH_SIZE")
    write, _ = counter_means(outdir, "WRITE_SIZE")
    tensor_bytes = 12800 * 4 * 64 * 4  # tools/pmc_linattn.py: rows x C x n fp32
    res = {"build_id": N.build_id(), "shape": {"C": 4, "n": 64, "rows": 12800}, "tensor_bytes": tensor_bytes, "kernels": {}}
    for name in sorted(set(fetch) | set(write)):
        short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("dq::", "")
        if not any(k in short for k in ("k_linattn", "k_q_sample", "k_la_", "k_level_fwd", "k_res_bwd_wg", "k_res_wg_reduce", "k_rmsnorm")):
            continue
        f_kb, w_kb = fetch.get(name, 0.0), write.get(name, 0.0)
        res["kernels"][short] = {"fetch_kb": round(f_kb, 1), "write_kb": round(w_kb, 1), "dispatches": nf.get(name, 0),
                                 "hbm_bytes": round((2 * f_kb + w_kb) * 1024)}
    cal = next((v for k, v in res["kernels"].items() if "k_q_sample" in k), None)
    if cal:  # reads 2 tensors, writes 1: FETCH_SIZE * factor = 2 tensors
        res["fetch_factor_measured"] = round(2 * tensor_bytes / (cal["fetch_kb"] * 1024), 3)
    # kernels that read 4 bytes per lane (k_level_fwd, k_res_bwd_wg): the guide calls FETCH_SIZE uncalibrated there, so the factor is
    # measured on k_rmsnorm_fwd (same access width; reads exactly one tensor) and applied to them instead of the 16-byte factor 2
    cal4 = next((v for k, v in res["kernels"].items() if "k_rmsnorm" in k), None)
    if cal4 and cal4["fetch_kb"] > 0:
        f4 = tensor_bytes / (cal4["fetch_kb"] * 1024)
        res["fetch_factor_measured_4B"] = round(f4, 3)
        for k, v in res["kernels"].items():
            if any(t in k for t in ("k_level_fwd", "k_res_bwd_wg", "k_rmsnorm")):
                v["hbm_bytes"] = round((f4 * v["fetch_kb"] + v["write_kb"]) * 1024)
                v["fetch_factor"] = round(f4, 3)
    sq = {}
    for ctr in ("SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES"):
        m, _ = counter_means(outdir, ctr)
        for name, v in m.items():
            short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("dq::", "")
            if "k_linattn" in short or "k_la_" in short:
                sq.setdefault(short, {})[ctr] = v
    res["sq"] = sq
    # the sampling-size launch (tools/pmc_linattn.py sample): k_linattn_fwd over 512 x 400 rows
    sf, snf = counter_means(outdir, "FETCH_SIZE", "S_")
    sw, _ = counter_means(outdir, "WRITE_SIZE", "S_")
    res["sample_shape"] = {"C": 4, "n": 64, "rows": 512 * 400}
    res["sample_kernels"], res["sample_sq"] = {}, {}
    for name in sorted(set(sf) | set(sw)):
        short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("dq::", "")
        if "k_linattn_fwd" in short:
            f_kb, w_kb = sf.get(name, 0.0), sw.get(name, 0.0)
            res["sample_kernels"][short] = {"fetch_kb": round(f_kb, 1), "write_kb": round(w_kb, 1), "dispatches": snf.get(name, 0),
                                            "hbm_bytes": round((2 * f_kb + w_kb) * 1024)}
    for ctr in ("SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES"):
        m, _ = counter_means(outdir, ctr, "S_")
        for name, v in m.items():
            short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("dq::", "")
            if "k_linattn_fwd" in short:
                res["sample_sq"].setdefault(short, {})[ctr] = v
    with open(os.path.join(outdir, "pmc_linattn.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
