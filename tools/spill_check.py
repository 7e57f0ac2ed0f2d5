#!/usr/bin/env python3
"""Cross-check the kernels a set of rocprofv3 kernel-stats CSVs launched against the per-kernel resource table of the shipped build
(tools/kernel_resources.py --csv): which launched kernels spill registers or use scratch.
usage: tools/spill_check.py profiles/r04_kernel_resources.csv profiles/r04_train_kernel_stats.csv [more stats.csv ...]"""
import csv
import re
import sys


def norm(n):
    n = re.sub(r"\s+", " ", n.strip().strip('"'))
    n = n.replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", n).replace("void ", "").strip()


res = {}
for r in csv.DictReader(open(sys.argv[1])):
    res[norm(r["kernel"])] = r
launched, missing, spilling, sspilling = set(), [], [], []
for path in sys.argv[2:]:
    if path.endswith(".txt"):  # tools/kernel_census.sh output: "<calls>\t<name>"
        names = [ln.rstrip("\n").split("\t", 1)[1] for ln in open(path) if "\t" in ln]
    else:
        names = [r["Name"] for r in csv.DictReader(open(path))]
    for nm in names:
        n = norm(nm)
        if not (n.startswith("dq::") or n.startswith("k_")):
            continue  # torch / runtime kernels
        launched.add(n)
for n in sorted(launched):
    r = res.get(n)
    if r is None:
        missing.append(n)
        continue
    if int(r["vgpr_spill"]) or int(r["scratch"]):
        spilling.append((n, r["vgpr"], r["scratch"], r["vgpr_spill"]))
    if int(r["sgpr_spill"]):
        sspilling.append((n, r["sgpr"], r["sgpr_spill"]))
print(f"{len(launched)} distinct library kernels launched; {len(missing)} not matched by name in the table ({missing})")
print(f"launched kernels with .vgpr_spill_count > 0 or scratch > 0: {len(spilling)}")
for s in spilling:
    print("  ", s)
print(f"launched kernels with .sgpr_spill_count > 0 (scalars parked in VGPR lanes: v_writelane / v_readlane, no memory traffic): {len(sspilling)}")
for s in sorted(sspilling, key=lambda t: -int(t[2])):
    print("  ", s)
