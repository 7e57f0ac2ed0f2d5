#!/usr/bin/env python3
"""Compiled kernels of the shipped libdq_hip.so (tools/kernel_resources.py --csv) against the kernels anything launched
(gpurun_out/<tag>_census_launched.txt from tools/kernel_census.sh: the GPU test suite + the whole bench under rocprofv3).
usage: tools/kernel_census.py gpurun_out/r05_census_launched.txt [--list]"""
import re
import subprocess
import sys


def norm(n):
    n = re.sub(r"\s+", " ", n.strip().strip('"'))
    n = n.replace("(anonymous namespace)::", "").replace("dq::", "")
    n = re.sub(r"\(.*$", "", n).replace("void ", "").strip()
    return n


launched = {}
for ln in open(sys.argv[1]):
    c, n = ln.rstrip("\n").split("\t", 1)
    n = norm(n)
    if n.startswith(("k_", "dq_")) or "k_" in n.split("<")[0]:
        launched[n] = launched.get(n, 0) + int(c)
out = subprocess.run([sys.executable, "tools/kernel_resources.py"], capture_output=True, text=True).stdout
compiled = set()
for ln in out.splitlines():
    m = re.search(r"sspill\s+(.*)$", ln)
    if m:
        compiled.add(norm(m.group(1)))
never = sorted(compiled - set(launched))
unknown = sorted(set(launched) - compiled)
print(f"{len(compiled)} kernels compiled, {len(set(launched) & compiled)} of them launched by the suite / bench; ratio {len(compiled) / max(1, len(set(launched) & compiled)):.2f}")
print(f"{len(never)} compiled and never launched; {len(unknown)} launched names not matched (first: {[u[:60] for u in unknown[:3]]})")
if "--list" in sys.argv:
    for n in never:
        print("  never:", n)
