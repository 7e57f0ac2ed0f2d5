#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch figures from the gfx950 code objects' metadata notes.
usage: tools/kernel_stats.py [build-dir | file.o] [name-regex]     (default: the package's build/ directory)"""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
here = os.path.dirname(os.path.abspath(__file__))
f = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "diffusion-deconvolution-dia-msms-data_amd", "build")
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
objs = sorted(glob.glob(os.path.join(f, "*.o"))) if os.path.isdir(f) else [f]
rows = []
with tempfile.TemporaryDirectory() as t:
    for o in objs:
        co, fat = os.path.join(t, "dev.co"), os.path.join(t, "fat.bin")
        if subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", o], capture_output=True).returncode:
            continue  # (no device code in this object)
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            f"--input={fat}", f"--output={co}"], capture_output=True)
        if r.returncode or not os.path.exists(co) or os.path.getsize(co) == 0:
            continue
        txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        os.remove(co)
        for blk in txt.split("- .agpr_count:")[1:]:
            def g(k):
                m = re.search(r"\." + k + r":\s*(\S+)", blk)
                return m.group(1) if m else "?"
            rows.append((g("name"), blk.split()[0], g("vgpr_count"), g("sgpr_count"), g("group_segment_fixed_size"),
                         g("private_segment_fixed_size"), g("vgpr_spill_count")))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for r, dn in zip(rows, names):
    dn = re.sub(r"\(.*", "", dn).replace("void ", "").replace("dq::", "").replace("(anonymous namespace)::", "")
    if pat.search(dn):
        print(f"{dn[:56]:56s} vgpr {r[2]:>4s} agpr {r[1]:>4s} sgpr {r[3]:>4s} lds {r[4]:>6s} scratch {r[5]:>5s} spill_v {r[6]:>4s}")
