#!/usr/bin/env python3
"""Per-kernel register / LDS / spill table of a gfx950 code object (the shipped libdq_hip.so or a single build/*.o).

    python tools/kernel_resources.py [path ...] [--spills] [--csv out.csv]

Unbundles the offload bundle (clang-offload-bundler), reads the AMDGPU metadata note (llvm-readelf --notes) and prints, per kernel:
VGPRs (.vgpr_count), AGPRs, SGPRs, LDS bytes (.group_segment_fixed_size), scratch bytes (.private_segment_fixed_size) and
.vgpr_spill_count / .sgpr_spill_count.  --spills lists only kernels with a spill or scratch."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(path):
    out = []
    tmp = tempfile.mkdtemp(prefix="dq_co_")
    lst = subprocess.run([f"{LLVM}/clang-offload-bundler", "--list", "--type=o", f"--input={path}"], capture_output=True, text=True)
    targets = [t for t in lst.stdout.split() if "gfx950" in t]
    if not targets:  # not a bundle: maybe already a device ELF
        return [path]
    for i, t in enumerate(targets):
        o = os.path.join(tmp, f"co{i}.elf")
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={path}", f"--targets={t}", f"--output={o}"], check=True)
        out.append(o)
    return out


def so_code_objects(path):
    """A linked .so keeps the bundle in its .hip_fatbin section."""
    tmp = tempfile.mkdtemp(prefix="dq_fb_")
    fb = os.path.join(tmp, "fatbin")
    r = subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fb], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(fb) or os.path.getsize(fb) == 0:
        return []
    data = open(fb, "rb").read()
    # concatenated bundles (one per translation unit), each starting with the magic string
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), data)]
    outs = []
    for k, st in enumerate(starts):
        en = starts[k + 1] if k + 1 < len(starts) else len(data)
        piece = os.path.join(tmp, f"b{k}.bundle")
        open(piece, "wb").write(data[st:en])
        outs += code_objects(piece)
    return outs


def kernels_of(elf):
    txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", elf], capture_output=True, text=True).stdout
    rows = []
    for blk in txt.split("  - .agpr_count:")[1:]:
        blk = ".agpr_count:" + blk
        g = lambda key: (re.search(rf"\.{key}:\s*(\S+)", blk) or [None, "0"])[1]
        name = g("name")
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        rows.append({"kernel": dem, "vgpr": int(g("vgpr_count")), "agpr": int(g("agpr_count")), "sgpr": int(g("sgpr_count")),
                     "lds": int(g("group_segment_fixed_size")), "scratch": int(g("private_segment_fixed_size")),
                     "vgpr_spill": int(g("vgpr_spill_count")), "sgpr_spill": int(g("sgpr_spill_count"))})
    return rows


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    only_spills = "--spills" in sys.argv
    csv = sys.argv[sys.argv.index("--csv") + 1] if "--csv" in sys.argv else None
    if csv in args:
        args.remove(csv)
    if not args:
        args = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "diffusion-deconvolution-dia-msms-data_amd", "libdq_hip.so")]
    rows = []
    for p in args:
        elfs = so_code_objects(p) or code_objects(p)  # (.so and .o both keep their bundles in a .hip_fatbin section)
        for e in elfs:
            rows += kernels_of(e)
    rows.sort(key=lambda r: r["kernel"])
    if only_spills:
        rows = [r for r in rows if r["vgpr_spill"] or r["scratch"]]
    hdr = ["kernel", "vgpr", "agpr", "sgpr", "lds", "scratch", "vgpr_spill", "sgpr_spill"]
    lines = [",".join(hdr)] + [",".join('"' + str(r[h]) + '"' if h == "kernel" else str(r[h]) for h in hdr) for r in rows]
    if csv:
        open(csv, "w").write("\n".join(lines) + "\n")
    for r in rows:
        print(f"{r['vgpr']:4d} v {r['agpr']:4d} a {r['sgpr']:4d} s {r['lds']:7d} lds {r['scratch']:6d} scr {r['vgpr_spill']:4d} vspill {r['sgpr_spill']:4d} sspill  {r['kernel'][:150]}")
    print(f"{len(rows)} kernels" + (" with spills / scratch" if only_spills else ""))


if __name__ == "__main__":
    main()
