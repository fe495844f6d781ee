#!/usr/bin/env python3
"""VGPRs / scratch / LDS of every kernel in libmgf_hip.so (llvm-readelf notes of the gfx950 code objects).
Usage: python tools/kernel_resources.py [name pattern] [--scratch]   (--scratch: only kernels that use scratch memory)"""
import glob, os, re, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"
pat = next((a for a in sys.argv[1:] if not a.startswith("--")), "")
only_scratch = "--scratch" in sys.argv
tmp = tempfile.mkdtemp()
shutil.copy(os.path.join(ROOT, "mgf_amd", "libmgf_hip.so"), os.path.join(tmp, "l.so"))
subprocess.run([LLVM + "llvm-objdump", "--offloading", "l.so"], cwd=tmp, capture_output=True)
txt = "".join(subprocess.run([LLVM + "llvm-readelf", "--notes", f], capture_output=True, text=True).stdout for f in glob.glob(os.path.join(tmp, "l.so.*gfx950")))
shutil.rmtree(tmp)
rows = set()
for blk in txt.split("  - .agpr_count")[1:]:
    def g(k):
        m = re.search(r"\." + k + r":\s+(\S+)", blk)
        return m.group(1) if m else "?"
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void ", "").replace("mgf::", "")
    rows.add((name, g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"), g("sgpr_spill_count"), g("vgpr_spill_count")))
print(f"{'kernel':80s} vgpr sgpr scratch    lds sgpr-spills vgpr-spills")
for r in sorted(rows):
    if pat in r[0] and (not only_scratch or r[3] not in ("0", "?")):
        print(f"{r[0][:80]:80s} {r[1]:>4s} {r[2]:>4s} {r[3]:>7s} {r[4]:>6s} {r[5]:>11s} {r[6]:>11s}")
