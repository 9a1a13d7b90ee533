#!/usr/bin/env python3
"""VGPR / spill / scratch / SGPR use of the gfx950 kernels inside a hipcc object file:
scripts/obj_resources.py <file.o> [substring of the mangled kernel name]"""
import re, subprocess, sys, tempfile, os
LLVM = "/opt/rocm/lib/llvm/bin/"
obj, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
with tempfile.TemporaryDirectory() as d:
    fb, co = os.path.join(d, "fb.bin"), os.path.join(d, "k.co")
    subprocess.check_call([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fb, obj])
    subprocess.check_call([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fb, "--output=" + co,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"])
    t = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    sz = {m.group(2): int(m.group(1), 16) for m in re.finditer(r"^\s*\d+: \S+\s+(\S+)\s+FUNC\s+\S+\s+\S+\s+\S+\s+(\S+)$",
          subprocess.run([LLVM + "llvm-readelf", "-sW", co], capture_output=True, text=True).stdout, re.M)}
for blk in t.split("- .agpr_count")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, blk) or [None, "?"])[1]
    n = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if pat in n:
        print(n[:100], "vgpr", g("vgpr_count"), "spill", g("vgpr_spill_count"), "scratch", g("private_segment_fixed_size"),
              "sgpr", g("sgpr_count"), "lds", g("group_segment_fixed_size"))
