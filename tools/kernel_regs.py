#!/usr/bin/env python3
"""Register counts, spills and LDS of the gfx950 kernels inside a built library, from the code objects' metadata notes.
    python tools/kernel_regs.py [library.so] [name-substring ...]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                          "metagenome_vector_sketches_amd", "libmvs_hip.so")
want = sys.argv[2:]
with tempfile.TemporaryDirectory() as tmp:
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib], check=True)
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    for k, s in enumerate(starts):
        piece = os.path.join(tmp, "bundle%d" % k)
        open(piece, "wb").write(blob[s:starts[k + 1] if k + 1 < len(starts) else len(blob)])
        co = os.path.join(tmp, "co%d" % k)
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + piece,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True)
        if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
            continue
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True).stdout.decode()
        for blk in notes.split("- .agpr_count:")[1:]:
            f = dict(re.findall(r"\.(\w+):\s+(\S+)", "agpr_count:" + blk.split("\n  - ")[0]))
            name = subprocess.run(["c++filt", f.get("name", "?")], capture_output=True).stdout.decode().strip()
            name = name.replace("mvs::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            if want and not any(w in name for w in want):
                continue
            print("%-58s vgpr %3s agpr %3s sgpr %3s spill v%s s%s lds %6s scratch %s" %
                  (name[:58], f.get("vgpr_count"), f.get("agpr_count"), f.get("sgpr_count"), f.get("vgpr_spill_count"),
                   f.get("sgpr_spill_count"), f.get("group_segment_fixed_size"), f.get("private_segment_fixed_size")))
