#!/bin/bash
# VGPR / SGPR / spill / LDS figures of the kernels in an object file: tools/kernel_regs.sh movedepth_amd/csrc/costvol.o [name regex]
set -e
obj=$1; pat=${2:-.}
tmp=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat.bin $obj
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$tmp/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/dev.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/dev.co > $tmp/notes.txt
PAT="$pat" python3 - $tmp/notes.txt <<'PY'
import os, re, subprocess, sys
txt = open(sys.argv[1]).read()
pat = os.environ["PAT"]
for blk in txt.split("- .agpr_count")[1:]:
    g = lambda k: (re.search(r"\n    \.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("symbol").replace(".kd", "")
    name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
    if re.search(pat, name):
        print("%-70s vgpr %s agpr %s sgpr %s spill %s lds %s scratch %s" % (name[:70], g("vgpr_count"), blk.split("\n")[0].strip(), g("sgpr_count"), g("vgpr_spill_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
PY
rm -rf $tmp
