#!/bin/bash
# gfx950 disassembly of the kernels in an object file: tools/kernel_isa.sh movedepth_amd/csrc/costvol_f16.o > /tmp/isa.s
set -e
obj=$1
tmp=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat.bin $obj
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$tmp/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/dev.co
/opt/rocm/lib/llvm/bin/llvm-objdump -d --mcpu=gfx950 $tmp/dev.co | c++filt
rm -rf $tmp
