#!/bin/bash
# A/B builds of the library: tools/ab_build.sh NAME -DFLAG=... builds build_ab/libmd_NAME.so with the sources named in FILES (default:
# costvol, compiled for all three element types) recompiled under the extra flags and every other object taken from the in-tree build
# (run `make -C movedepth_amd/csrc` first); select it at run time with MOVEDEPTH_HIP_LIB=build_ab/libmd_NAME.so.
#   FILES="conv3d_c1 bnrelu3d" tools/ab_build.sh nt -DMD_C1_NT=1 -DMD_BN_NT=1
set -e
cd "$(dirname "$0")/../movedepth_amd/csrc"
name=$1; shift
FILES=${FILES:-costvol}
out=../../build_ab; mkdir -p $out/$name
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed"
skip=""
for f in $FILES; do
    /opt/rocm/bin/hipcc $F "$@" -c $f.hip -o $out/$name/$f.o &
    skip="$skip -e ^$f\\.o\$"
    if [ $f = costvol ]; then
        /opt/rocm/bin/hipcc $F "$@" -DMD_CV_IO=1 -c costvol.hip -o $out/$name/costvol_bf16.o &
        /opt/rocm/bin/hipcc $F "$@" -DMD_CV_IO=2 -c costvol.hip -o $out/$name/costvol_f16.o &
        skip="$skip -e ^costvol_bf16\\.o\$ -e ^costvol_f16\\.o\$"
    fi
done
wait
others=$(ls *.o | grep -v $skip)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libmd_$name.so $others $out/$name/*.o
echo built $out/libmd_$name.so
