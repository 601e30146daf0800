#!/bin/bash
# A/B builds of the plane-sweep kernels: tools/ab_build.sh NAME -DFLAG=... builds build_ab/libmd_NAME.so with costvol.hip (all three
# element types) recompiled under the extra flags; select it at run time with MOVEDEPTH_HIP_LIB=build_ab/libmd_NAME.so.
set -e
cd "$(dirname "$0")/../movedepth_amd/csrc"
name=$1; shift
out=../../build_ab; mkdir -p $out/$name
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed"
/opt/rocm/bin/hipcc $F "$@" -c costvol.hip -o $out/$name/costvol.o &
/opt/rocm/bin/hipcc $F "$@" -DMD_CV_IO=1 -c costvol.hip -o $out/$name/costvol_bf16.o &
/opt/rocm/bin/hipcc $F "$@" -DMD_CV_IO=2 -c costvol.hip -o $out/$name/costvol_f16.o &
wait
others=$(ls *.o | grep -v '^costvol')
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libmd_$name.so $others $out/$name/*.o
echo built $out/libmd_$name.so
