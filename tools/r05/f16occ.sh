#!/bin/bash
# Round 5: half-precision forward with more resident waves (3 workgroups of 8 waves per CU; 4-wave workgroups)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_f16occ; mkdir -p $O
B="timeout 300 python tools/bench_costvol.py --layout ndhwc --feat nhwc"
{
for v in "" build_ab/libmd_fw6.so build_ab/libmd_fnw4.so; do
for cfg in "" "--dtype f16" "--dtype bf16" "--B 6 --h 80 --w 256 --D 128 --dtype bf16"; do
  echo "== ${v:-in-tree} shape='${cfg:-fp32}'"
  env PRIOR=smooth MOVEDEPTH_HIP_LIB=$v $B $cfg 2>&1 | grep "kernel only.*fwd" | sed 's/(dispatch start.stop events inside the library) //'
done; done
} > $O/f16occ.txt 2>&1
cat $O/f16occ.txt
