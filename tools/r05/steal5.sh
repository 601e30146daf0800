#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_steal5; mkdir -p $O
B="timeout 300 python tools/bench_costvol.py --layout ndhwc --feat nhwc"
run() { local tag=$1; shift; echo "== $tag"; env "$@" $B 2>&1 | grep "kernel only\|Error\|error\|per-workgroup\|mean \|least" | grep -v fwd | sed 's/(dispatch start.stop events inside the library) //'; }
suite() {
  run sane PRIOR=smooth "$@"
  run moderate PRIOR=smooth POSE_ROT=0.05 POSE_TRANS=0.3 "$@"
  run wild PRIOR=smooth POSE_ROT=0.3 POSE_TRANS=2.0 "$@"
  run kitti PRIOR=kitti POSE_KITTI=1.0 "$@"
}
{
echo "#### no gather-mode atomics (results wrong), donation off"; suite MD_COSTVOL_STEAL=0 MOVEDEPTH_HIP_LIB=build_ab/libmd_noatomic.so
echo "#### no gather-mode atomics (results wrong), donation on"; suite MOVEDEPTH_HIP_LIB=build_ab/libmd_noatomic.so
echo "#### no gather-mode atomics (results wrong), donation on, max 64 gchunk 32"; suite MOVEDEPTH_HIP_LIB=build_ab/libmd_noatomic.so MD_COSTVOL_STEAL_MAX=64 MD_COSTVOL_STEAL_GCHUNK=32
} > $O/cases.txt 2>&1
cat $O/cases.txt
