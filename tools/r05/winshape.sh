#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_winshape; mkdir -p $O
B="timeout 300 python tools/bench_costvol.py --layout ndhwc --feat nhwc"
run() { local tag=$1; shift; echo "== $tag"; env "$@" $B 2>&1 | grep "kernel only\|Error\|error" | grep bwd | sed 's/(dispatch start.stop events inside the library) //'; }
{
for v in "" win23 win33 win101 win04; do
echo "#### ${v:-shipped (22 x 6)}"
L=""; [ -n "$v" ] && L="MOVEDEPTH_HIP_LIB=build_ab/libmd_$v.so"
run sane PRIOR=smooth A=1 $L
run white PRIOR=white $L
run moderate PRIOR=smooth POSE_ROT=0.05 POSE_TRANS=0.3 $L
run wild PRIOR=smooth POSE_ROT=0.3 POSE_TRANS=2.0 $L
run kitti PRIOR=kitti POSE_KITTI=1.0 $L
run kitti2 PRIOR=kitti POSE_KITTI=2.0 $L
done
} > $O/cases.txt 2>&1
cat $O/cases.txt
