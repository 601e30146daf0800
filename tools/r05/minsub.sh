#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_minsub; mkdir -p $O
B="timeout 300 python tools/bench_costvol.py --layout ndhwc --feat nhwc"
run() { local tag=$1; shift; echo "== $tag"; env "$@" $B 2>&1 | grep "kernel only\|Error\|error" | sed 's/(dispatch start.stop events inside the library) //'; }
{
for v in 8 16 24 32 48; do
echo "#### min sub-slice $v (forward and backward)"
L="MD_COSTVOL_MIN_SUB=$v MD_COSTVOL_MIN_SUB_BWD=$v"
run white PRIOR=white $L
run moderate PRIOR=smooth POSE_ROT=0.05 POSE_TRANS=0.3 $L
run wild PRIOR=smooth POSE_ROT=0.3 POSE_TRANS=2.0 $L
run kitti PRIOR=kitti POSE_KITTI=1.0 $L
run kitti2 PRIOR=kitti POSE_KITTI=2.0 $L
B0=$B; B="$B --B 6 --h 80 --w 256 --D 128 --dtype bf16"
run "cfg4 moderate" PRIOR=smooth POSE_ROT=0.05 POSE_TRANS=0.3 $L
B=$B0
done
} > $O/cases.txt 2>&1
cat $O/cases.txt
