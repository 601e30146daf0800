#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_fwdk; mkdir -p $O
B="timeout 300 python tools/bench_costvol.py --layout ndhwc --feat nhwc"
run() { local tag=$1; shift; echo "== $tag"; env "$@" $B 2>&1 | grep "kernel only\|Error\|error" | grep fwd | sed 's/(dispatch start.stop events inside the library) //'; }
{
for ov in 1.0 2.5 2.0 3.0; do
echo "#### oversub $ov"
run sane PRIOR=smooth MD_COSTVOL_FWD_OVERSUB=$ov
run "f16 sane" PRIOR=smooth DT=f16 MD_COSTVOL_FWD_OVERSUB=$ov
run moderate PRIOR=smooth POSE_ROT=0.05 POSE_TRANS=0.3 MD_COSTVOL_FWD_OVERSUB=$ov
run kitti PRIOR=kitti POSE_KITTI=1.0 MD_COSTVOL_FWD_OVERSUB=$ov
B0=$B; B="$B --B 6 --h 80 --w 256 --D 128 --dtype bf16"
run "cfg4 sane" PRIOR=smooth MD_COSTVOL_FWD_OVERSUB=$ov
run "cfg4 moderate" PRIOR=smooth POSE_ROT=0.05 POSE_TRANS=0.3 MD_COSTVOL_FWD_OVERSUB=$ov
B="$B0 --B 2 --h 48 --w 160"
run "B=2" PRIOR=smooth MD_COSTVOL_FWD_OVERSUB=$ov
B="$B0 --B 12"
run "B=12" PRIOR=smooth MD_COSTVOL_FWD_OVERSUB=$ov
B=$B0
done
} > $O/cases.txt 2>&1
cat $O/cases.txt
