#!/bin/bash
# work stealing in the plane-sweep backward: parity tests, then the parallax cases with and without it
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_steal1; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "costvol" > $O/pytest_costvol.log 2>&1; echo "pytest rc $?" >> $O/pytest_costvol.log
tail -5 $O/pytest_costvol.log
B="timeout 300 python tools/bench_costvol.py --layout ndhwc --feat nhwc"
run() { local tag=$1; shift; echo "== $tag"; env "$@" $B 2>&1 | grep "kernel only\|stats\|lifetimes\|timeline\|per-workgroup\|mean \|least squares\|Error\|error" | sed 's/(dispatch start.stop events inside the library) //'; }
{
for st in 1 0; do
  echo "#### MD_COSTVOL_STEAL=$st"
  run sane PRIOR=smooth MD_COSTVOL_STEAL=$st
  run white PRIOR=white MD_COSTVOL_STEAL=$st
  run moderate PRIOR=smooth POSE_ROT=0.05 POSE_TRANS=0.3 MD_COSTVOL_STEAL=$st
  run wild PRIOR=smooth POSE_ROT=0.3 POSE_TRANS=2.0 MD_COSTVOL_STEAL=$st
  run kitti PRIOR=kitti POSE_KITTI=1.0 MD_COSTVOL_STEAL=$st
  B0=$B; B="$B --B 6 --h 80 --w 256 --D 128 --dtype bf16"
  run "cfg4 sane" PRIOR=smooth MD_COSTVOL_STEAL=$st
  run "cfg4 moderate" PRIOR=smooth POSE_ROT=0.05 POSE_TRANS=0.3 MD_COSTVOL_STEAL=$st
  run "cfg4 wild" PRIOR=smooth POSE_ROT=0.3 POSE_TRANS=2.0 MD_COSTVOL_STEAL=$st
  B=$B0
done
echo "#### per-workgroup records with stealing"
W="MD_CV_STATS=1 MD_CV_WGSTATS=1 MOVEDEPTH_HIP_LIB=build_ab/libmd_wgstats.so"
run moderate PRIOR=smooth POSE_ROT=0.05 POSE_TRANS=0.3 $W
run kitti PRIOR=kitti POSE_KITTI=1.0 $W
run wild PRIOR=smooth POSE_ROT=0.3 POSE_TRANS=2.0 $W
} > $O/cases.txt 2>&1
cat $O/cases.txt
