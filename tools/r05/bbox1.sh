#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_bbox1; mkdir -p $O
if [ "$1" = test ]; then
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "costvol" > $O/pytest_costvol.log 2>&1; echo "pytest rc $?" >> $O/pytest_costvol.log
tail -3 $O/pytest_costvol.log
fi
B="timeout 300 python tools/bench_costvol.py --layout ndhwc --feat nhwc"
run() { local tag=$1; shift; echo "== $tag"; env "$@" $B 2>&1 | grep "kernel only\|Error\|error\|per-workgroup\|mean \|least\|stats" | sed 's/(dispatch start.stop events inside the library) //'; }
suite() {
  run sane PRIOR=smooth "$@"
  run white PRIOR=white "$@"
  run moderate PRIOR=smooth POSE_ROT=0.05 POSE_TRANS=0.3 "$@"
  run wild PRIOR=smooth POSE_ROT=0.3 POSE_TRANS=2.0 "$@"
  run kitti PRIOR=kitti POSE_KITTI=1.0 "$@"
  run kitti2 PRIOR=kitti POSE_KITTI=2.0 "$@"
}
{
suite A=1
echo "#### forward 1440 workgroups"
suite MD_COSTVOL_NWG=1440
echo "#### counters"
suite MD_CV_STATS=1
echo "#### per-workgroup records"
W="MD_CV_STATS=1 MD_CV_WGSTATS=1 MOVEDEPTH_HIP_LIB=build_ab/libmd_wgstats.so"
run moderate PRIOR=smooth POSE_ROT=0.05 POSE_TRANS=0.3 $W MD_CV_WGSTATS_DUMP=$O/moderate
run kitti PRIOR=kitti POSE_KITTI=1.0 $W MD_CV_WGSTATS_DUMP=$O/kitti
} > $O/cases.txt 2>&1
cat $O/cases.txt
