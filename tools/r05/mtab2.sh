#!/bin/bash
# Round 5: merge table against the queue-only kernel in the same call; per-workgroup records of both.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_mtab2; mkdir -p $O
B="timeout 300 python tools/bench_costvol.py --layout ndhwc --feat nhwc"
run() { local tag=$1; shift; echo "== $tag"; env "$@" $B 2>&1 | grep "kernel only\|Error\|error\|per-workgroup\|mean \|least\|stats" | sed 's/(dispatch start.stop events inside the library) //'; }
suite() {
  run sane PRIOR=smooth "$@"
  run moderate PRIOR=smooth POSE_ROT=0.05 POSE_TRANS=0.3 "$@"
  run wild PRIOR=smooth POSE_ROT=0.3 POSE_TRANS=2.0 "$@"
  run kitti PRIOR=kitti POSE_KITTI=1.0 "$@"
  run kitti2 PRIOR=kitti POSE_KITTI=2.0 "$@"
}
{
echo "#### queue only (HEAD)"; suite MOVEDEPTH_HIP_LIB=build_ab/libmd_head.so
echo "#### table"; suite A=1
for v in head new; do
echo "#### per-workgroup records: $v"
W="MD_CV_STATS=1 MD_CV_WGSTATS=1 MOVEDEPTH_HIP_LIB=build_ab/libmd_wg_$v.so"
run moderate PRIOR=smooth POSE_ROT=0.05 POSE_TRANS=0.3 $W MD_CV_WGSTATS_DUMP=$O/moderate_$v
run kitti PRIOR=kitti POSE_KITTI=1.0 $W MD_CV_WGSTATS_DUMP=$O/kitti_$v
done
} > $O/cases.txt 2>&1
cat $O/cases.txt
