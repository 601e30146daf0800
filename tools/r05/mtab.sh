#!/bin/bash
# Round 5: the cell-keyed LDS merge table for gather-mode d_src adds: parity, then the parallax cases.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_mtab; mkdir -p $O
if [ "$1" = test ]; then
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "costvol" > $O/pytest_costvol.log 2>&1; echo "pytest rc $?" >> $O/pytest_costvol.log
tail -3 $O/pytest_costvol.log
fi
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
suite A=1
for c in 16 64 96; do echo "#### chunk $c"; suite MOVEDEPTH_HIP_LIB=build_ab/libmd_gc$c.so; done
echo "#### counters"
run moderate PRIOR=smooth POSE_ROT=0.05 POSE_TRANS=0.3 MD_CV_STATS=1
run wild PRIOR=smooth POSE_ROT=0.3 POSE_TRANS=2.0 MD_CV_STATS=1
} > $O/cases.txt 2>&1
cat $O/cases.txt
