#!/bin/bash
# work donation v3: parity, then parameter sweep on the parallax cases
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_steal3; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "costvol" > $O/pytest_costvol.log 2>&1; echo "pytest rc $?" >> $O/pytest_costvol.log
tail -3 $O/pytest_costvol.log
B="timeout 300 python tools/bench_costvol.py --layout ndhwc --feat nhwc"
run() { local tag=$1; shift; echo "== $tag"; env "$@" $B 2>&1 | grep "kernel only\|Error\|error\|per-workgroup\|mean \|least" | grep -v fwd | sed 's/(dispatch start.stop events inside the library) //'; }
suite() {
  run sane PRIOR=smooth "$@"
  run moderate PRIOR=smooth POSE_ROT=0.05 POSE_TRANS=0.3 "$@"
  run wild PRIOR=smooth POSE_ROT=0.3 POSE_TRANS=2.0 "$@"
  run kitti PRIOR=kitti POSE_KITTI=1.0 "$@"
}
{
echo "#### steal off"; suite MD_COSTVOL_STEAL=0
echo "#### defaults (min 8 max 32 gchunk 16)"; suite A=1
echo "#### donate never (min 1000)"; suite MD_COSTVOL_STEAL_MIN=1000
for mx in 16 96; do echo "#### max $mx"; suite MD_COSTVOL_STEAL_MAX=$mx; done
for mn in 4 16 24; do echo "#### min $mn"; suite MD_COSTVOL_STEAL_MIN=$mn; done
for g in 8 32; do echo "#### gchunk $g"; suite MD_COSTVOL_STEAL_GCHUNK=$g; done
echo "#### per-workgroup records, defaults"
W="MD_CV_STATS=1 MD_CV_WGSTATS=1 MOVEDEPTH_HIP_LIB=build_ab/libmd_wgstats.so"
run moderate PRIOR=smooth POSE_ROT=0.05 POSE_TRANS=0.3 $W
run kitti PRIOR=kitti POSE_KITTI=1.0 $W
} > $O/cases.txt 2>&1
cat $O/cases.txt
