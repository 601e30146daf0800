#!/bin/bash
# Round 5: which gathered ranges should use the cell table?  MD_COSTVOL_GATHER_TABLE = footprint / window factor above which they do.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_mtab3; mkdir -p $O
if [ "$1" = test ]; then
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "costvol" > $O/pytest_costvol.log 2>&1; echo "pytest rc $?" >> $O/pytest_costvol.log
tail -3 $O/pytest_costvol.log
fi
B="timeout 300 python tools/bench_costvol.py --layout ndhwc --feat nhwc"
run() { local tag=$1; shift; echo "== $tag"; env "$@" $B 2>&1 | grep "kernel only.*bwd\|Error\|error" | sed 's/(dispatch start.stop events inside the library) //'; }
suite() {
  run sane PRIOR=smooth "$@"
  run moderate PRIOR=smooth POSE_ROT=0.05 POSE_TRANS=0.3 "$@"
  run wild PRIOR=smooth POSE_ROT=0.3 POSE_TRANS=2.0 "$@"
  run kitti PRIOR=kitti POSE_KITTI=1.0 "$@"
  run kitti2 PRIOR=kitti POSE_KITTI=2.0 "$@"
}
{
for t in ${TABS:-0 1.5 2 3 4 6 1e9}; do echo "#### table above $t x window"; suite MD_COSTVOL_GATHER_TABLE=$t; done
} > $O/cases.txt 2>&1
cat $O/cases.txt
