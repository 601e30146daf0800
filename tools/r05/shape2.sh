#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_shape2; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "costvol" > $O/pytest_costvol.log 2>&1; echo "pytest rc $?" >> $O/pytest_costvol.log
tail -3 $O/pytest_costvol.log
MD_COSTVOL_BWD_SHAPE0=1 timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "costvol" > $O/pytest_costvol1.log 2>&1; echo "pytest rc $?" >> $O/pytest_costvol1.log
tail -3 $O/pytest_costvol1.log
B="timeout 300 python tools/bench_costvol.py --layout ndhwc --feat nhwc"
run() { local tag=$1; shift; echo "== $tag"; env "$@" $B 2>&1 | grep "kernel only\|Error\|error" | grep bwd | sed 's/(dispatch start.stop events inside the library) //'; }
{
for rep in 1 2; do for v in 0 1 2; do
echo "#### first shape $v (0 = 22 x 6, 1 = 19 x 7, 2 = 26 x 5), repetition $rep"
L="MD_COSTVOL_BWD_SHAPE0=$v"
run sane PRIOR=smooth $L
run white PRIOR=white $L
run moderate PRIOR=smooth POSE_ROT=0.05 POSE_TRANS=0.3 $L
run wild PRIOR=smooth POSE_ROT=0.3 POSE_TRANS=2.0 $L
run kitti PRIOR=kitti POSE_KITTI=1.0 $L
run kitti2 PRIOR=kitti POSE_KITTI=2.0 $L
run "f16 sane" PRIOR=smooth DT=f16 $L
B0=$B; B="$B --B 6 --h 80 --w 256 --D 128 --dtype bf16"
run "cfg4 sane" PRIOR=smooth $L
B=$B0
done; done
} > $O/cases.txt 2>&1
cat $O/cases.txt
