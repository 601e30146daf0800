#!/bin/bash
# Round 5: the cell-table backward at config 4's shape, driving scene
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_mtab7; mkdir -p $O
B="timeout 300 python tools/bench_costvol.py --layout ndhwc --feat nhwc"
{
for cfg in "--B 6 --h 80 --w 256 --D 128 --dtype bf16" "--B 6 --h 80 --w 256 --D 128"; do
for t in 0 1; do
for c in "sane|PRIOR=smooth" "white|PRIOR=white" "kitti|PRIOR=kitti POSE_KITTI=1.0" "kitti2|PRIOR=kitti POSE_KITTI=2.0"; do
  echo "== shape='$cfg' MD_COSTVOL_GATHER_TABLE=$t case: ${c%%|*}"
  env ${c##*|} MD_COSTVOL_GATHER_TABLE=$t $B $cfg 2>&1 | grep "kernel only.*bwd" | sed 's/(dispatch start.stop events inside the library) //'
done; done; done
} > $O/cases.txt 2>&1
cat $O/cases.txt
