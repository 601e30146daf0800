#!/bin/bash
# Round 5: the forward runs 720 half-item slices on 512 slots (placement study: a third of the workgroups start 20-30 us late).  Other grids.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_nwg; mkdir -p $O
B="timeout 300 python tools/bench_costvol.py --layout ndhwc --feat nhwc"
{
for cfg in "" "--dtype f16" "--B 6 --h 80 --w 256 --D 128 --dtype bf16"; do
for n in 0 360 480 512 540 600 720 1024; do
  echo "== shape='${cfg:-fp32}' MD_COSTVOL_NWG=$n"
  env PRIOR=smooth MD_COSTVOL_NWG=$n $B $cfg 2>&1 | grep "kernel only.*fwd" | sed 's/(dispatch start.stop events inside the library) //'
done; done
for n in 0 360 512; do
  echo "== moderate MD_COSTVOL_NWG=$n"; env PRIOR=smooth POSE_ROT=0.05 POSE_TRANS=0.3 MD_COSTVOL_NWG=$n $B 2>&1 | grep "kernel only.*fwd" | sed 's/(dispatch start.stop events inside the library) //'
  echo "== kitti MD_COSTVOL_NWG=$n"; env PRIOR=kitti POSE_KITTI=1.0 MD_COSTVOL_NWG=$n $B 2>&1 | grep "kernel only.*fwd" | sed 's/(dispatch start.stop events inside the library) //'
done
} > $O/nwg.txt 2>&1
cat $O/nwg.txt
