#!/bin/bash
# PMC counters of the plane sweep under parallax: sane / moderate / kitti, donation off / on
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_pmc1; mkdir -p $O
A="--layout ndhwc --feat nhwc"
PMC_PASSES=3 MD_COSTVOL_STEAL=0 PRIOR=smooth bash tools/pmc_costvol.sh $O/sane_off.txt $A > /dev/null 2>&1
PMC_PASSES=3 MD_COSTVOL_STEAL=0 PRIOR=smooth POSE_ROT=0.05 POSE_TRANS=0.3 bash tools/pmc_costvol.sh $O/moderate_off.txt $A > /dev/null 2>&1
PMC_PASSES=3 MD_COSTVOL_STEAL=1 PRIOR=smooth POSE_ROT=0.05 POSE_TRANS=0.3 bash tools/pmc_costvol.sh $O/moderate_on.txt $A > /dev/null 2>&1
PMC_PASSES=3 MD_COSTVOL_STEAL=0 PRIOR=kitti POSE_KITTI=1.0 bash tools/pmc_costvol.sh $O/kitti_off.txt $A > /dev/null 2>&1
PMC_PASSES=3 MD_COSTVOL_STEAL=0 PRIOR=smooth POSE_ROT=0.3 POSE_TRANS=2.0 bash tools/pmc_costvol.sh $O/wild_off.txt $A > /dev/null 2>&1
for f in sane_off moderate_off moderate_on kitti_off wild_off; do echo "=== $f"; cat $O/$f.txt; done
