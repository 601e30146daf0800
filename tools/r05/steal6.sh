#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_steal6; mkdir -p $O
B="timeout 300 python tools/bench_costvol.py --layout ndhwc --feat nhwc"
run() { local tag=$1; shift; echo "== $tag"; env "$@" $B 2>&1 | grep "kernel only\|Error\|error\|per-workgroup\|mean \|least" | grep -v fwd | sed 's/(dispatch start.stop events inside the library) //'; }
{
W="MD_CV_STATS=1 MD_CV_WGSTATS=1 MOVEDEPTH_HIP_LIB=build_ab/libmd_wgstats.so"
run "moderate max 64 gchunk 32" PRIOR=smooth POSE_ROT=0.05 POSE_TRANS=0.3 MD_COSTVOL_STEAL_MAX=64 MD_COSTVOL_STEAL_GCHUNK=32 $W MD_CV_WGSTATS_DUMP=$O/moderate
run "kitti max 64 gchunk 32" PRIOR=kitti POSE_KITTI=1.0 MD_COSTVOL_STEAL_MAX=64 MD_COSTVOL_STEAL_GCHUNK=32 $W MD_CV_WGSTATS_DUMP=$O/kitti
run "kitti off" PRIOR=kitti POSE_KITTI=1.0 MD_COSTVOL_STEAL=0 $W MD_CV_WGSTATS_DUMP=$O/kitti_off
} > $O/cases.txt 2>&1
cat $O/cases.txt
