#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_shape3; mkdir -p $O
B="timeout 300 python tools/bench_costvol.py --layout ndhwc --feat nhwc"
run() { local tag=$1; shift; echo "== $tag"; env "$@" $B 2>&1 | grep "kernel only\|Error\|error\|per-workgroup\|mean \|least" | grep -v fwd | sed 's/(dispatch start.stop events inside the library) //'; }
{
W="MD_CV_STATS=1 MD_CV_WGSTATS=1 MOVEDEPTH_HIP_LIB=build_ab/libmd_wgstats.so"
run "kitti shape0=0" PRIOR=kitti POSE_KITTI=1.0 $W MD_CV_WGSTATS_DUMP=$O/kitti0
run "kitti shape0=1" PRIOR=kitti POSE_KITTI=1.0 MD_COSTVOL_BWD_SHAPE0=1 $W MD_CV_WGSTATS_DUMP=$O/kitti1
} > $O/cases.txt 2>&1
cat $O/cases.txt
