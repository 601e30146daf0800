#!/bin/bash
# Round-5 diagnostics 2: per-workgroup records (-DMD_CL_WGSTATS=1 build) of the plane sweep under parallax; gather-slack sweep.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_diag2; mkdir -p $O
B="timeout 300 python tools/bench_costvol.py --layout ndhwc --feat nhwc"
run() { local tag=$1; shift; echo "== $tag"; env "$@" $B 2>&1 | grep "kernel only\|stats\|lifetimes\|timeline\|per-workgroup\|mean \|least squares" | sed 's/(dispatch start.stop events inside the library) //'; }
{
W="MD_CV_STATS=1 MD_CV_WGSTATS=1 MOVEDEPTH_HIP_LIB=build_ab/libmd_wgstats.so"
run sane PRIOR=smooth A=1 $W MD_CV_WGSTATS_DUMP=$O/sane
run moderate PRIOR=smooth POSE_ROT=0.05 POSE_TRANS=0.3 $W MD_CV_WGSTATS_DUMP=$O/moderate
run wild PRIOR=smooth POSE_ROT=0.3 POSE_TRANS=2.0 $W MD_CV_WGSTATS_DUMP=$O/wild
run kitti PRIOR=kitti POSE_KITTI=1.0 $W MD_CV_WGSTATS_DUMP=$O/kitti
echo "#### gather slack (shipped library)"
for sl in 1.0 1.5 3.0 1000; do
  run "moderate slack $sl" PRIOR=smooth POSE_ROT=0.05 POSE_TRANS=0.3 MD_COSTVOL_GATHER_SLACK=$sl
  run "kitti slack $sl" PRIOR=kitti POSE_KITTI=1.0 MD_COSTVOL_GATHER_SLACK=$sl
done
echo "#### forward grids, interleaved x2"
for rep in 1 2; do for g in 0 1440; do
  run "sane NWG=$g" PRIOR=smooth MD_COSTVOL_NWG=$g
  run "white NWG=$g" PRIOR=white MD_COSTVOL_NWG=$g
  run "moderate NWG=$g" PRIOR=smooth POSE_ROT=0.05 POSE_TRANS=0.3 MD_COSTVOL_NWG=$g
done; done
} > $O/cases.txt 2>&1
cat $O/cases.txt
