#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_place; mkdir -p $O
B="timeout 300 python tools/bench_costvol.py --layout ndhwc --feat nhwc"
W="MD_CV_STATS=1 MD_CV_WGSTATS=1 MOVEDEPTH_HIP_LIB=build_ab/libmd_place.so"
{
env PRIOR=smooth $W MD_CV_WGSTATS_DUMP=$O/sane $B 2>&1 | grep "kernel only"
python tools/r05/place.py $O/sane_fwd.npy $O/sane_bwd.npy
env PRIOR=smooth $W MD_CV_WGSTATS_DUMP=$O/sane16 $B --dtype f16 2>&1 | grep "kernel only"
python tools/r05/place.py $O/sane16_fwd.npy
} > $O/place.txt 2>&1
cat $O/place.txt
