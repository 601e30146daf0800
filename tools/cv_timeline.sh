#!/bin/bash
# Per-phase cycles of thread 0 of every workgroup (a -DMD_CL_TIMELINE=1 build: tools/ab_build.sh tl -DMD_CL_TIMELINE=1), summed over the
# launch's workgroups: tools/cv_timeline.sh <tag> [bench_costvol args]   forward: set-up, staging, walk, redo+barrier; backward: see MD_TL
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=$1; shift
O=gpurun_out/$tag; mkdir -p $O
for dt in f32 f16; do for c in "PRIOR=smooth" "PRIOR=kitti POSE_KITTI=1.0" "PRIOR=smooth POSE_ROT=0.05 POSE_TRANS=0.3"; do
  echo "== $dt $c"
  env $c MD_CV_STATS=1 MD_CV_TIMELINE=1 MOVEDEPTH_HIP_LIB=$GRAFT_REPO_ROOT/build_ab/libmd_tl.so timeout 300 python tools/bench_costvol.py --layout ndhwc --feat nhwc --dtype $dt "$@" 2>&1 | grep "timeline\|kernel only\|lifetimes"
done; done > $O/timeline.txt 2>&1
cat $O/timeline.txt
