#!/bin/bash
# Plane-sweep timing matrix for one gpurun call (dispatch events inside the library; tools/bench_costvol.py):
#   tools/cv_cases.sh <tag> [lib.so ...]     each lib: a tools/ab_build.sh build (MOVEDEPTH_HIP_LIB); none: the in-tree library
# cases: sane / white / moderate / wild / driving 1 m / driving 2 m  x  fp32, fp16 at config 2's shape; bf16 at config 4's (sane, driving 1 m).
# CASES="sane moderate" DTYPES="f32" narrow it.  Output: gpurun_out/<tag>/cases.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $O
B="timeout 300 python tools/bench_costvol.py --layout ndhwc --feat nhwc"
CASES=${CASES:-"sane white moderate wild kitti1 kitti2"}
DTYPES=${DTYPES:-"f32 f16"}
run_case() {   # $1 = case, rest = bench args
  c=$1; shift
  case $c in
    sane)     env PRIOR=smooth $B "$@";;
    white)    env PRIOR=white $B "$@";;
    moderate) env PRIOR=smooth POSE_ROT=0.05 POSE_TRANS=0.3 $B "$@";;
    wild)     env PRIOR=smooth POSE_ROT=0.3 POSE_TRANS=2.0 $B "$@";;
    kitti1)   env PRIOR=kitti POSE_KITTI=1.0 $B "$@";;
    kitti2)   env PRIOR=kitti POSE_KITTI=2.0 $B "$@";;
  esac
}
{
for lib in "${@:-}"; do
  export MOVEDEPTH_HIP_LIB=$lib
  echo "#### library: ${lib:-in-tree}"
  for dt in $DTYPES; do for c in $CASES; do
    echo "== $dt $c"
    run_case $c --dtype $dt 2>&1 | grep "kernel only\|stats \|backward policy" | sed 's/(dispatch start.stop events inside the library) //'
  done; done
  if [ -z "$NO_CFG4" ]; then for c in sane kitti1; do
    echo "== cfg4 bf16 $c"
    run_case $c --B 6 --h 80 --w 256 --D 128 --dtype bf16 2>&1 | grep "kernel only\|backward policy" | sed 's/(dispatch start.stop events inside the library) //'
  done; fi
done
} > $O/cases.txt 2>&1
cat $O/cases.txt
