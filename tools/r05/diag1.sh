#!/bin/bash
# Round-5 diagnostics, first GPU call: where the plane sweep's time goes under parallax (workgroup lifetimes, finer grids, timeline build),
# bench.py's N = 8 branch on one shared GPU over gloo, photometric PMC passes on the current sources.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_diag1; mkdir -p $O
B="timeout 300 python tools/bench_costvol.py --layout ndhwc --feat nhwc"
run() {  # tag, env..., then args after --
  local tag=$1; shift
  echo "== $tag"
  env "$@" MD_CV_STATS=1 $B 2>&1 | grep "kernel only\|stats\|lifetimes\|timeline" | sed 's/(dispatch start.stop events inside the library) //'
}
{
for g in "" "MD_COSTVOL_NWG_BWD=1440 MD_COSTVOL_NWG=1440" "MD_COSTVOL_NWG_BWD=2880 MD_COSTVOL_NWG=2880"; do
  echo "#### grid: ${g:-default}"
  run sane PRIOR=smooth A=1 $g
  run moderate PRIOR=smooth POSE_ROT=0.05 POSE_TRANS=0.3 $g
  run wild PRIOR=smooth POSE_ROT=0.3 POSE_TRANS=2.0 $g
  run kitti PRIOR=kitti POSE_KITTI=1.0 $g
done
echo "#### timeline build (bwd phases, thread 0 of each workgroup)"
for c in "PRIOR=smooth A=1" "PRIOR=smooth POSE_ROT=0.05 POSE_TRANS=0.3" "PRIOR=kitti POSE_KITTI=1.0"; do
  run "timeline $c" $c MD_CV_TIMELINE=1 MOVEDEPTH_HIP_LIB=build_ab/libmd_timeline.so
done
} > $O/costvol_cases.txt 2>&1
cat $O/costvol_cases.txt
if [ "$1" != quick ]; then
MD_SHARE_GPU=1 timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 8 --steps 3 --warmup 2 > $O/bench_n8_shared_gpu_gloo.json 2> $O/n8.err; echo "n8 rc $?"; tail -c 700 $O/bench_n8_shared_gpu_gloo.json; tail -5 $O/n8.err
timeout 1200 bash tools/pmc_photo.sh $O/photo_pmc.txt > $O/pmc_photo.log 2>&1; echo "pmc_photo rc $?"
fi
