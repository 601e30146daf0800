#!/bin/bash
# Round-5 profile bundle (run on the GPU box through gpurun): tools/r05_profiles.sh <stage ...>
#   stages: test bench step pmc parallax photopmc convpmc cfg n8
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_profiles; mkdir -p $O
for stage in "$@"; do case $stage in
test)
  timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
  timeout 900 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log ;;
bench)
  timeout 1500 python bench.py > $O/r05_bench_line.json 2> $O/bench.err; echo "bench rc $?"; tail -c 300 $O/r05_bench_line.json ;;
step)
  timeout 1500 bash tools/profile_step.sh $O/r05_bench_kernel_stats.csv > $O/r05_bench_kernel_stats.log 2>&1; grep "^# " $O/r05_bench_kernel_stats.csv | head -16 ;;
pmc)
  timeout 2400 bash tools/make_profiles.sh $O/mk > $O/make_profiles.log 2>&1
  for f in costvol_pmc_ndhwc.txt costvol_pmc_bgd.txt costvol_kernel_stats_ndhwc.csv bench_costvol_ndhwc.log costvol_fwd_pmc.json; do cp $O/mk/$f $O/r05_$f 2>/dev/null; done
  cat $O/r05_costvol_fwd_pmc.json ;;
parallax)
  { echo "# Final kernels of round 5: tools/bench_costvol.py --layout ndhwc --feat nhwc, dispatch events inside the library; MD_CV_STATS=1 counters"
    B="timeout 300 python tools/bench_costvol.py --layout ndhwc --feat nhwc"
    for cfg in "" "--dtype f16" "--B 6 --h 80 --w 256 --D 128 --dtype bf16"; do
      for c in "sane|PRIOR=smooth" "white-noise prior|PRIOR=white" "moderate (POSE_ROT=0.05 POSE_TRANS=0.3)|PRIOR=smooth POSE_ROT=0.05 POSE_TRANS=0.3" "wild (POSE_ROT=0.3 POSE_TRANS=2.0)|PRIOR=smooth POSE_ROT=0.3 POSE_TRANS=2.0" "driving scene 1 m per frame|PRIOR=kitti POSE_KITTI=1.0" "driving scene 2 m per frame|PRIOR=kitti POSE_KITTI=2.0"; do
        echo "== shape='${cfg:-config 2: B=6 48x160 D=96 fp32}' case: ${c%%|*}"
        env ${c##*|} MD_CV_STATS=1 $B $cfg 2>&1 | grep "kernel only\|stats" | sed 's/(dispatch start.stop events inside the library) //'
      done
    done; } > $O/r05_parallax_final.txt 2>&1; tail -30 $O/r05_parallax_final.txt ;;
photopmc)
  timeout 1500 bash tools/pmc_photo.sh $O/r05_photo_pmc.txt > $O/photopmc.log 2>&1; head -3 $O/r05_photo_pmc.txt ;;
convpmc)
  SCRIPTS=bench_conv3d_c16 timeout 1200 bash tools/pmc_conv.sh $O/r05_conv3d_c16_pmc.txt > /dev/null 2>&1; grep "^conv\|=>" $O/r05_conv3d_c16_pmc.txt
  { for v in "MD_C16_BF3=1 MD_C16_BF3_WGRAD=1" "MD_C16_BF3=0 MD_C16_BF3_WGRAD=0"; do echo "== $v"; env $v NO_LIB=1 timeout 600 python tools/bench_conv3d_c16.py 2>&1 | grep "fwd\|bwd-"; done; } > $O/r05_conv3d_c16_standalone.txt 2>&1; cat $O/r05_conv3d_c16_standalone.txt ;;
cfg)
  timeout 1500 python bench.py --steps 30 --warmup 30 --no_cpu_baseline --trainer_args="--res_arch 50 --height 320 --width 1024 --num_depth_bins 128 --amp bf16" > $O/r05_bench_line_cfg4.json 2> $O/cfg4.err; tail -c 300 $O/r05_bench_line_cfg4.json
  timeout 1500 python bench.py --steps 30 --warmup 30 --no_cpu_baseline --trainer_args="--frame_ids 0 -2 -1 1 --matching_ids 0 -2 -1 1 --amp fp16" > $O/r05_bench_line_cfg5.json 2> $O/cfg5.err; tail -c 300 $O/r05_bench_line_cfg5.json ;;
n8)
  MD_SHARE_GPU=1 timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 8 --steps 3 --warmup 2 > $O/r05_bench_n8_shared_gpu_gloo.json 2> $O/n8.err; echo "n8 rc $?"; tail -c 300 $O/r05_bench_n8_shared_gpu_gloo.json ;;
esac; done
