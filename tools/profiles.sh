#!/bin/bash
# Profile bundle of a round (run on the GPU box through gpurun): R=r06 tools/profiles.sh <stage ...>   -> gpurun_out/${R}_profiles/${R}_*
#   stages: test bench step pmc parallax photopmc convpmc cfg n8     (copy what is to be judged into profiles/)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=${R:-r06}
O=$GRAFT_REPO_ROOT/gpurun_out/${R}_profiles; mkdir -p $O
for stage in "$@"; do case $stage in
test)
  timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
  timeout 900 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log ;;
bench)
  timeout 1500 python bench.py > $O/${R}_bench_line.json 2> $O/bench.err; echo "bench rc $?"; tail -c 300 $O/${R}_bench_line.json ;;
step)
  timeout 1500 bash tools/profile_step.sh $O/${R}_bench_kernel_stats.csv > $O/${R}_bench_kernel_stats.log 2>&1; grep "^# " $O/${R}_bench_kernel_stats.csv | head -16 ;;
pmc)
  timeout 2400 bash tools/make_profiles.sh $O/mk > $O/make_profiles.log 2>&1
  for f in costvol_pmc_ndhwc.txt costvol_pmc_bgd.txt costvol_kernel_stats_ndhwc.csv bench_costvol_ndhwc.log costvol_fwd_pmc.json; do cp $O/mk/$f $O/${R}_$f 2>/dev/null; done
  cat $O/${R}_costvol_fwd_pmc.json ;;
parallax)
  tools/cv_cases.sh ${R}_cases > /dev/null 2>&1; cp gpurun_out/${R}_cases/cases.txt $O/${R}_costvol_cases.txt; tail -20 $O/${R}_costvol_cases.txt ;;
photopmc)
  timeout 1500 bash tools/pmc_photo.sh $O/${R}_photo_pmc.txt > $O/photopmc.log 2>&1; head -3 $O/${R}_photo_pmc.txt ;;
convpmc)
  SCRIPTS=bench_conv3d_c16 timeout 1200 bash tools/pmc_conv.sh $O/${R}_conv3d_c16_pmc.txt > /dev/null 2>&1; grep "^conv\|=>" $O/${R}_conv3d_c16_pmc.txt
  NO_LIB=1 timeout 600 python tools/bench_conv3d_c16.py 2>&1 | grep "fwd\|bwd-" > $O/${R}_conv3d_c16_standalone.txt; cat $O/${R}_conv3d_c16_standalone.txt ;;
cfg)
  timeout 1500 python bench.py --steps 30 --warmup 30 --no_cpu_baseline --trainer_args="--res_arch 50 --height 320 --width 1024 --num_depth_bins 128 --amp bf16" > $O/${R}_bench_line_cfg4.json 2> $O/cfg4.err; tail -c 300 $O/${R}_bench_line_cfg4.json
  timeout 1500 python bench.py --steps 30 --warmup 30 --no_cpu_baseline --trainer_args="--frame_ids 0 -2 -1 1 --matching_ids 0 -2 -1 1 --amp fp16" > $O/${R}_bench_line_cfg5.json 2> $O/cfg5.err; tail -c 300 $O/${R}_bench_line_cfg5.json ;;
n8)
  MD_SHARE_GPU=1 timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 8 --steps 3 --warmup 2 > $O/${R}_bench_n8_shared_gpu_gloo.json 2> $O/n8.err; echo "n8 rc $?"; tail -c 300 $O/${R}_bench_n8_shared_gpu_gloo.json ;;
esac; done
