#!/bin/bash
# Round-4 profile bundle (run on the GPU box through gpurun): tools/r04_profiles.sh <stage ...>   (stages: bench step pmc wild cfg syncbn syncbn_trace n2)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_profiles; mkdir -p $O
for stage in "$@"; do case $stage in
bench)
  timeout 1200 python bench.py > $O/r04_bench_line.json 2> $O/bench.err; tail -c 400 $O/r04_bench_line.json ;;
step)
  timeout 1500 bash tools/profile_step.sh $O/r04_bench_kernel_stats.csv > $O/r04_bench_kernel_stats.log 2>&1; grep "^# " $O/r04_bench_kernel_stats.csv | head -16 ;;
pmc)
  timeout 2400 bash tools/make_profiles.sh $O/mk > $O/make_profiles.log 2>&1
  for f in costvol_pmc_ndhwc.txt costvol_pmc_bgd.txt costvol_kernel_stats_ndhwc.csv bench_costvol_ndhwc.log costvol_fwd_pmc.json; do cp $O/mk/$f $O/r04_$f 2>/dev/null; done
  cat $O/r04_costvol_fwd_pmc.json ;;
wild)
  { echo "# POSE_ROT / POSE_TRANS = axis-angle ~ N(0, rot^2) rad, translation ~ N(0, trans^2): tools/bench_costvol.py --layout ndhwc --feat {nhwc,nchw} --prior smooth, dispatch events inside the library"
    for feat in nhwc nchw; do for cfg in "" "--B 6 --h 80 --w 256 --D 128 --dtype bf16"; do for pose in "sane" "POSE_ROT=0.05 POSE_TRANS=0.3" "POSE_ROT=0.3 POSE_TRANS=2.0"; do
      echo "== feat=$feat shape='${cfg:-config 2: B=6 48x160 D=96 fp32}' poses: $pose"
      if [ "$pose" = sane ]; then e="A=1"; else e="$pose"; fi
      env $e MD_CV_STATS=1 timeout 300 python tools/bench_costvol.py --layout ndhwc --feat $feat --prior smooth $cfg 2>&1 | grep "kernel only\|stats" | sed 's/(dispatch start.stop events inside the library) //'
    done; done; done; } > $O/r04_wild_pose.txt 2>&1; tail -20 $O/r04_wild_pose.txt ;;
cfg)
  timeout 1500 python bench.py --steps 30 --warmup 30 --no_cpu_baseline --trainer_args="--res_arch 50 --height 320 --width 1024 --num_depth_bins 128 --amp bf16" > $O/r04_bench_line_cfg4.json 2> $O/cfg4.err; tail -c 300 $O/r04_bench_line_cfg4.json
  timeout 1500 python bench.py --steps 30 --warmup 30 --no_cpu_baseline --trainer_args="--frame_ids 0 -2 -1 1 --matching_ids 0 -2 -1 1 --amp fp16" > $O/r04_bench_line_cfg5.json 2> $O/cfg5.err; tail -c 300 $O/r04_bench_line_cfg5.json ;;
syncbn)
  { for v in "" "--force_sync_bn 1" "--force_sync_bn 1 --sync_bn_impl hip"; do :; done
    echo "# per-rank cost of synchronised BatchNorm, measured on one GPU: bench.py --steps 30 --warmup 10 --no_cpu_baseline, interleaved"
    for rep in 1 2; do
      echo "== plain step (library BatchNorm)"; timeout 900 python bench.py --steps 30 --warmup 10 --no_cpu_baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.2f ms/step %.1f images/s'%(d['ms_per_step'], d['value']))"
      echo "== --force_sync_bn 1 (networks.HipSyncBatchNorm, csrc/syncbn.hip, ReLU fused; a group of one)"; timeout 900 python bench.py --steps 30 --warmup 10 --no_cpu_baseline --trainer_args="--force_sync_bn 1" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.2f ms/step %.1f images/s'%(d['ms_per_step'], d['value'])); print('   ', {k:(round(v['us_per_step'],1), v['dispatches_per_step']) for k,v in d['photometric_kernels_in_step'].items() if 'bn' in k})"
      echo "== every BatchNorm on torch's native kernels (what torch.nn.SyncBatchNorm is built from; tools/micro/bn_native_step.py)"; timeout 900 python tools/micro/bn_native_step.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.2f ms/step %.1f images/s'%(d['ms_per_step'], d['value']))"
    done
    echo "# kernel time per layer shape, forward + backward (tools/bench_bn.py)"
    python tools/bench_bn.py 2>&1 | grep -v "amdgpu.ids\|cuDNN\|benchmark_limit"
    echo "# host-side cost per call (tools/micro/bn_host_overhead.py)"
    python tools/micro/bn_host_overhead.py 2>&1 | grep -v amdgpu.ids; } > $O/r04_syncbn.txt 2>&1
  TRAINER_ARGS="--force_sync_bn 1" timeout 1500 bash tools/profile_step.sh $O/r04_bench_kernel_stats_syncbn.csv > $O/r04_bench_kernel_stats_syncbn.log 2>&1
  head -14 $O/r04_syncbn.txt ;;
syncbn_trace)
  TRAINER_ARGS="--force_sync_bn 1" timeout 1500 bash tools/profile_step.sh $O/r04_bench_kernel_stats_syncbn.csv > $O/r04_bench_kernel_stats_syncbn.log 2>&1
  grep "^# " $O/r04_bench_kernel_stats_syncbn.csv | head -16 ;;
n2)
  MD_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 3 > $O/r04_bench_n2_shared_gpu_gloo.json 2> $O/n2.err; tail -c 600 $O/r04_bench_n2_shared_gpu_gloo.json ;;
esac; done
