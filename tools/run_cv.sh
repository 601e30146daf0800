# round-2 closing measurements (scratch driver; outputs under gpurun_out/)
O=$GRAFT_REPO_ROOT/gpurun_out
python bench.py > $O/bench_line.json 2>$O/bench_line.err; tail -c 600 $O/bench_line.json; echo
python bench.py --steps 30 --warmup 60 --no_cpu_baseline --trainer_args="--res_arch 50 --height 320 --width 1024 --num_depth_bins 128 --amp bf16" > $O/bench_line_cfg4.json 2>$O/cfg4.err; cut -c1-700 $O/bench_line_cfg4.json; echo
python bench.py --steps 30 --warmup 30 --no_cpu_baseline --epoch 9 --trainer_args="--frame_ids 0 -2 -1 1 --matching_ids 0 -2 -1 1 --amp fp16" > $O/bench_line_cfg5.json 2>$O/cfg5.err; cut -c1-700 $O/bench_line_cfg5.json; echo
SCRIPTS=bench_conv3d_c1 tools/pmc_conv.sh $O/conv_c1_pmc.txt > /dev/null 2>&1
tools/prof_conv_c1.sh gen2 > $O/conv_c1_gen2.txt 2>&1; MD_CONV3D_C1_GEN1=1 tools/prof_conv_c1.sh gen1 > $O/conv_c1_gen1.txt 2>&1
tail -6 $O/conv_c1_gen2.txt
