python -m pytest tests/test_hip_parity.py tests/test_baseline_configs.py -m gpu -q -k "costvol" 2>&1 | tail -1
run() { echo "=== $*"; env "$@" MD_CV_STATS=1 python tools/bench_costvol.py --layout ndhwc --iters 10 2>&1 | grep -E "kernel only|stats" | cut -c1-200; }
run2() { echo "=== cfg4 $*"; env "$@" MD_CV_STATS=1 python tools/bench_costvol.py --layout ndhwc --iters 10 --h 80 --w 256 --D 128 2>&1 | grep -E "kernel only|stats" | cut -c1-200; }
run PRIOR=smooth
run PRIOR=white
run2 PRIOR=smooth DT=bf16
run2 PRIOR=white DT=bf16
MD_CV_STATS=1 MD_BENCH_DUMP_TIMES=1 python bench.py --steps 6 --warmup 4 --no_cpu_baseline --trainer_args="--res_arch 50 --height 320 --width 1024 --num_depth_bins 128 --amp bf16" 2>&1 >/dev/null | grep -E "costvol" | cut -c1-400
