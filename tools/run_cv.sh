python -m pytest tests/test_hip_parity.py tests/test_baseline_configs.py -m gpu -q -k "costvol" 2>&1 | grep -a "passed\|failed" | tail -1
run() { echo "=== $*"; env "$@" python tools/bench_costvol.py --layout ndhwc --iters 10 2>&1 | grep -E "kernel only" | cut -c1-200; }
run PRIOR=smooth
run PRIOR=smooth MD_COSTVOL_TWO_PHASE=0
run PRIOR=white
run PRIOR=white MD_COSTVOL_TWO_PHASE=0
run PRIOR=smooth
run PRIOR=smooth MD_COSTVOL_TWO_PHASE=0
python bench.py --steps 30 --warmup 15 --no_cpu_baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('bench', d['value'], d['ms_per_step'], 'fwd', r['avg_launch_us'], 'bwd', r['bwd_avg_launch_us'])"
