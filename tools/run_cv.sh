run() { echo "=== $*"; env "$@" python tools/bench_costvol.py --layout ndhwc --iters 10 2>&1 | grep -E "kernel only.*bwd" | cut -c60-200; }
run PRIOR=smooth
for v in 4 5; do run PRIOR=smooth MOVEDEPTH_HIP_LIB=$GRAFT_REPO_ROOT/gpurun_probe_lib$v.so; done
