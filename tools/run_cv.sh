run() { echo "=== $*"; env "$@" python tools/bench_costvol.py --layout ndhwc --iters 10 2>&1 | grep -E "kernel only.*fwd" | cut -c60-200; }
run PRIOR=smooth
run PRIOR=smooth MD_COSTVOL_NWG=1080
run PRIOR=smooth MD_COSTVOL_NWG=1440
run PRIOR=smooth MD_COSTVOL_NWG=512
run PRIOR=smooth MD_COSTVOL_NWG=360
run PRIOR=smooth
