run() { echo "=== $*: $(env "$@" python tools/bench_costvol.py --layout ndhwc --iters 30 2>&1 | grep -E "kernel only" | cut -c62-130 | tr '\n' ' ')"; }
for nwg in 0 720 768 1080 1440; do
run PRIOR=smooth MD_COSTVOL_NWG=$nwg
done
run PRIOR=white MD_COSTVOL_NWG=720
run PRIOR=const PRIOR_CONST=0.21 POSE_TX=0.002 POSE_TZ=0.001 MD_COSTVOL_NWG=720
