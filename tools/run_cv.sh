python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "costvol" 2>&1 | tail -3
run() { echo "=== $*: $(env "$@" python tools/bench_costvol.py --layout ndhwc --iters 20 2>&1 | grep -E "fwd|bwd" | cut -c1-22 | tr '\n' ' ')"; }
for cl in 0 1; do
for pc in 0.21 8.0 32.0; do for tx in 0.0 0.05; do
run MD_COSTVOL_CL=$cl PRIOR_CONST=$pc POSE_TX=$tx POSE_TZ=0.0 PRIOR=const
done; done
run MD_COSTVOL_CL=$cl PRIOR=white
run MD_COSTVOL_CL=$cl PRIOR=smooth
done
