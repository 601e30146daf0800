#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_full1; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout 900 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log
timeout 1500 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc $?"
python - <<PY
import json
d=json.loads(open("$O/bench_line.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("step %.2f ms, %.1f images/s; fwd %.1f us (%.3f), bwd %.1f us" % (d["ms_per_step"], d["value"], r["avg_launch_us"], r["frac"], r["bwd_avg_launch_us"]))
for k,v in r.get("parallax_cases",{}).get("cases",{}).items(): print("  ", k[:40], "fwd %.1f us %.3f  bwd %.1f us %.3f" % (v["fwd_avg_us"], v["fwd_frac"], v["bwd_avg_us"], v["bwd_frac"]))
print({k: round(v["avg_us"],1) for k,v in d.get("reg3d_handoff_kernels",{}).items()})
print({k: round(v["us_per_step"],1) for k,v in d.get("photometric_kernels_in_step",{}).items()})
PY
