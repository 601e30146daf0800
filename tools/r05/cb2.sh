#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_cb2; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "conv3d_channel_blocks or conv0" > $O/pytest_cb.log 2>&1; tail -2 $O/pytest_cb.log; timeout 300 python tools/r05/cb_bench.py 2>&1 | grep -v amdgpu.ids | head -1
timeout 1500 python -m pytest tests/test_trainer_parity.py tests/test_baseline_configs.py tests/test_step_golden.py tests/test_dp_trainer_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 900 python bench.py --no_cpu_baseline > $O/bench_on.json 2> $O/bench_on.err; tail -c 200 $O/bench_on.json; echo


python - <<PY
import json
for f in ("bench_on","bench_off","bench_on2"):
    try:
        d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["value"])
    except Exception as e: print(f, "ERR", e)
PY
