#!/bin/bash
# Plane-sweep regression / timing suite for one gpurun call: tools/cv_suite.sh <tag> [test]  (test: also run the costvol parity tests)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
if [ "$2" = test ]; then
  timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "costvol" > $O/pytest_costvol.log 2>&1; echo "pytest rc $?" >> $O/pytest_costvol.log
  tail -2 $O/pytest_costvol.log
fi
B="timeout 300 python tools/bench_costvol.py --layout ndhwc --feat nhwc --prior smooth"
$B > $O/f32.log 2>&1
$B --dtype f16 > $O/f16.log 2>&1
$B --dtype bf16 > $O/bf16.log 2>&1
$B --B 6 --h 80 --w 256 --D 128 --dtype bf16 > $O/cfg4_bf16.log 2>&1
POSE_ROT=0.3 POSE_TRANS=2.0 $B > $O/wild.log 2>&1
POSE_ROT=0.05 POSE_TRANS=0.3 $B > $O/moderate.log 2>&1
for f in f32 f16 bf16 cfg4_bf16 wild moderate; do echo "== $f"; grep -h "kernel only" $O/$f.log | sed 's/(dispatch start.stop events inside the library) //'; done
if [ "$3" = bench ]; then
  timeout 900 python bench.py --steps 30 --warmup 10 --no_cpu_baseline > $O/bench.log 2>&1
  tail -1 $O/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('step', d['ms_per_step'], 'img/s', d['value'], 'fwd', r['avg_launch_us'], r['frac'], 'bwd', r['bwd_avg_launch_us'])"
fi
