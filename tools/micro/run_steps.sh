#!/bin/bash
# swap in an experimental library build and get true kernel durations with rocprofv3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cp $ROOT/movedepth_amd/libmovedepth_hip.so /tmp/lib_orig.so
for n in 0 8 24; do
  cp $ROOT/tools/micro/libmd_steps$n.so $ROOT/movedepth_amd/libmovedepth_hip.so
  for lay in bgd ndhwc; do
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$n$lay -o c -- python $ROOT/tools/bench_costvol.py --layout $lay --iters 20 > /dev/null 2>&1
    python -c "
import csv
for r in csv.DictReader(open('/tmp/ps_$n$lay/c_kernel_stats.csv')):
    if 'costvol_fwd' in r['Name']: print('steps=$n layout=$lay fwd avg %.1f us min %.1f' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
"
  done
done
cp /tmp/lib_orig.so $ROOT/movedepth_amd/libmovedepth_hip.so
