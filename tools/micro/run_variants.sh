#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cp $ROOT/movedepth_amd/libmovedepth_hip.so /tmp/lib_orig.so
for v in "$@"; do
  cp $ROOT/tools/micro/libmd_$v.so $ROOT/movedepth_amd/libmovedepth_hip.so
  for lay in ndhwc; do echo "variant=$v layout=$lay"; python $ROOT/tools/bench_costvol.py --layout $lay 2>/dev/null | grep -E "fwd"; done
done
cp /tmp/lib_orig.so $ROOT/movedepth_amd/libmovedepth_hip.so
