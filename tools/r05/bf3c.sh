#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in "" bf3w1; do
  L=""; [ -n "$v" ] && L="MOVEDEPTH_HIP_LIB=build_ab/libmd_$v.so"
  echo "== ${v:-shipped}"; env $L NO_LIB=1 timeout 600 python tools/bench_conv3d_c16.py 2>&1 | grep "fwd\|bwd-data"
done
