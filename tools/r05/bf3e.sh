#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_bf3a; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "conv0" > $O/pytest_conv0.log 2>&1; echo "pytest rc $?" >> $O/pytest_conv0.log
tail -12 $O/pytest_conv0.log
for v in 1 0; do echo "== MD_C16_BF3_WGRAD=$v"; MD_C16_BF3_WGRAD=$v NO_LIB=1 timeout 600 python tools/bench_conv3d_c16.py 2>&1 | grep "fwd\|bwd-"; done
