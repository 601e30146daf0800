#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_bf3a; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "conv0" > $O/pytest_conv0.log 2>&1; echo "pytest rc $?" >> $O/pytest_conv0.log
tail -15 $O/pytest_conv0.log
for v in 1 0; do echo "== MD_C16_BF3=$v"; MD_C16_BF3=$v NO_LIB=1 timeout 600 python tools/bench_conv3d_c16.py 2>&1 | grep -v amdgpu.ids; done
