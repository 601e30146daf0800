#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_cb1; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "conv3d_channel_blocks or conv0" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -25 $O/pytest.log
timeout 300 python tools/r05/cb_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/bench.txt
