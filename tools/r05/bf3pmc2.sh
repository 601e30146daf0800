#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_bf3pmc; mkdir -p $O
SCRIPTS=bench_conv3d_c16 bash tools/pmc_conv.sh $O/c16_bf3_v2.txt > /dev/null 2>&1
grep -A22 "fwd_bf3" $O/c16_bf3_v2.txt | head -26
