#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_bf3pmc; mkdir -p $O
SCRIPTS=bench_conv3d_c16 bash tools/pmc_conv.sh $O/c16_bf3.txt > /dev/null 2>&1
cat $O/c16_bf3.txt
# also the 16->1 forward, for VERDICT item 6
SCRIPTS=bench_conv3d_c1 bash tools/pmc_conv.sh $O/c1.txt > /dev/null 2>&1
cat $O/c1.txt
