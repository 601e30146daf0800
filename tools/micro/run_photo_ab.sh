#!/bin/bash
# A/B of photometric-kernel builds on tools/bench_photo.py: tools/micro/run_photo_ab.sh <out dir name> <lib names under build_ab/ ...; "new" = in-tree>
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift
mkdir -p $O
for n in "$@"; do
  if [ $n = new ]; then L=movedepth_amd/libmovedepth_hip.so; else L=build_ab/libmd_$n.so; fi
  echo "== $n" >> $O/bench.txt
  MOVEDEPTH_HIP_LIB=$L timeout 300 python tools/bench_photo.py --unfused 0 --iters 30 2>&1 | tail -4 >> $O/bench.txt
done
cat $O/bench.txt
