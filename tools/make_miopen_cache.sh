#!/bin/bash
# Fills a MIOpen user find-db / kernel cache with the solver-search results of every convolution of the bench workload
# (BASELINE config 2), starting from the in-tree cache.  Run on the GPU box (about 9 minutes), then copy the result into
# movedepth_amd/miopen_cache/.   usage: tools/make_miopen_cache.sh <outdir>
OUT=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $OUT/db $OUT/cache
cp -r $ROOT/movedepth_amd/miopen_cache/db/. $OUT/db/ 2>/dev/null
cp -r $ROOT/movedepth_amd/miopen_cache/cache/. $OUT/cache/ 2>/dev/null
export MIOPEN_USER_DB_PATH=$OUT/db MIOPEN_CUSTOM_CACHE_DIR=$OUT/cache
python $ROOT/bench.py --steps 5 --warmup 3 --no_cpu_baseline --trainer_args="--miopen_find 2" | tail -c 400
python $ROOT/bench.py --steps 20 --warmup 10 --no_cpu_baseline --trainer_args="--miopen_find 2" | tail -c 400
du -sh $OUT/db $OUT/cache; ls -la $OUT/db $OUT/cache | head -30
