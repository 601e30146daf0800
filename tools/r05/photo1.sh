#!/bin/bash
# Round 5: photometric forward -- exact /3 by Markstein's quotient, 3 against 4 waves per SIMD; bit-equality tests first.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_photo1; mkdir -p $O
timeout 1200 python -m pytest tests/test_photo_fused.py -x -q -m gpu > $O/pytest_photo.log 2>&1; echo "pytest rc $?" >> $O/pytest_photo.log; tail -3 $O/pytest_photo.log
{
for v in "" build_ab/libmd_head.so build_ab/libmd_pw4.so; do
echo "#### ${v:-in-tree}"
MOVEDEPTH_HIP_LIB=$v timeout 300 python tools/bench_photo.py --unfused 0 2>&1 | grep -v amdgpu.ids
done
} > $O/photo.txt 2>&1
cat $O/photo.txt
