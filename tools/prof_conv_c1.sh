#!/bin/bash
# rocprofv3 kernel-trace averages of the conv3d_c1 kernels (stand-alone, config-2 volume).  Usage: tools/prof_conv_c1.sh [tag]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-run}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc_$tag
NO_LIB=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_$tag -o c -- python $R/tools/bench_conv3d_c1.py 2>/dev/null | grep -v "^W2\|^E2"
f=$(find /tmp/pc_$tag -name "c_kernel_stats.csv" | head -1)
echo "--- $tag: kernel, calls, avg ns, min ns, max ns"
python - "$f" <<PY
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "conv3d_c1" in r["Name"]:
        print("%-70s calls %4s  avg %7.1f us  min %7.1f  max %7.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
