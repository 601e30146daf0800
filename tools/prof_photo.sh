#!/bin/bash
# kernel-trace stats of tools/bench_photo.py (per-kernel durations by grid size).  usage: tools/prof_photo.sh
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pp; rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -o p -- python $ROOT/tools/bench_photo.py --iters 10 --unfused 0 "$@" > /tmp/pp.log 2>&1
python - <<PY
import csv, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open("/tmp/pp/p_kernel_trace.csv")):
    n = r["Kernel_Name"]
    if "photo_" in n or "up_adjoint" in n or "smooth" in n:
        k = n.split("(anonymous namespace)::")[-1].split("(")[0] + " grid=%s" % r["Grid_Size_X"] if "Grid_Size_X" in r else n
        d[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(d):
    v = d[k]
    print("%-70s calls %4d avg %8.1f us  min %8.1f" % (k, len(v), sum(v) / len(v), min(v)))
PY
