#!/bin/bash
# rocprofv3 kernel-trace stats of the training step (bench.py, config 2).  usage: tools/profile_step.sh <out.csv>
OUT=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_step -o s -- python $ROOT/bench.py --steps 10 --warmup 3 --no_cpu_baseline > /tmp/prof_step.log 2>&1
tail -2 /tmp/prof_step.log | cut -c1-300
python - <<PY
import csv
rows = list(csv.DictReader(open("/tmp/prof_step/s_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open("$OUT", "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 3 (13 steps incl. warm-up + MIOpen find)\n")
    f.write("# total kernel time %.1f ms\n" % (tot / 1e6))
    f.write("name,calls,total_ms,avg_us,min_us,max_us,percent\n")
    mine = ("costvol", "warp_", "ssim_", "reproj_", "masked_min", "smooth_", "sel_", "schedule_", "fuse_", "disp_up")
    keep = rows[:40] + [r for r in rows[40:] if any(m in r["Name"] for m in mine)]
    for r in keep:
        f.write("\"%s\",%s,%.3f,%.2f,%.2f,%.2f,%s\n" % (r["Name"][:140], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
print(open("$OUT").read()[:6000])
PY
