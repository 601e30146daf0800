#!/bin/bash
# Round profile bundle (run on the GPU box): kernel-trace stats of the bench step + PMC passes of the plane-sweep
# kernels in both layouts.  usage: tools/make_profiles.sh <outdir>
OUT=$1; mkdir -p $OUT
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
$ROOT/tools/profile_step.sh $OUT/bench_kernel_stats.csv > $OUT/bench_kernel_stats.log 2>&1
$ROOT/tools/pmc_costvol.sh $OUT/costvol_pmc_ndhwc.txt --layout ndhwc > /dev/null 2>&1
$ROOT/tools/pmc_costvol.sh $OUT/costvol_pmc_bgd.txt --layout bgd > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cv -o c -- python $ROOT/tools/bench_costvol.py --layout ndhwc > $OUT/bench_costvol_ndhwc.log 2>&1
python - <<PY
import csv, json
rows = list(csv.DictReader(open("/tmp/prof_cv/c_kernel_stats.csv")))
with open("$OUT/costvol_kernel_stats_ndhwc.csv", "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python tools/bench_costvol.py --layout ndhwc\n")
    f.write("name,calls,avg_us,min_us,max_us\n")
    for r in rows:
        if "costvol" in r["Name"] or "cl_fwd_kernel" in r["Name"] or "cl_bwd_kernel" in r["Name"]:
            f.write("\"%s\",%s,%.2f,%.2f,%.2f\n" % (r["Name"][:120], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
vals = {}
sec = None
for line in open("$OUT/costvol_pmc_ndhwc.txt"):
    t = line.split()
    if line.startswith("fwd") or line.startswith("bwd"):
        sec = t[0]
    elif sec and len(t) == 2:
        vals[(sec, t[0])] = float(t[1])
import subprocess, time
w = vals.get(("fwd", "WRITE_SIZE"), 0) * 1024
fch = vals.get(("fwd", "FETCH_SIZE"), 0) * 1024
bw = vals.get(("bwd", "WRITE_SIZE"), 0) * 1024
bf = vals.get(("bwd", "FETCH_SIZE"), 0) * 1024
try:
    commit = subprocess.check_output(["git", "-C", "$ROOT", "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
except Exception:
    commit = "n/a (gpurun snapshot has no .git)"
import sys
sys.path.insert(0, "$ROOT")
from bench import costvol_source_hash
json.dump({"kernel": "cl_fwd_kernel<2,4,8,true,true> = md_costvol_fwd (B=6, 48x160, D=96, C=32, G=16, channels-last features and volume, fused schedule)",
           "collected": time.strftime("%Y-%m-%d %H:%M:%S UTC", time.gmtime()), "commit": commit,
           "kernel_source_sha256": costvol_source_hash(),
           "command": "tools/pmc_costvol.sh (rocprofv3 --kernel-trace --pmc <one group per pass> -- python tools/bench_costvol.py --iters 5 --layout ndhwc --feat nhwc)",
           "write_bytes_per_launch": w, "fetch_bytes_per_launch_raw": fch,
           "fetch_note": "FETCH_SIZE under-reports wide coalesced reads by up to 2x on gfx950 (MI355X_MICROARCH.md); "
                         "hbm_bytes_per_launch takes the 2x upper bound for the read side",
           "hbm_bytes_per_launch": w + 2 * fch,
           "bwd_kernel": "cl_bwd_kernel<2,4,4,true,true> = md_costvol_bwd", "bwd_write_bytes_per_launch": bw,
           "bwd_fetch_bytes_per_launch_raw": bf, "bwd_hbm_bytes_per_launch_upper": bw + 2 * bf},
          open("$OUT/costvol_fwd_pmc.json", "w"), indent=1)
print(open("$OUT/costvol_fwd_pmc.json").read())
PY
ls -la $OUT
