#!/bin/bash
# rocprofv3 kernel-trace stats of the training step (bench.py, config 2) in steady state: a first, unprofiled run
# fills a private MIOpen find-db / kernel cache, the profiled run re-uses it, so the trace holds (almost) no
# solver-search kernels.  usage: tools/profile_step.sh <out.csv>
OUT=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export MIOPEN_USER_DB_PATH=/tmp/md_prof_miopen/db MIOPEN_CUSTOM_CACHE_DIR=/tmp/md_prof_miopen/cache
mkdir -p $MIOPEN_USER_DB_PATH $MIOPEN_CUSTOM_CACHE_DIR
cp -r $ROOT/movedepth_amd/miopen_cache/db/. $MIOPEN_USER_DB_PATH/ 2>/dev/null
cp -r $ROOT/movedepth_amd/miopen_cache/cache/. $MIOPEN_CUSTOM_CACHE_DIR/ 2>/dev/null
python $ROOT/bench.py --steps 3 --warmup 2 --no_cpu_baseline > /tmp/prof_warm.log 2>&1
STEPS=10; WARM=3
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_step -o s -- python $ROOT/bench.py --steps $STEPS --warmup $WARM --no_cpu_baseline > /tmp/prof_step.log 2>&1
tail -1 /tmp/prof_step.log | cut -c1-400
python - <<PY
import csv
n = $STEPS + $WARM
rows = list(csv.DictReader(open("/tmp/prof_step/s_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
naive = sum(float(r["TotalDurationNs"]) for r in rows if "naive_conv" in r["Name"])
with open("$OUT", "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps %d --warmup %d  (%d steps; MIOpen caches warmed by a prior run)\n" % ($STEPS, $WARM, n))
    f.write("# total kernel time %.1f ms = %.2f ms per step; of that solver-search (naive_conv*) kernels %.1f ms\n" % (tot / 1e6, tot / 1e6 / n, naive / 1e6))
    f.write("name,calls,total_ms,ms_per_step,avg_us,min_us,max_us,percent\n")
    mine = ("costvol", "cl_fwd", "cl_bwd", "warp_", "ssim_", "reproj_", "masked_min", "smooth_", "sel_", "schedule_", "fuse_", "disp_up", "conv3d_c")
    keep = rows[:45] + [r for r in rows[45:] if any(m in r["Name"] for m in mine)]
    for r in keep:
        f.write("\"%s\",%s,%.3f,%.3f,%.2f,%.2f,%.2f,%s\n" % (r["Name"][:140], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                float(r["TotalDurationNs"]) / 1e6 / n, float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
# categories
cat = {"library 3-D conv (CK / naive)": ("ck::", "_ZN2ck", "naive_conv"), "library 2-D conv (Winograd / igemm)": ("miopenSp3AsmConv", "igemm_", "gemm_", "Cijk_"),
       "BatchNorm": ("BatchNorm",), "elementwise / copy / transpose / fill (torch, MIOpen)": ("elementwise", "transpose", "fillBuffer", "SubTensor", "copyBuffer", "CatArray", "reduce_kernel", "upsample", "reflection_pad", "index", "multi_tensor", "fused_adam"),
       "hand-written: cost volume": ("costvol", "cl_fwd_kernel", "cl_bwd_kernel"), "hand-written: reg3d first/last conv": ("conv3d_c",),
       "hand-written: fused BatchNorm+ReLU": ("bn_",),
       "hand-written: photometric + post-volume": ("warp_", "ssim_", "reproj_", "masked_min", "smooth_", "sel_", "schedule_", "fuse_", "disp_up", "convex", "backproject", "project3d")}
acc = {k: 0.0 for k in cat}; other = 0.0
for r in rows:
    t = float(r["TotalDurationNs"]) / 1e6 / n
    for k, pats in cat.items():
        if any(p in r["Name"] for p in pats):
            acc[k] += t; break
    else:
        other += t
with open("$OUT", "a") as f:
    f.write("# ---- ms per step by category\n")
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        f.write("# %-60s %7.2f\n" % (k, v))
    f.write("# %-60s %7.2f\n" % ("other", other))
print(open("$OUT").read()[-1800:])
PY
