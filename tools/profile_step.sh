#!/bin/bash
# rocprofv3 kernel trace of the training step (bench.py, config 2), summarised over WHOLE TIMED STEPS only.
# MIOpen re-runs its solver search for the 3-D regulariser's convolutions at every process start (~30 s of naive_conv* and
# candidate kernels inside the first step: the results of that search are not kept in the user find-db), so a summary over the
# whole process is dominated by search kernels.  The summary below is cut from the per-dispatch trace instead: the window runs
# from the identity-loss kernel (once per step) of the last warm-up step to that of the last timed step = STEPS whole periods
# of the step, none of them a first step.  usage: tools/profile_step.sh <out.csv>
OUT=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
STEPS=${STEPS:-10}; WARM=${WARM:-3}
rm -rf /tmp/prof_step
MD_BENCH_PHOTO_STEPS=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_step -o s -- python $ROOT/bench.py --steps $STEPS --warmup $WARM --no_cpu_baseline --trainer_args="${TRAINER_ARGS:-}" > /tmp/prof_step.log 2>&1
tail -1 /tmp/prof_step.log | cut -c1-300
python - <<PY
import csv, json, collections
steps, warm = $STEPS, $WARM
rows = list(csv.DictReader(open("/tmp/prof_step/s_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [int(r["Start_Timestamp"]) for r in rows if "photo_fwd_kernel" in r["Kernel_Name"] and ", true>" in r["Kernel_Name"]]
assert len(marks) == steps + warm, "expected %d identity-loss kernels, found %d" % (steps + warm, len(marks))
t0, t1 = marks[warm - 1], marks[-1]
win = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
naive = [r for r in win if "naive_conv" in r["Kernel_Name"]]
agg = collections.OrderedDict()
for r in win:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(r["Kernel_Name"], [0, 0.0, 1e30, 0.0])
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
busy = sum(a[1] for a in agg.values()) / 1e3 / steps
try:
    line = json.loads([l for l in open("/tmp/prof_step.log").read().splitlines() if l.startswith('{"metric"')][-1])
    bench_ms = line["ms_per_step"]
except Exception:
    bench_ms = float("nan")
cat = {"library 3-D conv (CK)": ("ck::", "_ZN2ck", "naive_conv"), "library 2-D conv (Winograd / igemm / GEMM)": ("miopenSp3AsmConv", "igemm_", "gemm_", "Cijk_", "Im2d2Col", "Col2Im"),
       "BatchNorm (library)": ("BatchNorm", "batch_norm"),
       "elementwise / copy / transpose / fill / optimizer (torch, MIOpen)": ("elementwise", "transpose", "fillBuffer", "SubTensor", "copyBuffer", "CatArray", "reduce_kernel", "upsample", "reflection_pad", "index", "multi_tensor", "fused_adam", "distribution", "OpTensor"),
       "hand-written: cost volume": ("costvol", "cl_fwd_kernel", "cl_bwd_kernel"), "hand-written: reg3d first/last conv": ("conv3d_c",),
       "hand-written: BatchNorm (fused BatchNorm+ReLU of reg3d; synchronised BatchNorm)": ("bn_",),
       "hand-written: photometric + post-volume": ("photo_", "up_adjoint", "warp_", "ssim_", "reproj_", "masked_min", "smooth_", "sel_", "sel4_", "schedule_", "fuse_", "disp_up", "convex", "backproject", "project3d", "pose_")}
acc = {k: 0.0 for k in cat}; other = 0.0
mine = tuple(p for k, v in cat.items() if k.startswith("hand-written") for p in v)
for n, a in agg.items():
    for k, pats in cat.items():
        if any(p in n for p in pats):
            acc[k] += a[1] / 1e3 / steps; break
    else:
        other += a[1] / 1e3 / steps
with open("$OUT", "w") as f:
    f.write("# rocprofv3 --kernel-trace -- python bench.py --steps %d --warmup %d; %d whole steps cut from the dispatch trace (tools/profile_step.sh)\n" % (steps, warm, steps))
    f.write("# kernel time %.2f ms per step (sum of kernel durations in the window); bench.py's wall clock in the same run: %.2f ms per step; solver-search (naive_conv*) dispatches in the window: %d\n" % (busy, bench_ms, len(naive)))
    f.write("name,calls_per_step,ms_per_step,avg_us,min_us,max_us\n")
    order = sorted(agg.items(), key=lambda kv: -kv[1][1])
    keep = order[:40] + [kv for kv in order[40:] if any(m in kv[0] for m in mine)]
    for n, a in keep:
        f.write("\"%s\",%.1f,%.4f,%.2f,%.2f,%.2f\n" % (n[:150], a[0] / steps, a[1] / 1e3 / steps, a[1] / a[0], a[2], a[3]))
    f.write("# ---- ms per step by category\n")
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        f.write("# %-70s %7.2f\n" % (k, v))
    f.write("# %-70s %7.2f\n" % ("other", other))
    f.write("# %-70s %7.2f\n" % ("sum", other + sum(acc.values())))
print(open("$OUT").read()[-2500:])
# neighbourhood of the plane-sweep launches inside one step (what runs just before matters: dirty lines still being written back)
print("# ---- plane-sweep launches of the last timed step: two predecessors, gap, own duration")
last = [r for r in rows if marks[-2] <= int(r["Start_Timestamp"]) < marks[-1]]
for i, r in enumerate(last):
    if "cl_fwd_kernel" in r["Kernel_Name"] or "cl_bwd_kernel" in r["Kernel_Name"]:
        for q in last[max(i - 2, 0):i + 1]:
            print("#   %-90s %8.1f us  (starts %8.1f us after the previous kernel ended)" % (q["Kernel_Name"][:90], (int(q["End_Timestamp"]) - int(q["Start_Timestamp"])) / 1e3,
                  (int(q["Start_Timestamp"]) - int(last[last.index(q) - 1]["End_Timestamp"])) / 1e3 if last.index(q) else 0.0))
        print("#")
PY
