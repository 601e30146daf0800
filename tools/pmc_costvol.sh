#!/bin/bash
# Collect PMC counters for the cost-volume kernels (separate passes; --kernel-trace only, as gpurun requires).
# usage: tools/pmc_costvol.sh <outdir> [bench args...]
set -u
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "GRBM_GUI_ACTIVE GRBM_TA_BUSY SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_ANY"
 "WRITE_SIZE"
 "FETCH_SIZE"
)
NP=${PMC_PASSES:-5}
i=0
for P in "${PASSES[@]:0:$NP}"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/pmc_$i -o p -- python $ROOT/tools/bench_costvol.py --iters 5 "$@" > /tmp/pmc_$i.log 2>&1
  i=$((i+1))
done
python - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("/tmp/pmc_*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "costvol" in r["Kernel_Name"] or "cl_fwd" in r["Kernel_Name"] or "cl_bwd" in r["Kernel_Name"]:
            k = "fwd" if "_fwd" in r["Kernel_Name"] else "bwd"
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
lines = []
for k, v in agg.items():
    lines.append("%s  avg_us(profiled)=%.1f" % (k, sum(dur[k]) / len(dur[k])))
    for c in sorted(v):
        lines.append("    %-24s %.4g" % (c, sum(v[c]) / len(v[c])))
open("$OUT", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
