#!/bin/bash
# Memory-pipeline counters of the plane-sweep kernels (round 6), separate --pmc passes with --kernel-trace only (as gpurun requires):
#   tools/pmc_costvol2.sh <out.txt> [bench_costvol args...]     MOVEDEPTH_HIP_LIB selects an A/B build
set -u
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY"
 "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_LEVEL_WAVES"
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WRITE_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
 "SQ_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_IFETCH"
)
i=0
for P in "${PASSES[@]}"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/pmc2_$i -o p -- python $ROOT/tools/bench_costvol.py --iters 5 --layout ndhwc --feat nhwc "$@" > /tmp/pmc2_$i.log 2>&1 || tail -3 /tmp/pmc2_$i.log
  i=$((i+1))
done
python - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("/tmp/pmc2_*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "cl_fwd" in r["Kernel_Name"] or "cl_bwd" in r["Kernel_Name"]:
            k = "fwd" if "_fwd" in r["Kernel_Name"] else "bwd"
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
lines = ["# tools/pmc_costvol2.sh $*"]
for k, v in agg.items():
    lines.append("%s  avg_us(profiled)=%.1f" % (k, sum(dur[k]) / len(dur[k])))
    for c in sorted(v):
        lines.append("    %-40s %.4g" % (c, sum(v[c]) / len(v[c])))
open("$OUT", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf /tmp/pmc2_*
