#!/bin/bash
# PMC counters of the fused photometric kernels (csrc/photo.hip), one counter group per pass (--kernel-trace only, as gpurun
# requires).  usage: tools/pmc_photo.sh <out.txt>
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
i=0
for P in "${PASSES[@]}"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/pmcp_$i -o p -- python $ROOT/tools/bench_photo.py --iters 4 --unfused 0 "$@" > /tmp/pmcp_$i.log 2>&1
  i=$((i+1))
done
python - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("/tmp/pmcp_*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if not any(s in n for s in ("photo_", "up_adjoint")):
            continue
        grid = r.get("Grid_Size", "")
        k = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0] + " grid=" + grid
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
lines = ["# rocprofv3 --pmc, tools/bench_photo.py (B=6, 192x640, 2 source frames; grid = threads: the mono group is 4 scales, the MVS group 1)",
         "# FETCH_SIZE / WRITE_SIZE in KB as reported (gfx950: double FETCH_SIZE for wide coalesced reads, MI355X_MICROARCH.md)"]
for k in sorted(agg):
    v = agg[k]
    lines.append("%s  avg_us(profiled)=%.1f" % (k, sum(dur[k]) / len(dur[k])))
    for c in sorted(v):
        lines.append("    %-24s %.4g" % (c, sum(v[c]) / len(v[c])))
open("$OUT", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
