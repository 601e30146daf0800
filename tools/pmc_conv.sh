#!/bin/bash
# PMC counters for the regulariser's hand-written convolutions (separate passes; --kernel-trace only).
# usage: [SCRIPTS=bench_conv3d_c1] tools/pmc_conv.sh <out.txt>
set -u
OUT=$1
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"
 "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
 "WRITE_SIZE"
 "FETCH_SIZE"
)
i=0
for P in "${PASSES[@]}"; do
  for S in ${SCRIPTS:-bench_conv3d_c16 bench_conv3d_c1}; do
    NO_LIB=1 rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/pmcc_${i}_$S -o p -- python $ROOT/tools/$S.py > /tmp/pmcc_${i}_$S.log 2>&1
  done
  i=$((i+1))
done
python - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("/tmp/pmcc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "conv3d_c" in n and "finish" not in n:
            import re
            k = re.search(r"(conv3d_c\w+?_kernel(<[^>]*>)?)", n).group(1)
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
lines = []
for k in sorted(agg):
    v = agg[k]
    lines.append("%s  avg_us(profiled)=%.1f  launches=%d" % (k, sum(dur[k]) / len(dur[k]), len(dur[k]) // max(len(v), 1)))
    for c in sorted(v):
        lines.append("    %-28s %.4g" % (c, sum(v[c]) / len(v[c])))
    g = lambda c: sum(v[c]) / len(v[c]) if c in v else float("nan")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over all 1024 SIMDs
        cyc = g("GRBM_GUI_ACTIVE") / 8
        us = sum(dur[k]) / len(dur[k])
        lines.append("    => core clock during the kernel %.2f GHz; MFMA pipes busy %.1f %% of the time"
                     % (cyc / us / 1e3, 100 * g("SQ_VALU_MFMA_BUSY_CYCLES") / (cyc * 1024)))
open("$OUT", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
