#!/bin/bash
# SQ / LDS counter passes for the headline kernel (separate runs, kernel-trace only).  bash tools/pmc.sh <tag> [bench args]
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --prewarm-ms 100 --no-cpu-baseline $*"
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/p$i" -o p -- python $ROOT/bench.py $ARGS > "$OUT/p$i.json" 2> "$OUT/p$i.log"
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mdvt::" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for (k, c), v in sorted(agg.items()):
        line = f"{k:42s} {c:24s} n={len(v):3d} mean={sum(v)/len(v):.4g}"
        print(line); fo.write(line + "\n")
PY
