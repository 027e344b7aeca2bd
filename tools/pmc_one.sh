#!/bin/bash
# SQ counter passes for ONE kbench configuration value (separate runs, kernel-trace only).
#   bash tools/pmc_one.sh <tag> <cfg value> <kbench args incl. --env NAME>
set -u
TAG=$1; CFG=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc1_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_EA_WRREQ_sum TCC_EA_RDREQ_sum" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_FLAT SQ_INSTS_SMEM" \
           "FETCH_SIZE" "WRITE_SIZE" ; do
  i=$((i+1))
  timeout 180 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/p$i" -o p -- python $ROOT/tools/kbench.py $CFG --rounds 2 --calls 3 "$@" > /dev/null 2> "$OUT/p$i.log"
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mdvt::" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0][-44:], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for (k, c), v in sorted(agg.items()):
        line = f"{k:46s} {c:24s} n={len(v):3d} mean={sum(v)/len(v):.4g}"
        print(line); fo.write(line + "\n")
PY
