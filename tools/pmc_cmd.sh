#!/bin/bash
# SQ / TCC counter passes (separate rocprofv3 runs, kernel-trace only, each under a timeout) for any command; per-kernel sums.
#   bash tools/pmc_cmd.sh <tag> <kernel-name substring> -- <command ...>
set -u
TAG=$1; FILTER=$2; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_EA_WRREQ_sum TCC_EA_RDREQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" ; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/p$i" -o p -- "$@" > /dev/null 2> "$OUT/p$i.log"
done
python - "$OUT" "$FILTER" <<'PY'
import csv, glob, sys, collections
out, flt = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if flt in r["Kernel_Name"]:
            k = (r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])
            agg[k] += float(r["Counter_Value"]); cnt[k] += 1
with open(out + "/summary.txt", "w") as fo:
    for (k, c), v in sorted(agg.items()):
        line = f"{k:42s} {c:28s} launches={cnt[(k,c)]:5d} sum={v:.4g}"
        print(line); fo.write(line + "\n")
PY
