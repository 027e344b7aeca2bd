#!/bin/bash
# Dynamic VALU instruction mix of a tools/kbench.py configuration by class (separate rocprofv3 counter runs, kernel-trace
# only), for pricing a kernel against the per-class issue rates tools/probe/valu_probe.hip measures.
#   bash tools/pmc_mix.sh <tag> <kbench args>      e.g.  bash tools/pmc_mix.sh band --mesh
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcmix_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="default --rounds 2 --calls 3 $*"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/p$i" -o p -- python $ROOT/tools/kbench.py $ARGS > /dev/null 2> "$OUT/p$i.log"
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
dur = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mdvt::" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0][-44:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for f in glob.glob(out + "/p*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mdvt::" in r["Kernel_Name"]:
            dur[r["Kernel_Name"].split("(")[0][-44:]].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3)
with open(out + "/summary.txt", "w") as fo:
    for k, v in sorted(dur.items()):
        line = f"{k:46s} {'duration_us (under counters)':24s} n={len(v):3d} mean={sum(v)/len(v):.4g}"
        print(line); fo.write(line + "\n")
    for (k, c), v in sorted(agg.items()):
        line = f"{k:46s} {c:24s} n={len(v):3d} mean={sum(v)/len(v):.4g}"
        print(line); fo.write(line + "\n")
PY
