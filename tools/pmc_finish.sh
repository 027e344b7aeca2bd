set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_EA_WRREQ_sum TCC_EA_RDREQ_sum TCP_TCC_READ_REQ_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmf/p$i -o p -- python $ROOT/tools/finish_bench.py --frames 16 --reps 2 "$@" > /dev/null 2> /tmp/pmf_$i.log
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pmf/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_telea_need" in k or "k_telea_fill" in k:
            agg[(k.split("(")[0][-20:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{k:22s} {c:24s} n={len(v):4d} sum={sum(v):.4g} mean={sum(v)/len(v):.4g}")
PY
