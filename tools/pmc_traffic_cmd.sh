#!/bin/bash
# HBM traffic of the kernels of any command, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in separate rocprofv3
# passes (kernel trace only), counter x 1024 B, FETCH_SIZE x 2 on gfx950.  Per-kernel means per launch.
#   bash tools/pmc_traffic_cmd.sh <tag> <kernel-name substring> -- <command ...>
set -u
TAG=$1; FILTER=$2; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmct_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/$c" -o p -- "$@" > /dev/null 2> "$OUT/$c.log"
done
python - "$OUT" "$FILTER" <<'PY'
import csv, glob, sys, collections
out, flt = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if flt in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
names = sorted({k for k, _ in agg})
tot_r = tot_w = 0.0
with open(out + "/summary.txt", "w") as fo:
    for k in names:
        rd = agg.get((k, "FETCH_SIZE"), [0]); wr = agg.get((k, "WRITE_SIZE"), [0])
        r_mb = sum(rd) / len(rd) * 1024 * 2 / 1e6; w_mb = sum(wr) / len(wr) * 1024 / 1e6
        tot_r += r_mb; tot_w += w_mb
        line = f"{k:42s} launches {len(rd):4d}  read {r_mb:9.2f} MB  written {w_mb:9.2f} MB per launch"
        print(line); fo.write(line + "\n")
    line = f"{'sum over the kernels (one launch each)':42s}                read {tot_r:9.2f} MB  written {tot_w:9.2f} MB"
    print(line); fo.write(line + "\n")
PY
