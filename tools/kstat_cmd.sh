#!/bin/bash
# Per-kernel average durations of any command (rocprofv3 --kernel-trace --stats).  bash tools/kstat_cmd.sh <filter> <command ...>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
FILTER=$1; shift
cd /tmp && export TMPDIR=/tmp
D=$(mktemp -d /tmp/kstat.XXXX)
( cd $ROOT && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- "$@" ) > $D/cmd.out 2>&1
tail -3 $D/cmd.out
f=$(find $D -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$FILTER" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"]:
        nm = r["Name"].replace("(anonymous namespace)::", "")
        print(f'{nm.split("(")[0][-52:]:54s} calls {int(r["Calls"]):5d}  avg {float(r["AverageNs"])/1e3:9.1f} us  total {float(r["TotalDurationNs"])/1e6:9.2f} ms')
PY
