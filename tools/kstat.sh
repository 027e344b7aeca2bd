#!/bin/bash
# Per-kernel average durations of a tools/kbench.py configuration (rocprofv3 --kernel-trace --stats).  bash tools/kstat.sh <kbench args>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
D=$(mktemp -d /tmp/kstat.XXXX)
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python $ROOT/tools/kbench.py default --rounds 2 --calls 5 "$@" > /dev/null 2>&1
f=$(find $D -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "mdvt" in r["Name"]:
        print(f'{r["Name"].split("(")[0][-48:]:50s} calls {int(r["Calls"]):4d}  avg {float(r["AverageNs"])/1e3:9.1f} us')
PY
