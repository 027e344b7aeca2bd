#!/bin/bash
# Per-level durations of the infill-mask completion's need / fill launches and per-kernel totals of one pass (GPU box).
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf
rocprofv3 --kernel-trace --output-format csv -d /tmp/pf -o t -- python $GRAFT_REPO_ROOT/tools/finish_bench.py --frames ${1:-16} --reps 1 ${@:2} > /dev/null 2>&1
f=$(find /tmp/pf -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
need, fill, tot = [], [], collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]; d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "k_telea_need" in n: need.append(d)
    if "k_telea_fill" in n: fill.append(d)
    if "telea" in n or "blur" in n:
        k = n.split("(")[0].split("::")[-1]; tot[k] = tot.get(k, 0) + d
print("need R..2", [round(x) for x in need])
print("fill 1..R", [round(x) for x in fill])
for k, v in tot.items(): print(f"{k:28s} {v:8.0f} us")
PY
