cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ov
rocprofv3 --kernel-trace --output-format csv -d /tmp/ov -o t -- python $GRAFT_REPO_ROOT/tools/kbench.py default --rounds 1 --calls 2 --mesh --infill --conv 2.5 --frames 16 > /dev/null 2>&1
f=$(find /tmp/ov -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "mdvt::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[-9]["Start_Timestamp"])
for r in rows[-9:]:
    print(f'{r["Kernel_Name"].split("(")[0][-36:]:38s} q={r.get("Queue_Id","?"):>3s} start {(int(r["Start_Timestamp"])-t0)/1e3:8.1f} end {(int(r["End_Timestamp"])-t0)/1e3:8.1f}')
PY
